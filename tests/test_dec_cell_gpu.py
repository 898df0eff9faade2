"""
ga_dec_cell (the fused residual branch of NVAE's ResidualCellDecoder, architecture.py:139-186) against (a) the same math in
plain PyTorch fp32 on the CPU, forward and through autograd, and (b) the three unfused launches it replaces
(ga_conv2d split-bf16 1x1 -> ga_dwconv5 -> ga_conv2d), which it follows to summation order.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

if not torch.cuda.is_available():
    pytest.skip('needs a GPU', allow_module_level=True)

from gen_adversarial_amd import _lib as L   # noqa: E402
from test_ops_gpu import DEV, close, g, nchw, nhwc, run_conv   # noqa: E402


def _split(w):
    hi = w.to(torch.bfloat16)
    return hi.contiguous(), (w - hi.float()).to(torch.bfloat16).contiguous()


def _cell(N, H, Cc, Hd, seed):
    x = g(N, Cc, H, H, seed=seed)
    w1 = g(Hd, Cc, 1, 1, seed=seed + 1, scale=1.0 / np.sqrt(Cc))
    b1 = g(Hd, seed=seed + 2, scale=0.3)
    wd = g(Hd, 1, 5, 5, seed=seed + 3, scale=0.2)
    bd = g(Hd, seed=seed + 4, scale=0.3)
    w2 = g(Cc, Hd, 1, 1, seed=seed + 5, scale=1.0 / np.sqrt(Hd))
    b2 = g(Cc, seed=seed + 6, scale=0.3)
    return x, w1, b1, wd, bd, w2, b2


def _torch_cell(x, w1, b1, wd, bd, w2, b2):
    t1 = (F.conv2d(x, w1, b1)).detach().requires_grad_(True)
    t2 = F.conv2d(F.silu(t1), wd, bd, padding=2, groups=wd.shape[0])
    t3 = F.conv2d(F.silu(t2), w2, b2)
    return t1, t2, t3


CASES = [(2, 16, 128, 768), (1, 16, 128, 96), (4, 8, 256, 1536), (6, 8, 256, 64), (3, 16, 128, 32)]


@pytest.mark.parametrize('N,H,Cc,Hd', CASES)
def test_dec_cell_forward_and_backward(N, H, Cc, Hd):
    assert L.lib.ga_dec_cell_supported(N, H, H, Cc, Hd) == 1
    x, w1, b1, wd, bd, w2, b2 = _cell(N, H, Cc, Hd, seed=11)
    t1, t2, t3 = _torch_cell(x, w1, b1, wd, bd, w2, b2)
    xd = nhwc(x)
    w1f, w2f = w1[:, :, 0, 0].contiguous().to(DEV), w2[:, :, 0, 0].contiguous().to(DEV)       # [Hd][C], [C][Hd]
    w2t = w2f.t().contiguous()                                                                  # [Hd][C]
    wdf = wd.reshape(Hd, 25).t().contiguous().to(DEV)
    wdb = wd.flip(2, 3).reshape(Hd, 25).t().contiguous().to(DEV)
    b1d, bdd, b2d = b1.to(DEV), bd.to(DEV), b2.to(DEV)
    w1h, w1l = _split(w1f)
    w2h, w2l = _split(w2f)
    w2th, w2tl = _split(w2t)

    y = torch.full((N, H, H, Cc), float('nan'), device=DEV)
    d = L.DecCellDesc()
    d.x, d.w1_hi, d.w1_lo, d.b1 = xd.data_ptr(), w1h.data_ptr(), w1l.data_ptr(), b1d.data_ptr()
    d.wd, d.wd_bwd, d.bd = wdf.data_ptr(), wdb.data_ptr(), bdd.data_ptr()
    d.w2_hi, d.w2_lo, d.b2, d.y = w2h.data_ptr(), w2l.data_ptr(), b2d.data_ptr(), y.data_ptr()
    d.N, d.H, d.W, d.C, d.Hd, d.backward = N, H, H, Cc, Hd, 0
    L.run(d)
    torch.cuda.synchronize()
    close(nchw(y), t3, 2e-4, 'fused forward vs torch')

    # the three launches it replaces
    u1 = torch.empty(N, H, H, Hd, device=DEV)
    run_conv(xd, w1f, u1, 1, bias=b1d, w_hi=w1h, w_lo=w1l)
    u2 = torch.empty(N, H, H, Hd, device=DEV)
    dw = L.DwDesc()
    dw.x, dw.w, dw.bias, dw.y = u1.data_ptr(), wdf.data_ptr(), bdd.data_ptr(), u2.data_ptr()
    dw.N, dw.H, dw.W, dw.C, dw.pro_act = N, H, H, Hd, L.GA_ACT_SILU
    L.run(dw)
    u3 = torch.empty(N, H, H, Cc, device=DEV)
    run_conv(u2, w2f, u3, 1, bias=b2d, pro_act=L.GA_ACT_SILU, w_hi=w2h, w_lo=w2l)
    torch.cuda.synchronize()
    scale = u3.abs().max().item()
    diff = (y - u3).abs().max().item()
    print(f'fused vs unfused forward: max |diff| {diff:.3e} of {scale:.3e}, bitwise equal: {torch.equal(y, u3)}')
    assert diff <= 2e-6 * scale
    if Cc == 128:                # the eight-wave form of the 128-channel cell: same sums in the same order
        y1 = torch.full((N, H, H, Cc), float('nan'), device=DEV)
        d.y, d.variant = y1.data_ptr(), 1
        L.run(d)
        torch.cuda.synchronize()
        assert torch.equal(y, y1), 'ga_dec_cell variant 1 (forward) differs from variant 0'

    # ---- backward: d loss / d t1 for d loss / d t3 = dout * ps[n] + pb[n]
    dout = g(N, Cc, H, H, seed=21)
    ps = g(N, Cc, seed=22).abs() * 0.1 + 0.05
    pb = g(N, Cc, seed=23) * 0.01
    dt3 = dout * ps.view(N, Cc, 1, 1) + pb.view(N, Cc, 1, 1)
    (gt1,) = torch.autograd.grad((t3 * dt3).sum(), [t1])
    dd, psd, pbd = nhwc(dout), ps.to(DEV), pb.to(DEV)
    dt1 = torch.full((N, H, H, Hd), float('nan'), device=DEV)
    b = L.DecCellDesc()
    b.x, b.w1_hi, b.w1_lo, b.b1 = xd.data_ptr(), w1h.data_ptr(), w1l.data_ptr(), b1d.data_ptr()
    b.wd, b.wd_bwd, b.bd = wdf.data_ptr(), wdb.data_ptr(), bdd.data_ptr()
    b.w2_hi, b.w2_lo = w2th.data_ptr(), w2tl.data_ptr()
    b.dout, b.pro_scale, b.pro_shift, b.y = dd.data_ptr(), psd.data_ptr(), pbd.data_ptr(), dt1.data_ptr()
    b.N, b.H, b.W, b.C, b.Hd, b.backward = N, H, H, Cc, Hd, 1
    L.run(b)
    torch.cuda.synchronize()
    close(nchw(dt1), gt1, 2e-4, 'fused backward vs autograd')
    if Cc == 128:
        dt1b = torch.full((N, H, H, Hd), float('nan'), device=DEV)
        b.y, b.variant = dt1b.data_ptr(), 1
        L.run(b)
        torch.cuda.synchronize()
        assert torch.equal(dt1, dt1b), 'ga_dec_cell variant 1 (backward) differs from variant 0'

    # unfused: 1x1 transpose with the per-row prologue and silu'(t2) epilogue, then the depthwise transpose with silu'(t1)
    v2 = torch.empty(N, H, H, Hd, device=DEV)
    run_conv(dd, w2t, v2, 1, pro_scale=psd, pro_shift=pbd, pro_per_row=1, dact_x=u2, dact_act=L.GA_ACT_SILU, lddact=Hd,
             w_hi=w2th, w_lo=w2tl)
    v1 = torch.empty(N, H, H, Hd, device=DEV)
    dwb = L.DwDesc()
    dwb.x, dwb.w, dwb.dact_x, dwb.y = v2.data_ptr(), wdb.data_ptr(), u1.data_ptr(), v1.data_ptr()
    dwb.N, dwb.H, dwb.W, dwb.C, dwb.dact_act = N, H, H, Hd, L.GA_ACT_SILU
    L.run(dwb)
    torch.cuda.synchronize()
    scale = v1.abs().max().item()
    diff = (dt1 - v1).abs().max().item()
    print(f'fused vs unfused backward: max |diff| {diff:.3e} of {scale:.3e}, bitwise equal: {torch.equal(dt1, v1)}')
    assert diff <= 2e-6 * scale


def test_dec_cell_on_a_side_stream_equals_the_null_stream():
    """the direct binding passes the hipStream_t untruncated: a launch on a fresh stream gives the NULL stream's bits"""
    N, H, Cc, Hd = 2, 16, 128, 96
    x, w1, b1, wd, bd, w2, b2 = _cell(N, H, Cc, Hd, seed=5)
    xd = nhwc(x)
    w1h, w1l = _split(w1[:, :, 0, 0].contiguous().to(DEV))
    w2h, w2l = _split(w2[:, :, 0, 0].contiguous().to(DEV))
    wdf = wd.reshape(Hd, 25).t().contiguous().to(DEV)
    wdb = wd.flip(2, 3).reshape(Hd, 25).t().contiguous().to(DEV)
    b1d, bdd, b2d = b1.to(DEV), bd.to(DEV), b2.to(DEV)
    outs = []
    for side in (False, True):
        y = torch.full((N, H, H, Cc), float('nan'), device=DEV)
        d = L.DecCellDesc()
        d.x, d.w1_hi, d.w1_lo, d.b1 = xd.data_ptr(), w1h.data_ptr(), w1l.data_ptr(), b1d.data_ptr()
        d.wd, d.wd_bwd, d.bd = wdf.data_ptr(), wdb.data_ptr(), bdd.data_ptr()
        d.w2_hi, d.w2_lo, d.b2, d.y = w2h.data_ptr(), w2l.data_ptr(), b2d.data_ptr(), y.data_ptr()
        d.N, d.H, d.W, d.C, d.Hd, d.backward = N, H, H, Cc, Hd, 0
        torch.cuda.synchronize()
        if side:
            s = torch.cuda.Stream(device=DEV)
            assert s.cuda_stream != 0
            L.run(d, s.cuda_stream)
            s.synchronize()
        else:
            L.run(d)
            torch.cuda.synchronize()
        outs.append(y)
    assert torch.isfinite(outs[1]).all() and torch.equal(outs[0], outs[1])


def test_dec_cell_refuses_unsupported_shapes():
    sup = L.lib.ga_dec_cell_supported
    assert sup(2, 16, 16, 128, 768) == 1 and sup(2, 8, 8, 256, 1536) == 1
    assert sup(2, 16, 16, 256, 1536) == 0          # an image larger than the 128-pixel workgroup
    assert sup(2, 4, 4, 512, 3072) == 0            # no kernel for 512 channels
    assert sup(3, 8, 8, 128, 768) == 0             # rows do not fill whole workgroups (4 images each)
    assert sup(2, 16, 16, 128, 100) == 0           # hidden width not a multiple of 32
    assert sup(4, 4, 4, 128, 768) == 0             # 8-pixel strips do not fit a 4-pixel row
    assert sup(32, 1, 8, 128, 768) == 0 and sup(1, 1, 256, 128, 768) == 0     # one-row images: no room for the 2 x 4 output blocks
    assert sup(4, 8, 8, 128, 768) == 0 and sup(8, 4, 4, 256, 1536) == 0      # framed planes of 4 / 8 images: beyond the LDS
    d = L.DecCellDesc()
    d.N, d.H, d.W, d.C, d.Hd = 2, 16, 16, 128, 768
    with pytest.raises(L.GaError):
        L.run(d)                                   # null pointers
