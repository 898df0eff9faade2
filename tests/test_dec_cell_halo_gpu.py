"""
ga_dec_cell_halo (the fused residual branch of NVAE's ResidualCellDecoder for the few-channel post-processing cells,
architecture.py:139-186 as model.py:211-228 instantiates them; an 8 x 16 tile per workgroup with a recomputed halo) against the same
math in plain PyTorch fp32 on the CPU, forward and through autograd (d x, with the identity-skip addend), at image sizes with
interior tiles, border tiles and both at once.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

if not torch.cuda.is_available():
    pytest.skip('needs a GPU', allow_module_level=True)

from gen_adversarial_amd import _lib as L   # noqa: E402
from test_ops_gpu import DEV, close, g, nchw, nhwc   # noqa: E402
from test_dec_cell_gpu import _cell, _split   # noqa: E402

CASES = [(2, 16, 32, 32, 96), (1, 64, 64, 32, 96), (3, 32, 32, 64, 192), (2, 8, 16, 64, 64), (1, 24, 48, 32, 32)]


def _descs(N, H, W, Cc, Hd, x, w1, b1, wd, bd, w2, b2):
    xd = nhwc(x)
    w1f, w2f = w1[:, :, 0, 0].contiguous().to(DEV), w2[:, :, 0, 0].contiguous().to(DEV)       # [Hd][C], [C][Hd]
    keep = dict(xd=xd, wdf=wd.reshape(Hd, 25).t().contiguous().to(DEV), wdb=wd.flip(2, 3).reshape(Hd, 25).t().contiguous().to(DEV),
                b1d=b1.to(DEV), bdd=bd.to(DEV), b2d=b2.to(DEV), w1=_split(w1f), w2=_split(w2f), w2t=_split(w2f.t().contiguous()),
                w1t=_split(w1f.t().contiguous()))

    def base(backward):
        d = L.DecCellHaloDesc()
        d.x, d.b1, d.wd, d.wd_bwd, d.bd = xd.data_ptr(), keep['b1d'].data_ptr(), keep['wdf'].data_ptr(), keep['wdb'].data_ptr(), keep['bdd'].data_ptr()
        d.w1_hi, d.w1_lo = (t.data_ptr() for t in keep['w1'])
        d.w2_hi, d.w2_lo = (t.data_ptr() for t in keep['w2t' if backward else 'w2'])
        d.w1t_hi, d.w1t_lo = (t.data_ptr() for t in keep['w1t'])
        d.b2 = keep['b2d'].data_ptr()
        d.N, d.H, d.W, d.Cin, d.Cout, d.Hd, d.backward, d.up = N, H, W, Cc, Cc, Hd, backward, 0
        return d
    return base, keep


@pytest.mark.parametrize('N,H,W,Cc,Hd', CASES)
def test_dec_cell_halo_forward_and_backward(N, H, W, Cc, Hd):
    assert L.lib.ga_dec_cell_halo_supported(N, H, W, Cc, Hd) == 1
    gen = torch.Generator().manual_seed(7)
    x = torch.randn(N, Cc, H, W, generator=gen)
    _, w1, b1, wd, bd, w2, b2 = _cell(1, 8, Cc, Hd, seed=11)
    xr = x.clone().requires_grad_(True)
    t1 = F.conv2d(xr, w1, b1)
    t2 = F.conv2d(F.silu(t1), wd, bd, padding=2, groups=Hd)
    t3 = F.conv2d(F.silu(t2), w2, b2)
    base, keep = _descs(N, H, W, Cc, Hd, x, w1, b1, wd, bd, w2, b2)

    y = torch.full((N, H, W, Cc), float('nan'), device=DEV)
    d = base(0)
    d.y = y.data_ptr()
    L.run(d)
    torch.cuda.synchronize()
    close(nchw(y), t3, 2e-4, 'fused forward vs torch')

    # ---- backward: d loss / d x for d loss / d t3 = dout * ps[n] + pb[n], plus the identity-skip addend
    dout = g(N, Cc, H, W, seed=21)
    ps = g(N, Cc, seed=22).abs() * 0.1 + 0.05
    pb = g(N, Cc, seed=23) * 0.01
    add = g(N, Cc, H, W, seed=24)
    dt3 = dout * ps.view(N, Cc, 1, 1) + pb.view(N, Cc, 1, 1)
    (gx,) = torch.autograd.grad((t3 * dt3).sum(), [xr])
    dd, psd, pbd, addd = nhwc(dout), ps.to(DEV), pb.to(DEV), nhwc(add)
    dx = torch.full((N, H, W, Cc), float('nan'), device=DEV)
    b = base(1)
    b.dout, b.pro_scale, b.pro_shift, b.addend, b.y = dd.data_ptr(), psd.data_ptr(), pbd.data_ptr(), addd.data_ptr(), dx.data_ptr()
    L.run(b)
    torch.cuda.synchronize()
    close(nchw(dx), gx + add, 2e-4, 'fused backward (dx + addend) vs autograd')
    b.addend = None
    L.run(b)
    torch.cuda.synchronize()
    close(nchw(dx), gx, 2e-4, 'fused backward vs autograd')


def test_dec_cell_halo_refuses_unsupported_shapes():
    sup = L.lib.ga_dec_cell_halo_supported
    assert sup(2, 64, 64, 32, 96) == 1 and sup(2, 32, 32, 64, 192) == 1
    assert sup(2, 64, 64, 128, 768) == 0           # wide cells: ga_dec_cell (whole images) or the unfused launches
    assert sup(2, 12, 16, 32, 96) == 0 and sup(2, 8, 24, 32, 96) == 0      # not whole 8 x 16 tiles
    assert sup(2, 64, 64, 32, 100) == 0
    d = L.DecCellHaloDesc()
    d.N, d.H, d.W, d.Cin, d.Cout, d.Hd = 2, 64, 64, 32, 32, 96
    with pytest.raises(L.GaError):
        L.run(d)                                   # null pointers
