"""
GPU parity tests, one per C entry point of include/ga_ops.h: each kernel against the same op written with plain
PyTorch fp32 on the CPU (forward and, through autograd, backward-to-input).  Tolerances are stated per test;
the dense contractions are exact-fp32 fma chains whose summation order differs from ATen's, hence ~1e-5 relative.
"""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

if not torch.cuda.is_available():
    pytest.skip('needs a GPU', allow_module_level=True)

from gen_adversarial_amd import _lib as L   # noqa: E402

DEV = 'cuda:0'
ACTS = {0: lambda u: u, 1: F.silu, 2: F.elu, 3: F.relu, 4: F.leaky_relu}


def nhwc(t):   # NCHW cpu -> NHWC gpu
    return t.permute(0, 2, 3, 1).contiguous().to(DEV)


def nchw(t):   # NHWC gpu -> NCHW cpu
    return t.permute(0, 3, 1, 2).contiguous().cpu()


def g(*shape, seed=0, scale=1.0):
    gen = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=gen) * scale


def close(a, b, tol, what=''):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    err = (a - b).abs().max().item()
    ref = b.abs().max().item() + 1e-12
    assert err <= tol * max(1.0, ref), f'{what}: max err {err:.3e} (ref max {ref:.3e})'


def fwd_w(w):
    return w.permute(0, 2, 3, 1).reshape(w.shape[0], -1).contiguous().to(DEV)


def bwd_w(w):
    return w.flip(2, 3).permute(1, 2, 3, 0).reshape(w.shape[1], -1).contiguous().to(DEV)


def run_conv(x, w, y, K, sn=1, sd=1, pad=0, tile=0, **kw):
    d = L.ConvDesc()
    d.x, d.ldx, d.C1 = x.data_ptr(), x.shape[3], x.shape[3]
    d.w, d.y, d.ldy, d.Cout = w.data_ptr(), y.data_ptr(), y.shape[3], y.shape[3]
    d.N, d.Hi, d.Wi = x.shape[0], x.shape[1], x.shape[2]
    d.Ho, d.Wo = y.shape[1], y.shape[2]
    d.KH = d.KW = K
    d.sn, d.sd, d.pad, d.tile = sn, sd, pad, tile
    keep = []
    for k, v in kw.items():
        if isinstance(v, torch.Tensor):
            keep.append(v)
            setattr(d, k, v.data_ptr())
        else:
            setattr(d, k, v)
    L.run(d)
    torch.cuda.synchronize()
    return d


CONV_CASES = [
    # N, H, Cin, Cout, K, stride, pro_act, affine, tile
    (2, 8, 32, 40, 3, 1, 1, True, 0),
    (2, 8, 32, 130, 3, 2, 1, True, 1),
    (3, 6, 3, 16, 3, 1, 0, True, 0),       # stem-like, non-vectorised path
    (2, 8, 20, 100, 3, 1, 2, False, 2),
    (2, 16, 48, 64, 1, 1, 0, False, 3),
    (2, 8, 160, 33, 1, 2, 1, False, 4),
    (5, 4, 8, 8, 3, 1, 1, True, 1),
    (2, 4, 6, 12, 1, 1, 3, False, 0),      # Cin % 4 != 0
    (1, 32, 36, 257, 3, 1, 1, True, 1),
    (3, 32, 8, 16, 3, 2, 1, True, 0),
    (3, 32, 8, 16, 3, 2, 1, True, 1),
    (3, 32, 8, 16, 3, 2, 1, True, 3),
    (3, 32, 8, 16, 3, 1, 1, True, 4),
    (3, 32, 8, 16, 3, 2, 0, False, 4),
    (1, 32, 8, 16, 3, 2, 0, False, 4),
    (1, 16, 8, 16, 3, 2, 0, False, 4),
    (1, 16, 32, 16, 3, 2, 0, False, 4),
    (8, 16, 16, 32, 3, 1, 3, False, 0),
    (8, 32, 8, 16, 3, 1, 3, False, 0),
    (8, 64, 3, 8, 3, 1, 0, True, 0),
]


@pytest.mark.parametrize('N,H,Cin,Cout,K,st,act,affine,tile', CONV_CASES)
def test_conv_forward_and_transpose(N, H, Cin, Cout, K, st, act, affine, tile):
    pad = K // 2
    x = g(N, Cin, H, H, seed=1)
    w = g(Cout, Cin, K, K, seed=2, scale=1.0 / np.sqrt(Cin * K * K))
    b = g(Cout, seed=3)
    s = (torch.rand(Cin, generator=torch.Generator().manual_seed(4)) + 0.5) if affine else None
    t = g(Cin, seed=5, scale=0.3) if affine else None
    xr = x.clone().requires_grad_(True)
    u = xr * s.view(1, -1, 1, 1) + t.view(1, -1, 1, 1) if affine else xr
    ref = F.conv2d(ACTS[act](u), w, b, stride=st, padding=pad)
    Ho = ref.shape[2]
    y = torch.empty(N, Ho, Ho, Cout, device=DEV)
    kw = dict(bias=b.to(DEV), pro_act=act)
    if affine:
        kw.update(pro_scale=s.to(DEV), pro_shift=t.to(DEV))
    xd = nhwc(x)
    run_conv(xd, fwd_w(w), y, K, sn=st, pad=pad, tile=tile, **kw)
    close(nchw(y), ref, 2e-5, 'conv fwd')

    # backward-to-input with the act' epilogue and an addend (= what an identity skip contributes)
    cot = g(*ref.shape, seed=6)
    (gx,) = torch.autograd.grad((ref * cot).sum(), [xr])
    extra = g(N, Cin, H, H, seed=7)
    dx = torch.empty(N, H, H, Cin, device=DEV)
    kw = dict(dact_x=xd, dact_act=act, addend=nhwc(extra), ldadd=Cin, lddact=Cin)
    if affine:
        kw.update(dact_scale=s.to(DEV), dact_shift=t.to(DEV))
    run_conv(nhwc(cot), bwd_w(w), dx, K, sn=1, sd=st, pad=K - 1 - pad, tile=tile, **kw)
    close(nchw(dx), gx + extra, 2e-5, 'conv bwd')


def test_conv_dual_source_bcast_addend_and_accumulate():
    N, H, C1, C2, Cout = 3, 4, 16, 6, 24
    x1, x2 = g(N, C1, H, H, seed=1), g(N, C2, H, H, seed=2)
    w = g(Cout, C1 + C2, 1, 1, seed=3, scale=0.2)
    add = g(1, Cout, H, H, seed=4)
    ref = F.conv2d(torch.cat([x1, x2], 1), w) + add
    y = torch.empty(N, H, H, Cout, device=DEV)
    x2d = nhwc(x2)
    run_conv(nhwc(x1), fwd_w(w), y, 1, x2=x2d, ldx2=C2, C2=C2, addend=nhwc(add), ldadd=Cout, addend_bcast_n=1)
    close(nchw(y), ref, 2e-5, 'dual source')
    # in-place accumulation (addend aliases y) plus second addend
    a2 = g(N, Cout, H, H, seed=5)
    run_conv(nhwc(x1), fwd_w(w), y, 1, x2=x2d, ldx2=C2, C2=C2, addend=y, ldadd=Cout, addend2=nhwc(a2), ldadd2=Cout)
    close(nchw(y), ref + F.conv2d(torch.cat([x1, x2], 1), w) + a2, 2e-5, 'accumulate')


@pytest.mark.parametrize('N,H,Cin,Cout,K,splits,tile', [(2, 4, 512, 20, 3, 8, 4), (4, 1, 1000, 100, 1, 16, 3),
                                                        (2, 8, 64, 6, 3, 4, 0), (1, 4, 96, 40, 1, 32, 0)])
def test_conv_split_k(N, H, Cin, Cout, K, splits, tile):
    pad = K // 2
    x = g(N, Cin, H, H, seed=1)
    w = g(Cout, Cin, K, K, seed=2, scale=1.0 / np.sqrt(Cin * K * K))
    b = g(Cout, seed=3)
    add = g(N, Cout, H, H, seed=4)
    ref = F.conv2d(F.silu(x), w, b, padding=pad) + add
    y = nhwc(add)                                           # accumulate in place: addend aliases y
    ws = torch.empty(splits * N * H * H * Cout, device=DEV)
    run_conv(nhwc(x), fwd_w(w), y, K, pad=pad, tile=tile, bias=b.to(DEV), pro_act=1, addend=y, ldadd=Cout,
             splits=splits, ws=ws, ws_floats=ws.numel())
    close(nchw(y), ref, 2e-5, 'split-K')
    d = L.ConvDesc()
    d.x = d.w = d.y = 16
    d.N = d.Hi = d.Wi = d.Ho = d.Wo = d.C1 = d.Cout = d.KH = d.KW = d.sn = d.sd = 1
    d.ldx = d.ldy = 1
    d.splits = 2                                            # no workspace
    assert L.lib.ga_conv2d(C.byref(d), None) == -1


@pytest.mark.parametrize('N,H,Cin,Cout,K,act,tile', [(2, 8, 32, 40, 3, 1, 0), (2, 16, 64, 128, 3, 1, 1), (4, 8, 128, 64, 1, 0, 2),
                                                     (2, 4, 256, 256, 3, 2, 3), (3, 8, 40, 24, 1, 3, 4), (1, 16, 8, 16, 3, 1, 0)])
def test_conv_bf16x3(N, H, Cin, Cout, K, act, tile):
    """split-bf16 contraction (w_hi/w_lo given): same op, ~2e-5 relative per product; asserted at 2e-4 of the output
    scale, forward and transpose."""
    pad = K // 2
    x = g(N, Cin, H, H, seed=1)
    w = g(Cout, Cin, K, K, seed=2, scale=1.0 / np.sqrt(Cin * K * K))
    b = g(Cout, seed=3)
    xr = x.clone().requires_grad_(True)
    ref = F.conv2d(ACTS[act](xr), w, b, padding=pad)
    y = torch.empty(N, H, H, Cout, device=DEV)
    wf = fwd_w(w)
    hi = wf.to(torch.bfloat16)
    lo = (wf - hi.float()).to(torch.bfloat16)
    hi2, lo2 = torch.empty_like(hi), torch.empty_like(lo)
    L.check(L.lib.ga_split_bf16(wf.data_ptr(), hi2.data_ptr(), lo2.data_ptr(), wf.numel(), None), 'split')
    torch.cuda.synchronize()
    assert torch.equal(hi, hi2) and torch.equal(lo, lo2)
    xd = nhwc(x)
    run_conv(xd, wf, y, K, pad=pad, tile=tile, bias=b.to(DEV), pro_act=act, w_hi=hi, w_lo=lo)
    close(nchw(y), ref, 2e-4, 'bf16x3 fwd')
    cot = g(*ref.shape, seed=6)
    (gx,) = torch.autograd.grad((ref * cot).sum(), [xr])
    wb = bwd_w(w)
    bh = wb.to(torch.bfloat16)
    bl = (wb - bh.float()).to(torch.bfloat16)
    dx = torch.empty(N, H, H, Cin, device=DEV)
    run_conv(nhwc(cot), wb, dx, K, pad=K - 1 - pad, tile=tile, dact_x=xd, dact_act=act, lddact=Cin, w_hi=bh, w_lo=bl)
    close(nchw(dx), gx, 2e-4, 'bf16x3 bwd')


@pytest.mark.parametrize('N,H,W,Cin,Cout,act,affine,tile,splits', [
    (3, 16, 16, 64, 128, 1, True, 5, 1),      # half an image per tile, BN-affine + SiLU prologue
    (5, 8, 8, 32, 96, 1, False, 6, 1),        # two images per tile, ragged last tile (5 images), Cout not a tile multiple
    (9, 4, 4, 64, 64, 0, False, 6, 2),        # eight 4x4 images per tile + tail, split-K over channel chunks
    (2, 32, 32, 32, 40, 3, False, 5, 1),      # 4 image rows per tile, ReLU
    (1, 64, 64, 32, 32, 2, False, 6, 1),      # 2 image rows per tile, ELU
    (2, 8, 16, 96, 128, 1, True, 5, 3),       # non-square image, three K splits
    (2, 6, 128, 32, 32, 0, False, 7, 1),      # wide image: the tile is one whole 128-pixel row (3 x 130 window), 128 x 32 tile
    (1, 5, 256, 64, 64, 0, False, 6, 2),      # row segments: two tiles per row, odd height, two K splits
    (2, 4, 384, 32, 96, 0, True, 5, 1),       # three segments per row, BN-affine prologue, Cout not a tile multiple
    (3, 8, 8, 64, 32, 0, False, 7, 1),        # 128 x 32 tile on small images
])
def test_conv_halo3(N, H, W, Cin, Cout, act, affine, tile, splits):
    """halo-staged 3x3 kernel (tile codes 5/6) against torch: forward with the prologue variants it instantiates, and as
    the backward-to-input convolution with the act' epilogue.  Same 2e-4 bar as the other split-bf16 kernel."""
    x = g(N, Cin, H, W, seed=1)
    w = g(Cout, Cin, 3, 3, seed=2, scale=1.0 / np.sqrt(Cin * 9))
    b = g(Cout, seed=3)
    sc = (torch.rand(Cin, generator=torch.Generator().manual_seed(4)) + 0.5) if affine else None
    sh = g(Cin, seed=5, scale=0.3) if affine else None
    xr = x.clone().requires_grad_(True)
    u = xr * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1) if affine else xr
    ref = F.conv2d(ACTS[act](u), w, b, padding=1)
    y = torch.full((N, H, W, Cout), float('nan'), device=DEV)
    wf = fwd_w(w)
    hi = wf.to(torch.bfloat16)
    lo = (wf - hi.float()).to(torch.bfloat16)
    ws = torch.empty(max(1, splits * N * H * W * Cout), device=DEV)
    kw = dict(bias=b.to(DEV), pro_act=act, w_hi=hi, w_lo=lo, splits=splits)
    if splits > 1:
        kw.update(ws=ws, ws_floats=ws.numel())
    if affine:
        kw.update(pro_scale=sc.to(DEV), pro_shift=sh.to(DEV))
    xd = nhwc(x)
    run_conv(xd, wf, y, 3, pad=1, tile=tile, **kw)
    close(nchw(y), ref, 2e-4, 'halo fwd')
    # backward-to-input of conv(act(x)): 3x3 over the cotangent with flipped weights, times act'(x)
    if not affine:
        cot = g(*ref.shape, seed=6)
        (gx,) = torch.autograd.grad((ref * cot).sum(), [xr])
        wb = bwd_w(w)
        bh = wb.to(torch.bfloat16)
        bl = (wb - bh.float()).to(torch.bfloat16)
        dx = torch.full((N, H, W, Cin), float('nan'), device=DEV)
        kb = dict(dact_x=xd, dact_act=act, lddact=Cin, w_hi=bh, w_lo=bl)
        if Cout % 32 == 0:
            run_conv(nhwc(cot), wb, dx, 3, pad=1, tile=tile, **kb)
            close(nchw(dx), gx, 2e-4, 'halo bwd')
        else:       # the halo kernel needs 32-channel chunks: the request is refused, not silently rerouted
            with pytest.raises(L.GaError):
                run_conv(nhwc(cot), wb, dx, 3, pad=1, tile=tile, **kb)


@pytest.mark.parametrize('N,H,W,Cin,Cout,act,mode,splits,dact', [
    (5, 16, 16, 128, 128, 1, 'affine', 1, 0),     # the 16^2 x 128 decoder / encoder shape, ragged last tiles
    (9, 8, 8, 64, 200, 0, 'none', 2, 1),          # two images per tile, Cout not a 128 multiple (zero-padded fragments), two K splits
    (17, 4, 4, 96, 128, 2, 'none', 3, 0),         # eight 4x4 images per tile + tail, three K splits, ELU
    (2, 32, 32, 32, 64, 3, 'none', 1, 1),         # 4 image rows per tile, one channel chunk, ReLU
    (3, 8, 8, 64, 128, 0, 'per_row', 1, 0),       # the SE-gate prologue of conv2^T
    (1, 2, 64, 32, 32, 4, 'none', 1, 0),          # 64-wide rows, LeakyReLU
    (96, 16, 16, 128, 128, 1, 'affine', 1, 0),    # 192 workgroups in flight: the prologue table is staged by all threads of a
    (160, 8, 8, 256, 256, 0, 'per_row', 1, 1),    # workgroup and read by all (a missing barrier showed only at this scale)
])
def test_conv_halo3_fragment_weights(N, H, W, Cin, Cout, act, mode, splits, dact):
    """tile 8 (weights read as ready MFMA fragments from ga_conv_desc.w_frag, engine_core.WeightStore.frag3) does the same
    arithmetic in the same order as tile 5: the two outputs are equal bit for bit, and both meet the split-bf16 bar against
    torch.  Without w_frag the request is refused."""
    from gen_adversarial_amd.engine_core import WeightStore
    x = g(N, Cin, H, W, seed=1)
    w = g(Cout, Cin, 3, 3, seed=2, scale=1.0 / np.sqrt(Cin * 9))
    b = g(Cout, seed=3)
    u = x
    kw = dict(bias=b.to(DEV), pro_act=act, splits=splits)
    if mode == 'affine':
        sc, sh = torch.rand(Cin, generator=torch.Generator().manual_seed(4)) + 0.5, g(Cin, seed=5, scale=0.3)
        u = x * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)
        kw.update(pro_scale=sc.to(DEV), pro_shift=sh.to(DEV))
    elif mode == 'per_row':
        sc, sh = torch.rand(N, Cin, generator=torch.Generator().manual_seed(4)) + 0.5, g(N, Cin, seed=5, scale=0.3)
        u = x * sc.view(N, Cin, 1, 1) + sh.view(N, Cin, 1, 1)
        kw.update(pro_scale=sc.to(DEV), pro_shift=sh.to(DEV), pro_per_row=1)
    ref = F.conv2d(ACTS[act](u), w, b, padding=1)
    ws_store = WeightStore(torch.device(DEV))
    wf = fwd_w(w)
    hi, lo = ws_store.split(wf)
    frag = ws_store.frag3(wf)
    kw.update(w_hi=hi, w_lo=lo)
    if dact:                     # the act' epilogue of a backward-to-input launch
        dx = g(N, Cout, H, W, seed=7)
        kw.update(dact_x=nhwc(dx), dact_act=1, lddact=Cout)
        sg = torch.sigmoid(dx)
        ref = ref * (sg * (1 + dx * (1 - sg)))
    if splits > 1:
        ws = torch.empty(splits * N * H * W * Cout, device=DEV)
        kw.update(ws=ws, ws_floats=ws.numel())
    xd = nhwc(x)
    y5 = torch.full((N, H, W, Cout), float('nan'), device=DEV)
    y8 = torch.full((N, H, W, Cout), float('nan'), device=DEV)
    run_conv(xd, wf, y5, 3, pad=1, tile=5, **kw)
    with pytest.raises(L.GaError):
        run_conv(xd, wf, y8, 3, pad=1, tile=8, **kw)
    assert torch.isnan(y8).all(), 'a refused launch wrote its output'
    run_conv(xd, wf, y8, 3, pad=1, tile=8, w_frag=frag, **kw)
    close(nchw(y8), ref, 2e-4, 'fragment-weight halo kernel vs torch')
    assert torch.equal(y5, y8), f'tile 8 differs from tile 5: max {float((y5 - y8).abs().max()):.3e}'


@pytest.mark.parametrize('N,H,W,Cout,act,mode,extra,Cin', [
    (3, 64, 64, 32, 1, 'none', 'none', 32),       # NVAE pre / post-processing cells: 32 -> 32 at 64 x 64, SiLU prologue
    (2, 64, 64, 32, 1, 'affine', 'none', 32),     # BatchNorm affine + SiLU (conv1 of an encoder cell)
    (5, 8, 16, 32, 0, 'per_row', 'dact', 32),     # one tile per image, the SE-gate prologue and the act' epilogue of a backward conv
    (2, 16, 48, 104, 2, 'none', 'none', 32),      # to_logits: ELU prologue, 104 output channels = 4 channel tiles, the last one 8 wide
    (1, 24, 32, 8, 3, 'none', 'addend', 32),      # 8 output channels (an image-pitch output), ReLU, identity-skip addend
    (70, 32, 32, 32, 4, 'none', 'dact', 32),      # 560 tiles over 512 workgroup slots: runs of 2 tiles and a short last run, LeakyReLU
    (9, 32, 32, 64, 0, 'affine', 'addend', 32),   # two channel tiles, PReLU-free affine without activation
    (3, 32, 32, 64, 1, 'affine', 'none', 64),     # 64 input channels (one workgroup per CU): 64 -> 64 at 32 x 32, BatchNorm + SiLU
    (5, 8, 16, 64, 0, 'per_row', 'dact', 64),     # ... the SE-gate prologue and the act' epilogue
    (40, 32, 32, 128, 0, 'none', 'addend', 64),   # ... 4 channel tiles, 640 tiles over 64 runs of 10
    (2, 24, 48, 32, 4, 'none', 'none', 64),       # ... LeakyReLU, rows and columns that are not powers of two
])
def test_conv_thin3_persistent_kernel(N, H, W, Cout, act, mode, extra, Cin):
    """tile 11 (csrc/conv_thin3.hip: persistent workgroups on 8 x 16 pixel tiles, the 32-channel layer's weights resident in LDS, the
    epilogue through a 2-D row map) does tile 7's arithmetic in tile 7's order: outputs equal bit for bit, both at the split-bf16 bar
    against torch; shapes it does not take and a missing fragment copy are refused without writing."""
    from gen_adversarial_amd.engine_core import WeightStore
    x = g(N, Cin, H, W, seed=1)
    w = g(Cout, Cin, 3, 3, seed=2, scale=1.0 / np.sqrt(Cin * 9))
    b = g(Cout, seed=3)
    u = x
    kw = dict(bias=b.to(DEV), pro_act=act)
    if mode == 'affine':
        sc, sh = torch.rand(Cin, generator=torch.Generator().manual_seed(4)) + 0.5, g(Cin, seed=5, scale=0.3)
        u = x * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)
        kw.update(pro_scale=sc.to(DEV), pro_shift=sh.to(DEV))
    elif mode == 'per_row':
        sc, sh = torch.rand(N, Cin, generator=torch.Generator().manual_seed(4)) + 0.5, g(N, Cin, seed=5, scale=0.3)
        u = x * sc.view(N, Cin, 1, 1) + sh.view(N, Cin, 1, 1)
        kw.update(pro_scale=sc.to(DEV), pro_shift=sh.to(DEV), pro_per_row=1)
    ref = F.conv2d(ACTS[act](u), w, b, padding=1)
    st = WeightStore(torch.device(DEV))
    wf = fwd_w(w)
    hi, lo = st.split(wf)
    kw.update(w_hi=hi, w_lo=lo)
    if extra == 'dact':
        dx = g(N, Cout, H, W, seed=7)
        kw.update(dact_x=nhwc(dx), dact_act=1, lddact=Cout)
        sg = torch.sigmoid(dx)
        ref = ref * (sg * (1 + dx * (1 - sg)))
    elif extra == 'addend':
        ad = g(N, Cout, H, W, seed=8)
        kw.update(addend=nhwc(ad), ldadd=Cout)
        ref = ref + ad
    xd = nhwc(x)
    y7 = torch.full((N, H, W, Cout), float('nan'), device=DEV)
    y11 = torch.full((N, H, W, Cout), float('nan'), device=DEV)
    run_conv(xd, wf, y7, 3, pad=1, tile=7 if 128 % W == 0 else 4, **kw)
    with pytest.raises(L.GaError):
        run_conv(xd, wf, y11, 3, pad=1, tile=11, **kw)             # no fragment copy
    assert torch.isnan(y11).all(), 'a refused launch wrote its output'
    run_conv(xd, wf, y11, 3, pad=1, tile=11, w_frag=st.frag_thin(wf), **kw)
    close(nchw(y11), ref, 2e-4, 'persistent thin 3x3 kernel vs torch')
    if Cin == 64 and 128 % W:          # the comparison kernel is the generic one (tile 4), which sums the two 32-channel chunks tap by tap
        assert float((y7 - y11).abs().max()) < 1e-5
    else:
        assert torch.equal(y7, y11), f'tile 11 differs from tile 7: max {float((y7 - y11).abs().max()):.3e}'


def test_conv_thin3_refuses_other_shapes():
    from gen_adversarial_amd.engine_core import WeightStore
    st = WeightStore(torch.device(DEV))
    for (Cin, H, W) in ((96, 16, 16), (32, 12, 16), (32, 8, 24), (64, 8, 24)):
        x = g(1, Cin, H, W, seed=1)
        w = g(32, Cin, 3, 3, seed=2)
        wf = fwd_w(w)
        hi, lo = st.split(wf)
        y = torch.full((1, H, W, 32), float('nan'), device=DEV)
        with pytest.raises(L.GaError):
            run_conv(nhwc(x), wf, y, 3, pad=1, tile=11, w_hi=hi, w_lo=lo, w_frag=hi)
        assert torch.isnan(y).all()


def test_conv_halo3_per_row_prologue():
    """the SE-gate prologue of the encoder cells' conv2^T (per-(row, channel) scale and shift) on the halo kernel"""
    N, H, Cin, Cout = 5, 8, 64, 96
    x = g(N, Cin, H, H, seed=1)
    w = g(Cout, Cin, 3, 3, seed=2, scale=1.0 / np.sqrt(Cin * 9))
    sc = torch.rand(N, Cin, generator=torch.Generator().manual_seed(3)) + 0.5
    sh = g(N, Cin, seed=4, scale=0.3)
    ref = F.conv2d(x * sc.view(N, Cin, 1, 1) + sh.view(N, Cin, 1, 1), w, None, padding=1)
    wf = fwd_w(w)
    hi = wf.to(torch.bfloat16)
    lo = (wf - hi.float()).to(torch.bfloat16)
    for tile in (5, 6, 7):
        y = torch.full((N, H, H, Cout), float('nan'), device=DEV)
        run_conv(nhwc(x), wf, y, 3, pad=1, tile=tile, pro_scale=sc.to(DEV), pro_shift=sh.to(DEV), pro_per_row=1, w_hi=hi, w_lo=lo)
        close(nchw(y), ref, 2e-4, f'halo per-row prologue tile {tile}')
    # row segments of a wide image (StyleGAN2's modulated convs at 256^2 .. 1024^2): the style scale per (row, channel)
    N, H, W, Cin, Cout = 3, 3, 256, 32, 32
    x = g(N, Cin, H, W, seed=5)
    w = g(Cout, Cin, 3, 3, seed=6, scale=1.0 / np.sqrt(Cin * 9))
    sc = torch.rand(N, Cin, generator=torch.Generator().manual_seed(7)) + 0.5
    ref = F.conv2d(x * sc.view(N, Cin, 1, 1), w, None, padding=1)
    wf = fwd_w(w)
    hi = wf.to(torch.bfloat16)
    lo = (wf - hi.float()).to(torch.bfloat16)
    for tile in (6, 7):
        y = torch.full((N, H, W, Cout), float('nan'), device=DEV)
        run_conv(nhwc(x), wf, y, 3, pad=1, tile=tile, pro_scale=sc.to(DEV), pro_shift=torch.zeros(N, Cin, device=DEV), pro_per_row=1,
                 w_hi=hi, w_lo=lo)
        close(nchw(y), ref, 2e-4, f'halo per-row prologue, row segments, tile {tile}')


@pytest.mark.parametrize('k,bf3', [(3, True), (3, False), (1, True)])
def test_stride2_transpose_by_subpixel_convs(k, bf3):
    """backward-to-input of a stride-2 conv as four anchored stride-1 ga_conv2d launches + ga_interleave2 (with act' and
    accumulation), against autograd"""
    from gen_adversarial_amd.folding import subpixel_weights
    N, Cin, Cout, h = 3, 32, 64, 8
    x = g(N, Cin, 2 * h, 2 * h, seed=1)
    w = g(Cout, Cin, k, k, seed=2, scale=1.0 / np.sqrt(Cin * k * k))
    xr = x.clone().requires_grad_(True)
    y = F.conv2d(F.silu(xr), w, stride=2, padding=(k - 1) // 2)
    cot = g(*y.shape, seed=3)
    (gx,) = torch.autograd.grad((y * cot).sum(), [xr])
    prev = g(N, Cin, 2 * h, 2 * h, seed=4)                       # a gradient already accumulated in the target
    dy = nhwc(cot)
    il = L.Interleave2Desc()
    keep = []
    for (a, b), (wm, kh, kw) in subpixel_weights(w.double()).items():
        wd = wm.to(DEV)
        plane = torch.full((N, h, h, Cin), float('nan'), device=DEV)
        kwargs = {}
        if bf3:
            hi = wd.to(torch.bfloat16)
            kwargs = dict(w_hi=hi, w_lo=(wd - hi.float()).to(torch.bfloat16))
        d = L.ConvDesc()
        d.x, d.ldx, d.C1, d.w, d.y, d.ldy, d.Cout = dy.data_ptr(), Cout, Cout, wd.data_ptr(), plane.data_ptr(), Cin, Cin
        d.N, d.Hi, d.Wi, d.Ho, d.Wo, d.KH, d.KW, d.sn, d.sd, d.pad = N, h, h, h, h, kh, kw, 1, 1, 0
        for kk, v in kwargs.items():
            keep.append(v)
            setattr(d, kk, v.data_ptr())
        L.run(d)
        il.s[2 * a + b] = plane.data_ptr()
        keep += [wd, plane]
    out = nhwc(prev).clone()
    xd = nhwc(x)
    il.y, il.addend, il.dact_x, il.dact_act = out.data_ptr(), out.data_ptr(), xd.data_ptr(), L.GA_ACT_SILU
    il.N, il.H, il.W, il.C = N, 2 * h, 2 * h, Cin
    L.run(il)
    torch.cuda.synchronize()
    close(nchw(out), gx + prev, 2e-4 if bf3 else 2e-5, f'sub-pixel transpose k{k}')


def _fuzz_cases(n, seed):
    rs = np.random.RandomState(seed)
    cases = []
    while len(cases) < n:
        K = int(rs.choice([1, 1, 3, 3, 3, 5]))
        stride = int(rs.choice([1, 1, 1, 2]))
        c = dict(N=int(rs.randint(1, 5)), H=int(rs.choice([4, 6, 8, 12, 16])), W=int(rs.choice([4, 8, 10, 16])),
                 Cin=int(rs.choice([4, 8, 20, 32, 48, 64, 96])), Cout=int(rs.choice([4, 12, 32, 40, 64, 100, 160])),
                 K=K, stride=stride, act=int(rs.randint(0, 5)), aff=int(rs.choice([0, 0, 1, 2])), bf3=bool(rs.randint(0, 2)),
                 tile=int(rs.choice([0, 0, 1, 2, 3, 4, 5, 6])), addend=bool(rs.randint(0, 2)), dact=int(rs.choice([0, 0, 1, 3, 4])),
                 splits=int(rs.choice([1, 1, 1, 2, 3])), seed=int(rs.randint(1 << 30)))
        if c['H'] % stride or c['W'] % stride:
            continue
        cases.append(c)
    return cases


@pytest.mark.parametrize('c', _fuzz_cases(48, 2024), ids=lambda c: 'N{N}_{H}x{W}_{Cin}to{Cout}_k{K}s{stride}_a{act}f{aff}_b{bf3}_t{tile}_s{splits}'.format(**c))
def test_conv_randomised_descriptors(c):
    """seeded random convolution descriptors (non-square images, odd channel counts, every prologue / epilogue feature,
    tile request, split-K, both arithmetic paths) against the same op composed in torch; a request the library refuses
    (GA_E_UNSUPPORTED for a tile that cannot take the shape) must be refused without writing the output"""
    N, H, W, Cin, Cout, K, st = c['N'], c['H'], c['W'], c['Cin'], c['Cout'], c['K'], c['stride']
    gen = torch.Generator().manual_seed(c['seed'])
    x = torch.randn(N, Cin, H, W, generator=gen)
    w = torch.randn(Cout, Cin, K, K, generator=gen) / np.sqrt(Cin * K * K)
    b = torch.randn(Cout, generator=gen)
    Ho, Wo = H // st, W // st
    u = x
    kw = {}
    if c['aff'] == 1:
        sc, sh = torch.rand(Cin, generator=gen) + 0.5, torch.randn(Cin, generator=gen) * 0.3
        u = x * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)
        kw.update(pro_scale=sc.to(DEV), pro_shift=sh.to(DEV))
    elif c['aff'] == 2:
        sc, sh = torch.rand(N, Cin, generator=gen) + 0.5, torch.randn(N, Cin, generator=gen) * 0.3
        u = x * sc.view(N, Cin, 1, 1) + sh.view(N, Cin, 1, 1)
        kw.update(pro_scale=sc.to(DEV), pro_shift=sh.to(DEV), pro_per_row=1)
    acts = {0: lambda t: t, 1: F.silu, 2: F.elu, 3: F.relu, 4: lambda t: F.leaky_relu(t, 0.01)}
    ref = F.conv2d(acts[c['act']](u), w, b, stride=st, padding=K // 2)
    assert ref.shape[2:] == (Ho, Wo) or K // 2 * 2 + 1 != K
    if c['dact']:
        v = torch.randn(N, Cout, Ho, Wo, generator=gen)
        vr = v.clone().requires_grad_(True)
        (dv,) = torch.autograd.grad(acts[c['dact']](vr).sum(), [vr])
        ref = ref * dv
        kw.update(dact_x=nhwc(v), lddact=Cout, dact_act=c['dact'])
    if c['addend']:
        a = torch.randn(N, Cout, Ho, Wo, generator=gen)
        ref = ref + a
        kw.update(addend=nhwc(a), ldadd=Cout)
    wf = fwd_w(w)
    if c['bf3']:
        hi = wf.to(torch.bfloat16)
        kw.update(w_hi=hi, w_lo=(wf - hi.float()).to(torch.bfloat16))
    if c['splits'] > 1:
        ws = torch.empty(c['splits'] * N * Ho * Wo * Cout, device=DEV)
        kw.update(splits=c['splits'], ws=ws, ws_floats=ws.numel())
    y = torch.full((N, Ho, Wo, Cout), float('nan'), device=DEV)
    try:
        run_conv(nhwc(x), wf, y, K, sn=st, pad=K // 2, tile=c['tile'], bias=b.to(DEV), pro_act=c['act'], **kw)
    except L.GaError as e:
        assert 'UNSUPPORTED' in str(e) and c['tile'] in (5, 6), (c, str(e))
        assert torch.isnan(y).all()
        return
    close(nchw(y), ref, 3e-4 if c['bf3'] else 3e-5, str(c))


def test_conv_per_row_prologue():
    N, H, Cin, Cout = 4, 4, 16, 8
    x = g(N, Cin, H, H, seed=1)
    w = g(Cout, Cin, 3, 3, seed=2, scale=0.1)
    s, t = g(N, Cin, seed=3), g(N, Cin, seed=4)
    ref = F.conv2d(x * s.view(N, Cin, 1, 1) + t.view(N, Cin, 1, 1), w, padding=1)
    y = torch.empty(N, H, H, Cout, device=DEV)
    run_conv(nhwc(x), fwd_w(w), y, 3, pad=1, pro_scale=s.to(DEV), pro_shift=t.to(DEV), pro_per_row=1)
    close(nchw(y), ref, 2e-5, 'per-row prologue')


def test_conv_rejects_bad_descriptors():
    d = L.ConvDesc()
    assert L.lib.ga_conv2d(C.byref(d), None) == -1
    with pytest.raises(L.GaError):
        L.run(d)


@pytest.mark.parametrize('N,H,Cc,up', [(5, 4, 48, False), (37, 4, 100, False), (3, 8, 96, False), (2, 16, 32, False), (2, 32, 12, False),
                                       (3, 8, 48, True), (2, 40, 8, True), (9, 4, 36, True)])
def test_dwconv5(N, H, Cc, up):
    hs = H // 2 if up else H
    x = g(N, Cc, hs, hs, seed=1)
    w = g(Cc, 1, 5, 5, seed=2, scale=0.2)
    b = g(Cc, seed=3)
    xr = x.clone().requires_grad_(True)
    a = F.silu(xr)
    if up:
        a = F.interpolate(a, scale_factor=2, mode='nearest')
    ref = F.conv2d(a, w, b, padding=2, groups=Cc)
    y = torch.empty(N, H, H, Cc, device=DEV)
    d = L.DwDesc()
    xd, wd, bd = nhwc(x), w.reshape(Cc, 25).t().contiguous().to(DEV), b.to(DEV)
    d.x, d.w, d.bias, d.y = xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), y.data_ptr()
    d.N, d.H, d.W, d.C, d.pro_act, d.up2 = N, H, H, Cc, 1, int(up)
    L.run(d)
    close(nchw(y), ref, 1e-5, 'dw fwd')

    cot = g(*ref.shape, seed=4)
    (gx,) = torch.autograd.grad((ref * cot).sum(), [xr])
    dx = torch.empty(N, hs, hs, Cc, device=DEV)
    b2 = L.DwDesc()
    cd, wf = nhwc(cot), w.flip(2, 3).reshape(Cc, 25).t().contiguous().to(DEV)
    b2.x, b2.w, b2.dact_x, b2.y = cd.data_ptr(), wf.data_ptr(), xd.data_ptr(), dx.data_ptr()
    b2.N, b2.H, b2.W, b2.C, b2.dact_act, b2.pool2 = N, H, H, Cc, 1, int(up)
    L.run(b2)
    close(nchw(dx), gx, 1e-5, 'dw bwd')


@pytest.mark.parametrize('N,H,Cc', [(3, 4, 64), (2, 8, 8), (2, 16, 132)])
def test_se_chain(N, H, Cc):
    """squeeze -> excite -> merge, and the backward pieces, against autograd of the same math."""
    Hd = max(Cc // 16, 4)
    t = g(N, Cc, H, H, seed=1)
    skip = g(N, Cc, H, H, seed=2)
    w1, b1, w2, b2 = g(Hd, Cc, seed=3, scale=0.3), g(Hd, seed=4), g(Cc, Hd, seed=5, scale=0.5), g(Cc, seed=6)
    tr = t.clone().requires_grad_(True)
    m = tr.mean(dim=[2, 3])
    gate = torch.sigmoid(F.linear(F.relu(F.linear(m, w1, b1)), w2, b2))
    ref = skip + 0.1 * (tr * gate.view(N, Cc, 1, 1))

    td, sd_ = nhwc(t), nhwc(skip)
    md, hid, gd = (torch.empty(N, Cc, device=DEV), torch.empty(N, Hd, device=DEV), torch.empty(N, Cc, device=DEV))
    r = L.ReduceDesc()
    r.a, r.out, r.N, r.P, r.C, r.scale = td.data_ptr(), md.data_ptr(), N, H * H, Cc, 1.0 / (H * H)
    L.run(r)
    close(md, m, 1e-6, 'squeeze')
    W = [v.to(DEV) for v in (w1, b1, w2, b2)]
    e = L.SeExciteDesc()
    e.m, e.w1, e.b1, e.w2, e.b2 = md.data_ptr(), *(v.data_ptr() for v in W)
    e.hid, e.gate, e.N, e.C, e.Hd, e.P, e.res_scale = hid.data_ptr(), gd.data_ptr(), N, Cc, Hd, H * H, 0.1
    L.run(e)
    close(gd, gate, 1e-6, 'gate')
    out = torch.empty(N, H, H, Cc, device=DEV)
    a = L.SeApplyDesc()
    a.skip, a.t, a.gate, a.out = sd_.data_ptr(), td.data_ptr(), gd.data_ptr(), out.data_ptr()
    a.N, a.H, a.W, a.C, a.skip_mode, a.res_scale = N, H, H, Cc, 0, 0.1
    L.run(a)
    close(nchw(out), ref, 1e-6, 'merge')

    cot = g(N, Cc, H, H, seed=7)
    (gt,) = torch.autograd.grad((ref * cot).sum(), [tr])
    cd = nhwc(cot)
    dg, ps, pb = (torch.empty(N, Cc, device=DEV) for _ in range(3))
    r2 = L.ReduceDesc()
    r2.a, r2.b, r2.out, r2.N, r2.P, r2.C, r2.scale = cd.data_ptr(), td.data_ptr(), dg.data_ptr(), N, H * H, Cc, 0.1
    L.run(r2)
    e2 = L.SeExciteDesc()
    e2.w1, e2.b1, e2.w2, e2.b2 = (v.data_ptr() for v in W)
    e2.hid, e2.gate, e2.dgate, e2.pro_scale, e2.pro_shift = hid.data_ptr(), gd.data_ptr(), dg.data_ptr(), ps.data_ptr(), pb.data_ptr()
    e2.N, e2.C, e2.Hd, e2.P, e2.res_scale, e2.backward = N, Cc, Hd, H * H, 0.1, 1
    L.run(e2)
    torch.cuda.synchronize()
    dt = nchw(cd) * ps.cpu().view(N, Cc, 1, 1) + pb.cpu().view(N, Cc, 1, 1)
    close(dt, gt, 1e-6, 'se backward')

    # fused forms (reduction inside the excite kernel) must give the same numbers
    hid2, g2, ps2, pb2 = (torch.empty_like(hid), torch.empty_like(gd), torch.empty_like(ps), torch.empty_like(pb))
    f = L.SeExciteDesc()
    f.t, f.w1, f.b1, f.w2, f.b2 = td.data_ptr(), *(v.data_ptr() for v in W)
    f.hid, f.gate, f.N, f.C, f.Hd, f.P, f.res_scale = hid2.data_ptr(), g2.data_ptr(), N, Cc, Hd, H * H, 0.1
    L.run(f)
    close(g2, gate, 1e-6, 'fused gate')
    f.backward, f.dout, f.pro_scale, f.pro_shift = 1, cd.data_ptr(), ps2.data_ptr(), pb2.data_ptr()
    L.run(f)
    torch.cuda.synchronize()
    close(ps2, ps, 1e-6, 'fused ps')
    close(pb2, pb, 1e-6 , 'fused pb')

    # squeeze + excite + merge in one launch: bitwise the gate and the output of the two launches above
    hid3, g3, out3 = torch.empty_like(hid), torch.empty_like(gd), torch.full_like(out, float('nan'))
    m = L.SeExciteDesc()
    m.t, m.w1, m.b1, m.w2, m.b2 = td.data_ptr(), *(v.data_ptr() for v in W)
    m.hid, m.gate, m.N, m.C, m.Hd, m.P, m.res_scale = hid3.data_ptr(), g3.data_ptr(), N, Cc, Hd, H * H, 0.1
    m.skip, m.out = sd_.data_ptr(), out3.data_ptr()
    L.run(m)
    out2 = torch.empty_like(out)                        # the separate merge on the fused form's gate
    a.gate, a.out = g2.data_ptr(), out2.data_ptr()
    L.run(a)
    torch.cuda.synchronize()
    assert torch.equal(g3, g2) and torch.equal(out3, out2)
    m.skip = None
    L.run(m)
    torch.cuda.synchronize()
    close(out3, 0.1 * td * g2.view(N, 1, 1, Cc), 1e-6, 'merge without a skip')


@pytest.mark.parametrize('N,h,Cc', [(2, 4, 16), (3, 8, 8), (1, 16, 4), (2, 1, 4)])
def test_bilinear_skip(N, h, Cc):
    low = g(N, Cc, h, h, seed=1)
    t = g(N, Cc, 2 * h, 2 * h, seed=2)
    gate = torch.rand(N, Cc, generator=torch.Generator().manual_seed(3))
    lr = low.clone().requires_grad_(True)
    ref = F.interpolate(lr, scale_factor=2, mode='bilinear', align_corners=True) + 0.1 * t * gate.view(N, Cc, 1, 1)
    out = torch.empty(N, 2 * h, 2 * h, Cc, device=DEV)
    a = L.SeApplyDesc()
    ld, td, gd = nhwc(low), nhwc(t), gate.to(DEV)
    a.skip, a.t, a.gate, a.out = ld.data_ptr(), td.data_ptr(), gd.data_ptr(), out.data_ptr()
    a.N, a.H, a.W, a.C, a.skip_mode, a.res_scale = N, 2 * h, 2 * h, Cc, 1, 0.1
    L.run(a)
    close(nchw(out), ref, 1e-6, 'bilinear fwd')
    cot = g(*ref.shape, seed=4)
    (gl,) = torch.autograd.grad((ref * cot).sum(), [lr])
    dl = torch.empty(N, h, h, Cc, device=DEV)
    b = L.BilinearBwdDesc()
    cd = nhwc(cot)
    b.dhigh, b.dlow, b.N, b.h, b.w, b.C, b.accumulate = cd.data_ptr(), dl.data_ptr(), N, h, h, Cc, 0
    L.run(b)
    close(nchw(dl), gl, 1e-6, 'bilinear bwd')


@pytest.mark.parametrize('first,pitch', [(True, 0), (False, 0), (True, 8), (False, 8)])
def test_sampler(first, pitch):
    """pitch > 0: mu_q, z and their cotangents carry `pitch` channels per pixel (the engine pads the 20 latent channels to 24 with
    zeros); the pad channels are neither read nor written"""
    N, h, NL, alpha, temp = 3, 4, 6, 0.35, 0.6
    LD = pitch or NL
    mq = g(N, NL, h, h, seed=1, scale=3)
    p = None if first else g(N, 2 * NL, h, h, seed=2, scale=3)
    eps = g(N, NL, h, h, seed=3)
    mqr = mq.clone().requires_grad_(True)
    pr = None if first else p.clone().requires_grad_(True)
    sc = lambda v: torch.tanh(v / 5.0) * 5.0
    mp, ls = (torch.zeros_like(mq), torch.zeros_like(mq)) if first else (pr[:, :NL], pr[:, NL:])
    ref = (1 - alpha) * sc(mp + mqr) + alpha * (eps * (temp * torch.exp(sc(ls))) + sc(mp))

    def padded(t):          # NCHW cpu -> NHWC gpu with LD channels, the pad ones poisoned
        out = torch.full((N, h, h, LD), 7.0, device=DEV)
        out[..., :NL] = nhwc(t)
        return out
    z = torch.full((N, h, h, LD), -3.0, device=DEV)
    d = L.SamplerDesc()
    mqd, ed = padded(mq), eps.to(DEV)
    pd = None if first else nhwc(p)
    d.mu_q, d.ldq, d.eps, d.eps_nchw, d.z, d.ldz = mqd.data_ptr(), LD, ed.data_ptr(), 1, z.data_ptr(), pitch
    if not first:
        d.p, d.ldp = pd.data_ptr(), 2 * NL
    d.N, d.h, d.w, d.NL, d.alpha, d.one_minus_alpha, d.temp = N, h, h, NL, alpha, 1 - alpha, temp
    L.run(d)
    close(nchw(z[..., :NL]), ref, 1e-6, 'sampler fwd')
    assert bool((z[..., NL:] == -3.0).all())
    cot = g(*ref.shape, seed=4)
    grads = torch.autograd.grad((ref * cot).sum(), [mqr] if first else [mqr, pr])
    dmq = torch.full((N, h, h, LD), -5.0, device=DEV)
    dp = torch.empty(N, h, h, 2 * NL, device=DEV)
    cd = padded(cot)
    d.backward, d.dz, d.dmu_q = 1, cd.data_ptr(), dmq.data_ptr()
    if not first:
        d.dp = dp.data_ptr()
    L.run(d)
    close(nchw(dmq[..., :NL]), grads[0], 1e-6, 'sampler dmu_q')
    assert bool((dmq[..., NL:] == -5.0).all())
    if not first:
        close(nchw(dp), grads[1], 1e-6, 'sampler dp')
    if pitch:               # replicas sharing mu_q: per-row gradients with the same pitch
        rows = torch.full((N, h, h, LD), -9.0, device=DEV)
        d.q_rep, d.dmu_q, d.dmu_q_rows = 1, dmq.data_ptr(), rows.data_ptr()
        d.q_rep = 3 if N % 3 == 0 else 1
        mq1 = padded(mq[:N // d.q_rep])
        d.mu_q = mq1.data_ptr()
        L.run(d)
        torch.cuda.synchronize()
        assert bool((rows[..., NL:] == -9.0).all()) and torch.isfinite(rows[..., :NL]).all()


def test_dml_mean():
    from oracle.nvae_oracle import disc_mix_logistic_mean
    N, H = 2, 6
    lg = g(N, 100, H, H, seed=1, scale=1.5)
    lr = lg.clone().requires_grad_(True)
    ref = disc_mix_logistic_mean(lr, 10) * 0.5 + 0.5
    ld = nhwc(lg)
    o1, o2 = torch.empty(N, 3, H, H, device=DEV), torch.empty(N, H, H, 3, device=DEV)
    d = L.DmlDesc()
    d.logits, d.ld, d.nmix, d.img_nchw, d.img_nhwc, d.N, d.H, d.W = ld.data_ptr(), 100, 10, o1.data_ptr(), o2.data_ptr(), N, H, H
    L.run(d)
    close(o1, ref, 1e-6, 'dml nchw')
    close(nchw(o2), ref, 1e-6, 'dml nhwc')
    cot = g(N, 3, H, H, seed=2)
    (gl,) = torch.autograd.grad((ref * cot).sum(), [lr])
    dl = torch.empty(N, H, H, 100, device=DEV)
    cd = nhwc(cot)
    d.backward, d.dimg_nhwc, d.dlogits = 1, cd.data_ptr(), dl.data_ptr()
    L.run(d)
    close(nchw(dl), gl, 1e-6, 'dml bwd')


@pytest.mark.parametrize('N,H,Cc', [(2, 8, 12), (8, 32, 16), (8, 64, 8), (3, 16, 32)])
def test_maxpool(N, H, Cc):
    x = g(N, Cc, H, H, seed=1)
    x[0, 0, 0, 0] = x[0, 0, 0, 1] = x[0, 0, 1, 0] = 5.0       # tie: first in scan order wins
    xr = x.clone().requires_grad_(True)
    ref = F.max_pool2d(xr, 2, 2)
    xd = nhwc(x)
    y = torch.empty(N, H // 2, H // 2, Cc, device=DEV)
    d = L.MaxpoolDesc()
    d.x, d.y, d.N, d.H, d.W, d.C = xd.data_ptr(), y.data_ptr(), N, H, H, Cc
    L.run(d)
    close(nchw(y), ref, 0, 'maxpool fwd')
    cot = g(*ref.shape, seed=2)
    (gx,) = torch.autograd.grad((ref * cot).sum(), [xr])
    dx = torch.empty(N, H, H, Cc, device=DEV)
    cd = nhwc(cot)
    d.backward, d.dy, d.dx = 1, cd.data_ptr(), dx.data_ptr()
    L.run(d)
    close(nchw(dx), gx, 0, 'maxpool bwd')


@pytest.mark.parametrize('N,H,W,Cc', [(2, 8, 8, 8), (3, 16, 12, 20), (1, 6, 4, 4)])
def test_maxpool3s2(N, H, W, Cc):
    """3x3 / stride 2 / pad 1 max pool and its gather-form backward against torch (ties included: quantised inputs)"""
    x = (g(N, Cc, H, W, seed=1) * 2).round() / 2          # many exact ties -> exercises the first-maximum rule
    xr = x.clone().requires_grad_(True)
    ref = F.max_pool2d(xr, 3, 2, 1)
    cot = g(*ref.shape, seed=2)
    (gx,) = torch.autograd.grad((ref * cot).sum(), [xr])
    xd = nhwc(x)
    y = torch.empty(N, H // 2, W // 2, Cc, device=DEV)
    d = L.Maxpool3s2Desc()
    d.x, d.y, d.N, d.H, d.W, d.C, d.backward = xd.data_ptr(), y.data_ptr(), N, H, W, Cc, 0
    L.run(d)
    close(nchw(y), ref, 0, 'maxpool3s2 fwd')
    dy, dx = nhwc(cot), torch.full((N, H, W, Cc), float('nan'), device=DEV)
    b = L.Maxpool3s2Desc()
    b.x, b.dy, b.dx, b.N, b.H, b.W, b.C, b.backward = xd.data_ptr(), dy.data_ptr(), dx.data_ptr(), N, H, W, Cc, 1
    L.run(b)
    torch.cuda.synchronize()
    close(nchw(dx), gx, 1e-6, 'maxpool3s2 bwd')


def test_avgpool_act():
    N, P, Cc = 3, 20, 72
    x = g(N, P, Cc, seed=1)
    xr = x.clone().requires_grad_(True)
    ref = F.relu(xr).mean(dim=1)
    cot = g(N, Cc, seed=2)
    (gx,) = torch.autograd.grad((ref * cot).sum(), [xr])
    xd, y, dy, dx = x.to(DEV), torch.empty(N, Cc, device=DEV), cot.to(DEV), torch.empty(N, P, Cc, device=DEV)
    d = L.AvgpoolActDesc()
    d.x, d.y, d.N, d.P, d.C, d.act, d.backward = xd.data_ptr(), y.data_ptr(), N, P, Cc, L.GA_ACT_RELU, 0
    L.run(d)
    b = L.AvgpoolActDesc()
    b.x, b.dy, b.dx, b.N, b.P, b.C, b.act, b.backward = xd.data_ptr(), dy.data_ptr(), dx.data_ptr(), N, P, Cc, L.GA_ACT_RELU, 1
    L.run(b)
    torch.cuda.synchronize()
    close(y.cpu(), ref, 1e-6, 'avgpool fwd')
    close(dx.cpu(), gx, 1e-6, 'avgpool bwd')


@pytest.mark.parametrize('bf3,tile,K', [(False, 0, 3), (True, 1, 3), (True, 5, 3), (True, 2, 1)])
def test_conv_prelu_and_leaky_relu(bf3, tile, K):
    """nn.PReLU(C) as prologue (GA_CONV_PRO_PRELU) and its derivative as epilogue (GA_CONV_DACT_PRELU); nn.LeakyReLU()
    as GA_ACT_LRELU — e4e's bottleneck_IR_SE and GradualStyleBlock (encoding/helpers.py:112, encoder.py:41-46)"""
    N, H, Cin, Cout = 2, 16, 64, 96
    x = g(N, Cin, H, H, seed=1)
    w = g(Cout, Cin, K, K, seed=2, scale=1.0 / np.sqrt(Cin * K * K))
    slope = (torch.rand(Cin, generator=torch.Generator().manual_seed(3)) - 0.3)          # both signs
    wf = fwd_w(w)
    kw = {}
    if bf3:
        hi = wf.to(torch.bfloat16)
        kw = dict(w_hi=hi, w_lo=(wf - hi.float()).to(torch.bfloat16))
    tol = 2e-4 if bf3 else 2e-5
    for mode in ('prelu', 'lrelu'):
        xr = x.clone().requires_grad_(True)
        a = F.prelu(xr, slope) if mode == 'prelu' else F.leaky_relu(xr, 0.01)
        ref = F.conv2d(a, w, padding=K // 2)
        y = torch.full((N, H, H, Cout), float('nan'), device=DEV)
        sl = slope.to(DEV)
        if mode == 'prelu':
            run_conv(nhwc(x), wf, y, K, pad=K // 2, tile=tile, pro_scale=sl, pro_shift=sl, flags=L.GA_CONV_PRO_PRELU, **kw)
        else:
            run_conv(nhwc(x), wf, y, K, pad=K // 2, tile=tile, pro_act=L.GA_ACT_LRELU, **kw)
        close(nchw(y), ref, tol, f'{mode} prologue')
        cot = g(*ref.shape, seed=6)
        (gx,) = torch.autograd.grad((ref * cot).sum(), [xr])
        wb = bwd_w(w)
        kb = {}
        if bf3:
            bh = wb.to(torch.bfloat16)
            kb = dict(w_hi=bh, w_lo=(wb - bh.float()).to(torch.bfloat16))
        dx = torch.full((N, H, H, Cin), float('nan'), device=DEV)
        if mode == 'prelu':
            run_conv(nhwc(cot), wb, dx, K, pad=K // 2, tile=tile, dact_x=nhwc(x), lddact=Cin, dact_scale=sl, dact_shift=sl,
                     flags=L.GA_CONV_DACT_PRELU, **kb)
        else:
            run_conv(nhwc(cot), wb, dx, K, pad=K // 2, tile=tile, dact_x=nhwc(x), lddact=Cin, dact_act=L.GA_ACT_LRELU, **kb)
        close(nchw(dx), gx, tol, f'{mode} epilogue')


@pytest.mark.parametrize('Cout,splits', [(64, 1), (20, 1), (64, 2)])
def test_conv_residual_flags(Cout, splits):
    """GA_CONV_ADDEND_RELU (y = conv + relu(addend)) and GA_CONV_ADDEND_PRE_DACT (y = (conv + addend) * act'(u)) in the
    vector, scalar and split-K epilogues"""
    N, H, Cin = 2, 8, 32
    x = g(N, Cin, H, H, seed=1)
    w = g(Cout, Cin, 1, 1, seed=2, scale=0.2)
    a = g(N, Cout, H, H, seed=3)
    u = g(N, Cout, H, H, seed=4)
    conv = F.conv2d(x, w)
    ws = torch.empty(max(1, splits * N * H * H * Cout), device=DEV)
    kw = dict(splits=splits)
    if splits > 1:
        kw.update(ws=ws, ws_floats=ws.numel())
    y = torch.empty(N, H, H, Cout, device=DEV)
    run_conv(nhwc(x), fwd_w(w), y, 1, addend=nhwc(a), ldadd=Cout, flags=L.GA_CONV_ADDEND_RELU, **kw)
    close(nchw(y), conv + F.relu(a), 2e-5, 'addend relu')
    run_conv(nhwc(x), fwd_w(w), y, 1, addend=nhwc(a), ldadd=Cout, dact_x=nhwc(u), lddact=Cout, dact_act=3,
             flags=L.GA_CONV_ADDEND_PRE_DACT, **kw)
    close(nchw(y), (conv + a) * (u > 0).float(), 2e-5, 'addend before act\'')


def test_image_io():
    B, rep, H = 2, 3, 8
    N = B * rep
    x = torch.rand(B, 3, H, H, generator=torch.Generator().manual_seed(1))
    noise = g(N, 3, H, H, seed=2)
    coef = 2.0 / noise.flatten(1).norm(dim=1)
    xr = x.clone().requires_grad_(True)
    rows = xr.repeat_interleave(rep, dim=0)
    ref = (rows + noise * coef.view(-1, 1, 1, 1)).clamp(0, 1)
    y = torch.empty(N, H, H, 3, device=DEV)
    d = L.ImageIoDesc()
    xd, nd, cd = x.to(DEV), noise.to(DEV), coef.to(DEV)
    d.x_nchw, d.noise_nchw, d.noise_coef, d.y_nhwc = xd.data_ptr(), nd.data_ptr(), cd.data_ptr(), y.data_ptr()
    d.N, d.C, d.H, d.W, d.rep = N, 3, H, H, rep
    L.run(d)
    close(nchw(y), ref, 1e-7, 'image fwd')
    cot = g(N, 3, H, H, seed=3)
    (gx,) = torch.autograd.grad((ref * cot).sum(), [xr])
    dx = torch.empty(B, 3, H, H, device=DEV)
    ct = nhwc(cot)
    d.backward, d.dy_nhwc, d.dx_nchw = 1, ct.data_ptr(), dx.data_ptr()
    L.run(d)
    close(dx, gx, 1e-6, 'image bwd')


@pytest.mark.parametrize('stride,cg,act', [(1, 4, 3), (2, 8, 3), (1, 16, 0), (2, 32, 3), (1, 12, 3)])
def test_gconv(stride, cg, act):
    """grouped conv (ResNeXt conv2) forward with ReLU prologue, and as its own backward-to-input (stride 1) with act'"""
    from gen_adversarial_amd.folding import conv_fwd_layout, grouped_bwd_weights
    N, G, H = 2, 3, 8
    Cc = G * cg
    x = g(N, Cc, H, H, seed=1)
    w = g(Cc, cg, 3, 3, seed=2, scale=1.0 / np.sqrt(cg * 9))
    b = g(Cc, seed=3)
    xr = x.clone().requires_grad_(True)
    ref = F.conv2d(ACTS[act](xr), w, b, stride=stride, padding=1, groups=G)
    Ho = H // stride
    y = torch.full((N, Ho, Ho, Cc), float('nan'), device=DEV)
    xd, wd, bd = nhwc(x), conv_fwd_layout(w).to(DEV), b.to(DEV)
    d = L.GconvDesc()
    d.x, d.w, d.bias, d.y = xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), y.data_ptr()
    d.N, d.Hi, d.Wi, d.Ho, d.Wo, d.C, d.cg, d.KH, d.KW, d.stride, d.pad, d.pro_act = N, H, H, Ho, Ho, Cc, cg, 3, 3, stride, 1, act
    L.run(d)
    torch.cuda.synchronize()
    close(nchw(y), ref, 2e-5, 'gconv fwd')
    if stride == 1:
        cot = g(*ref.shape, seed=4)
        (gx,) = torch.autograd.grad((ref * cot).sum(), [xr])
        wb = conv_fwd_layout(grouped_bwd_weights(w, G)).to(DEV)
        dy, dx = nhwc(cot), torch.full((N, H, H, Cc), float('nan'), device=DEV)
        bdesc = L.GconvDesc()
        bdesc.x, bdesc.w, bdesc.y, bdesc.dact_x = dy.data_ptr(), wb.data_ptr(), dx.data_ptr(), xd.data_ptr()
        bdesc.N, bdesc.Hi, bdesc.Wi, bdesc.Ho, bdesc.Wo, bdesc.C, bdesc.cg = N, H, H, H, H, Cc, cg
        bdesc.KH, bdesc.KW, bdesc.stride, bdesc.pad, bdesc.dact_act = 3, 3, 1, 1, act
        L.run(bdesc)
        torch.cuda.synchronize()
        close(nchw(dx), gx, 2e-5, 'gconv bwd')


def test_image_io_space_to_depth():
    """s2d layout: pixel (h, w) channel c at [n, h/2, w/2, ((h&1)*2 + (w&1))*ld + c], pad channels zero; backward reads it back"""
    N, Cc, H, W, ld, rep = 4, 3, 6, 8, 8, 2
    x = torch.rand(N // rep, Cc, H, W, generator=torch.Generator().manual_seed(1))
    y = torch.full((N, H // 2, W // 2, 4 * ld), float('nan'), device=DEV)
    xd = x.to(DEV)
    d = L.ImageIoDesc()
    d.x_nchw, d.y_nhwc, d.N, d.C, d.H, d.W, d.rep, d.backward, d.ld, d.s2d = xd.data_ptr(), y.data_ptr(), N, Cc, H, W, rep, 0, ld, 1
    L.run(d)
    torch.cuda.synchronize()
    v = y.cpu().view(N, H // 2, W // 2, 2, 2, ld)
    assert torch.equal(v[..., Cc:], torch.zeros_like(v[..., Cc:]))
    img = v[..., :Cc].permute(0, 5, 1, 3, 2, 4).reshape(N, Cc, H, W)
    assert torch.equal(img, x.repeat_interleave(rep, dim=0))
    dy = torch.randn(N, H // 2, W // 2, 4 * ld, device=DEV)
    dx = torch.empty(N // rep, Cc, H, W, device=DEV)
    b = L.ImageIoDesc()
    b.x_nchw, b.dy_nhwc, b.dx_nchw, b.N, b.C, b.H, b.W, b.rep, b.backward, b.ld, b.s2d = (xd.data_ptr(), dy.data_ptr(), dx.data_ptr(),
                                                                                           N, Cc, H, W, rep, 1, ld, 1)
    L.run(b)
    torch.cuda.synchronize()
    g_img = dy.cpu().view(N, H // 2, W // 2, 2, 2, ld)[..., :Cc].permute(0, 5, 1, 3, 2, 4).reshape(N, Cc, H, W)
    ref = g_img.view(N // rep, rep, Cc, H, W).sum(dim=1)
    close(dx.cpu(), ref, 1e-6, 's2d image backward')


@pytest.mark.parametrize('H,k', [(64, 15), (32, 7), (16, 3), (20, 9)])
def test_gauss_blur_forward_and_adjoint(H, k):
    from oracle import defender_oracle as D
    x = torch.rand(2, 3, H, H, generator=torch.Generator().manual_seed(1))
    taps = D.gaussian_kernel1d(k)
    xr = x.clone().requires_grad_(True)
    p = k // 2
    y = F.pad(xr, (p, p, p, p), mode='reflect')
    y = F.conv2d(y, taps.view(1, 1, 1, k).expand(3, 1, 1, k), groups=3)
    ref = F.conv2d(y, taps.view(1, 1, k, 1).expand(3, 1, k, 1), groups=3)
    if H == 64:
        np.testing.assert_allclose(D.apply_gaussian_blur(x).numpy(), ref.detach().numpy(), atol=1e-6)   # k = 15 at 64x64
    xd, td = x.to(DEV), taps.to(DEV)
    out = torch.empty_like(xd)
    d = L.BlurDesc()
    d.x, d.y, d.taps, d.planes, d.H, d.W, d.k = xd.data_ptr(), out.data_ptr(), td.data_ptr(), 6, H, H, k
    L.run(d)
    close(out, ref, 1e-6, 'blur fwd')
    cot = g(2, 3, H, H, seed=2)
    (gx,) = torch.autograd.grad((ref * cot).sum(), [xr])
    cd = cot.to(DEV)
    dx = torch.empty_like(xd)
    d.x, d.y, d.backward = cd.data_ptr(), dx.data_ptr(), 1
    L.run(d)
    close(dx, gx, 1e-6, 'blur adjoint')


@pytest.mark.parametrize('H,W,k,radius', [(128, 128, 31, 0), (128, 128, 31, 12), (256, 256, 255, 12), (96, 200, 15, 0)])
def test_gauss_blur_large_planes_two_pass(H, W, k, radius):
    """the real image sizes of the cars (128 px, k = 31) and gender (256 px, k = 255) experiments
    (abstract_models.py:150-158): planes beyond LDS run as two passes through `tmp`; `radius` 12 skips taps below 1e-31 of
    the peak.  Forward and exact adjoint against the oracle's kornia restatement."""
    from oracle import defender_oracle as D
    x = torch.rand(2, 3, H, W, generator=torch.Generator().manual_seed(1))
    taps = D.gaussian_kernel1d(k)
    xr = x.clone().requires_grad_(True)
    p = k // 2
    y = F.pad(xr, (p, p, p, p), mode='reflect')
    y = F.conv2d(y, taps.view(1, 1, 1, k).expand(3, 1, 1, k), groups=3)
    ref = F.conv2d(y, taps.view(1, 1, k, 1).expand(3, 1, k, 1), groups=3)
    if H == W and k == D.blur_kernel_size(H):
        np.testing.assert_allclose(D.apply_gaussian_blur(x).numpy(), ref.detach().numpy(), atol=1e-6)
    xd, td = x.to(DEV), taps.to(DEV)
    out, tmp = torch.empty_like(xd), torch.empty_like(xd)
    d = L.BlurDesc()
    d.x, d.y, d.taps, d.planes, d.H, d.W, d.k = xd.data_ptr(), out.data_ptr(), td.data_ptr(), 6, H, W, k
    d.radius, d.tmp = radius, tmp.data_ptr()
    L.run(d)
    close(out, ref, 1e-6, 'blur fwd (two-pass)')
    cot = g(2, 3, H, W, seed=2)
    (gx,) = torch.autograd.grad((ref * cot).sum(), [xr])
    cd = cot.to(DEV)
    dx = torch.empty_like(xd)
    d.x, d.y, d.backward = cd.data_ptr(), dx.data_ptr(), 1
    L.run(d)
    close(dx, gx, 1e-6, 'blur adjoint (two-pass)')
    d.tmp = None                                              # a large plane without the intermediate buffer is rejected
    with pytest.raises(L.GaError):
        L.run(d)


def test_rep_sum_and_shared_addend():
    rows, rep, inner = 12, 4, 40
    x = g(rows, inner, seed=1).to(DEV)
    y = g(rows // rep, inner, seed=2).to(DEV)
    ref = y + x.view(rows // rep, rep, inner).sum(dim=1)
    r = L.RepSumDesc()
    r.x, r.y, r.rows, r.inner, r.rep, r.accumulate = x.data_ptr(), y.data_ptr(), rows, inner, rep, 1
    L.run(r)
    close(y, ref, 1e-6, 'rep_sum accumulate')
    r.accumulate = 0
    L.run(r)
    close(y, x.view(rows // rep, rep, inner).sum(dim=1), 1e-6, 'rep_sum')
    # conv whose addend has one row per group of `rep` output rows
    N, H, Cin, Cout = 6, 4, 16, 8
    xx, w, add = g(N, Cin, H, H, seed=3), g(Cout, Cin, 1, 1, seed=4, scale=0.2), g(N // 3, Cout, H, H, seed=5)
    refc = F.conv2d(xx, w) + add.repeat_interleave(3, dim=0)
    yy = torch.empty(N, H, H, Cout, device=DEV)
    run_conv(nhwc(xx), fwd_w(w), yy, 1, addend=nhwc(add), ldadd=Cout, addend_rep=3)
    close(nchw(yy), refc, 2e-5, 'addend_rep')


def test_plan_replay_matches_direct_calls():
    N, H, Cin, Cout = 2, 8, 16, 16
    x = nhwc(g(N, Cin, H, H, seed=1))
    w = fwd_w(g(Cout, Cin, 3, 3, seed=2, scale=0.1))
    y1 = torch.empty(N, H, H, Cout, device=DEV)
    y2 = torch.empty(N, H, H, Cout, device=DEV)
    d = run_conv(x, w, y1, 3, pad=1)
    d2 = L.ConvDesc.from_buffer_copy(d)
    d2.y = y2.data_ptr()
    p = L.Plan()
    p.add(d2, 'conv')
    ax = L.AxpbyDesc()
    ax.x, ax.y, ax.n, ax.alpha, ax.beta = y1.data_ptr(), y2.data_ptr(), y1.numel(), -1.0, 1.0
    p.add(ax, 'diff')
    p.run(torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert y2.abs().max().item() == 0.0
    ms, conv_ms, nconv = p.time(torch.cuda.current_stream().cuda_stream, iters=2, per_conv=True)
    assert ms > 0 and nconv == 1 and conv_ms > 0


@pytest.mark.parametrize('N,P,C,with_b', [(3, 5000, 32, True), (2, 16384, 4, True), (2, 4100, 96, False), (1, 70000, 64, True)])
def test_rowchan_reduce_two_stage(N, P, C, with_b):
    """long rows (StyleGAN2 style gradients): pixels split over workgroups through a workspace; deterministic"""
    gen = torch.Generator().manual_seed(N * 1000 + C)
    a = torch.randn(N, P, C, generator=gen).to(DEV)
    b = torch.randn(N, P, C, generator=gen).to(DEV) if with_b else None
    out = torch.zeros(N, C, device=DEV)
    ws = torch.zeros(64 * N * C, device=DEV)
    d = L.ReduceDesc()
    d.a, d.b, d.out, d.N, d.P, d.C, d.scale = a.data_ptr(), (b.data_ptr() if with_b else None), out.data_ptr(), N, P, C, 0.5
    d.ws, d.ws_floats = ws.data_ptr(), ws.numel()
    L.run(d, torch.cuda.current_stream().cuda_stream)
    ref = 0.5 * ((a * b) if with_b else a).double().sum(dim=1)
    assert (out.double() - ref).abs().max().item() < 1e-4 * max(1.0, ref.abs().max().item())
    first = out.clone()
    L.run(d, torch.cuda.current_stream().cuda_stream)
    assert torch.equal(first, out)
    d.ws, d.ws_floats = None, 0                              # single-stage path: same sum up to rounding
    L.run(d, torch.cuda.current_stream().cuda_stream)
    assert (out.double() - ref).abs().max().item() < 1e-4 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize('N,P,C,two_stage,with_skip', [(3, 5000, 32, True, True), (2, 16384, 4, True, False), (2, 300, 96, False, True),
                                                       (1, 70000, 64, True, True), (4, 64, 512, False, False)])
def test_rowchan_reduce_with_scaled_output(N, P, C, two_stage, with_skip):
    """StyledConv backward (generator.py:166-203 differentiated): sum_p a * b and scaled = a * gate[n,c] (+ skip, in place) from one
    pass; the sum equals the plain reduction bit for bit, the scaled output equals ga_se_apply's"""
    gen = torch.Generator().manual_seed(N * 1000 + C)
    a, b = torch.randn(N, P, C, generator=gen).to(DEV), torch.randn(N, P, C, generator=gen).to(DEV)
    gate = torch.randn(N, C, generator=gen).to(DEV)
    acc0 = torch.randn(N, P, C, generator=gen).to(DEV)
    out, out_ref = torch.zeros(N, C, device=DEV), torch.zeros(N, C, device=DEV)
    ws = torch.zeros(64 * N * C, device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    d = L.ReduceDesc()
    d.a, d.b, d.out, d.N, d.P, d.C, d.scale = a.data_ptr(), b.data_ptr(), out_ref.data_ptr(), N, P, C, 1.0
    if two_stage:
        d.ws, d.ws_floats = ws.data_ptr(), ws.numel()
    L.run(d, st)
    scaled = acc0.clone() if with_skip else torch.full((N, P, C), float('nan'), device=DEV)
    d.out, d.gate, d.scaled = out.data_ptr(), gate.data_ptr(), scaled.data_ptr()
    d.skip = scaled.data_ptr() if with_skip else None
    L.run(d, st)
    torch.cuda.synchronize()
    assert torch.equal(out, out_ref)
    ap = L.SeApplyDesc()
    ref = acc0.clone() if with_skip else torch.zeros(N, P, C, device=DEV)
    ap.skip = ref.data_ptr() if with_skip else None
    ap.t, ap.gate, ap.out = a.data_ptr(), gate.data_ptr(), ref.data_ptr()
    ap.N, ap.H, ap.W, ap.C, ap.skip_mode, ap.res_scale = N, 1, P, C, 0, 1.0
    L.run(ap, st)
    torch.cuda.synchronize()
    assert torch.equal(scaled, ref)
    want = a.double() * gate.double()[:, None, :] + (acc0.double() if with_skip else 0.0)
    assert (scaled.double() - want).abs().max().item() < 1e-5
    d.gate = None                                             # a scaled output without its gate is refused
    import ctypes
    assert L.lib.ga_rowchan_reduce(ctypes.byref(d), st) == -1           # GA_E_BADARG


@pytest.mark.parametrize('N,P,C,two_stage', [(2, 9000, 32, True), (3, 500, 64, False), (1, 65536, 32, True), (2, 4096, 512, True)])
def test_rowchan_reduce_forms_its_operand_from_a_4_lane_tensor(N, P, C, two_stage):
    """ToRGB backward: a[n,p,c] = sum_k a_w[c][k] a_src[n,p,k] is never stored; sum_p a * b and scaled = a * gate + skip equal the pass
    over a materialised `a` (same kernel, stored operand) to rounding of the 4-term dot product"""
    gen = torch.Generator().manual_seed(N * 100 + C)
    src = torch.randn(N, P, 4, generator=gen).to(DEV)
    src[..., 3] = 0.0                                                  # the padded lane of a 3-channel image
    w = (torch.randn(C, 4, generator=gen) * 0.5).to(DEV)
    b, gate, acc0 = torch.randn(N, P, C, generator=gen).to(DEV), torch.randn(N, C, generator=gen).to(DEV), torch.randn(N, P, C, generator=gen).to(DEV)
    a = torch.einsum('npk,ck->npc', src.double(), w.double())
    st = torch.cuda.current_stream().cuda_stream
    ws = torch.zeros(64 * N * C, device=DEV)
    out, scaled = torch.zeros(N, C, device=DEV), acc0.clone()
    d = L.ReduceDesc()
    d.a_src, d.a_w, d.b, d.out, d.N, d.P, d.C, d.scale = src.data_ptr(), w.data_ptr(), b.data_ptr(), out.data_ptr(), N, P, C, 1.0
    d.gate, d.skip, d.scaled = gate.data_ptr(), scaled.data_ptr(), scaled.data_ptr()
    if two_stage:
        d.ws, d.ws_floats = ws.data_ptr(), ws.numel()
    L.run(d, st)
    torch.cuda.synchronize()
    ref = (a * b.double()).sum(dim=1)
    assert (out.double() - ref).abs().max().item() < 1e-4 * max(1.0, ref.abs().max().item())
    want = a * gate.double()[:, None, :] + acc0.double()
    assert (scaled.double() - want).abs().max().item() < 1e-5
    first = out.clone()
    scaled.copy_(acc0)
    L.run(d, st)
    torch.cuda.synchronize()
    assert torch.equal(first, out)                                     # deterministic
    d.a_w = None
    import ctypes
    assert L.lib.ga_rowchan_reduce(ctypes.byref(d), st) == -1          # neither a nor (a_src, a_w)


@pytest.mark.parametrize('N,H,C,with_red,act', [(2, 16, 32, True, L.GA_ACT_FLRELU), (3, 8, 8, False, L.GA_ACT_FLRELU), (1, 64, 64, True, L.GA_ACT_NONE)])
def test_modout_reads_t_in_depth_to_space_form(N, H, C, with_red, act):
    """the up-sampling StyledConv keeps t in the parity conv's depth-to-space form: tail forward, its adjoint (dt in planes only) and the
    fused demodulation reduction equal the interleaved path bit for bit"""
    gen = torch.Generator().manual_seed(H * 100 + C)
    P, h2 = H * H, H // 2
    s2d = torch.randn(N, h2, h2, 4 * C, generator=gen).to(DEV)
    t = s2d.view(N, h2, h2, 2, 2, C).permute(0, 1, 3, 2, 4, 5).reshape(N, H, H, C).contiguous()     # pixel (2y+a, 2x+b) <- plane 2a+b
    scale = (1.0 + 0.2 * torch.randn(N, C, generator=gen)).to(DEV)
    add = torch.randn(P, C, generator=gen).to(DEV)
    dout = torch.randn(N, P, C, generator=gen).to(DEV)
    st = torch.cuda.current_stream().cuda_stream

    def desc(backward, planes):
        m = L.ModoutDesc()
        m.scale, m.add, m.N, m.P, m.C, m.act, m.backward = scale.data_ptr(), add.data_ptr(), N, P, C, act, backward
        if planes:
            m.W, m.ld_planes = H, 4 * C
            for i in range(4):
                m.t_planes[i] = s2d.data_ptr() + 4 * i * C
        else:
            m.t = t.data_ptr()
        return m
    o_ref, o = torch.zeros(N, P, C, device=DEV), torch.zeros(N, P, C, device=DEV)
    m = desc(0, False); m.out = o_ref.data_ptr(); L.run(m, st)
    m = desc(0, True); m.out = o.data_ptr(); L.run(m, st)
    torch.cuda.synchronize()
    assert torch.equal(o, o_ref)
    u = scale[:, None, :] * t.view(N, P, C) + add[None]
    want = torch.nn.functional.leaky_relu(u, 0.2) * 2 ** 0.5 if act == L.GA_ACT_FLRELU else u
    assert (o - want).abs().max().item() < 1e-5
    # backward: interleaved dt + planes (old form) against planes only, t from planes
    dt_ref, pl_ref, pl = torch.zeros(N, P, C, device=DEV), torch.zeros(N, h2, h2, 4 * C, device=DEV), torch.zeros(N, h2, h2, 4 * C, device=DEV)
    red_ref, red = torch.zeros(N, C, device=DEV), torch.zeros(N, C, device=DEV)
    ws = torch.zeros(256 * N * C, device=DEV)
    for planes, dt, plb, rd in ((False, dt_ref, pl_ref, red_ref), (True, None, pl, red)):
        b = desc(1, planes)
        b.dout = dout.data_ptr()
        b.dt = dt.data_ptr() if dt is not None else None
        b.W, b.ld_planes = H, 4 * C
        for i in range(4):
            b.dt_planes[i] = plb.data_ptr() + 4 * i * C
        if with_red:
            b.red, b.ws, b.ws_floats = rd.data_ptr(), ws.data_ptr(), ws.numel()
        L.run(b, st)
    torch.cuda.synchronize()
    assert torch.equal(pl, pl_ref) and torch.equal(red, red_ref)
    assert torch.equal(pl.view(N, h2, h2, 2, 2, C).permute(0, 1, 3, 2, 4, 5).reshape(N, P, C), dt_ref)
    b = desc(1, True)                                         # neither dt nor planes: refused
    b.dout = dout.data_ptr()
    import ctypes
    assert L.lib.ga_modout(ctypes.byref(b), st) == -1                   # GA_E_BADARG


def test_conv_beyond_2gb_runs_in_row_sub_batches():
    """an input past the fast loader's 31-bit byte offsets (StyleGAN2's 1024^2 maps at a few dozen rows) is convolved in
    sub-batches of rows: same numbers as one launch per row"""
    N, H, C, Co = 3, 1024, 192, 8                                   # 3 x 1024^2 x 192 x 4 B = 2.4 GB
    gen = torch.Generator(device=DEV).manual_seed(3)
    x = torch.randn(N, H, H, C, device=DEV, generator=gen)
    w = (torch.randn(Co, C, device=DEV, generator=gen) / C ** 0.5).contiguous()
    hi = w.to(torch.bfloat16)
    lo = (w - hi.float()).to(torch.bfloat16)
    sc = (1.0 + 0.1 * torch.randn(N, C, device=DEV, generator=gen)).contiguous()
    sh = torch.zeros(N, C, device=DEV)
    add = torch.randn(N, H, H, Co, device=DEV, generator=gen)

    def run(n0, n, out):
        d = L.ConvDesc()
        d.x, d.ldx, d.C1, d.w, d.w_hi, d.w_lo = x[n0:].data_ptr(), C, C, w.data_ptr(), hi.data_ptr(), lo.data_ptr()
        d.pro_scale, d.pro_shift, d.pro_per_row = sc[n0:].data_ptr(), sh[n0:].data_ptr(), 1
        d.y, d.ldy, d.Cout, d.addend, d.ldadd = out[n0:].data_ptr(), Co, Co, add[n0:].data_ptr(), Co
        d.N, d.Hi, d.Wi, d.Ho, d.Wo, d.KH, d.KW, d.sn, d.sd, d.pad = n, H, H, H, H, 1, 1, 1, 1, 0
        L.run(d, torch.cuda.current_stream().cuda_stream)

    whole, rows = torch.zeros(N, H, H, Co, device=DEV), torch.zeros(N, H, H, Co, device=DEV)
    run(0, N, whole)
    for n in range(N):
        run(n, 1, rows)
    assert torch.equal(whole, rows)
    ref = torch.einsum('nhc,oc->nho', (x[:, 500, :64] * sc[:, None, :]).double(), w.double()) + add[:, 500, :64].double()
    assert (whole[:, 500, :64].double() - ref).abs().max().item() < 1e-3


def test_conv_sub_batches_with_every_per_row_operand():
    """the row sub-batch path of ga_conv2d (taken above 2 GB; forced here with ga_debug_set_conv_row_limit) with a per-row
    prologue, an addend shared by `addend_rep` consecutive rows (EoT replicas reading one encoder feature map), a second
    addend and an act' source: bitwise equal to the single launch"""
    N, H, C, Co, rep = 12, 16, 64, 32, 3
    gen = torch.Generator(device=DEV).manual_seed(5)
    x = torch.randn(N, H, H, C, device=DEV, generator=gen)
    w = (torch.randn(Co, 9 * C, device=DEV, generator=gen) / (9 * C) ** 0.5).contiguous()
    hi = w.to(torch.bfloat16)
    lo = (w - hi.float()).to(torch.bfloat16)
    sc = (1.0 + 0.1 * torch.randn(N, C, device=DEV, generator=gen)).contiguous()
    sh = (0.1 * torch.randn(N, C, device=DEV, generator=gen)).contiguous()
    add = torch.randn(N // rep, H, H, Co, device=DEV, generator=gen)
    add2 = torch.randn(N, H, H, Co, device=DEV, generator=gen)
    dact = torch.randn(N, H, H, Co, device=DEV, generator=gen)

    def run(out, precise):
        d = L.ConvDesc()
        d.x, d.ldx, d.C1, d.w = x.data_ptr(), C, C, w.data_ptr()
        if not precise:
            d.w_hi, d.w_lo = hi.data_ptr(), lo.data_ptr()
        d.pro_scale, d.pro_shift, d.pro_per_row = sc.data_ptr(), sh.data_ptr(), 1
        d.y, d.ldy, d.Cout = out.data_ptr(), Co, Co
        d.addend, d.ldadd, d.addend_rep = add.data_ptr(), Co, rep
        d.addend2, d.ldadd2 = add2.data_ptr(), Co
        d.dact_x, d.lddact, d.dact_act = dact.data_ptr(), Co, L.GA_ACT_SILU
        d.N, d.Hi, d.Wi, d.Ho, d.Wo, d.KH, d.KW, d.sn, d.sd, d.pad = N, H, H, H, H, 3, 3, 1, 1, 1
        L.run(d, torch.cuda.current_stream().cuda_stream)

    for precise in (True, False):
        whole, parts = torch.zeros(N, H, H, Co, device=DEV), torch.zeros(N, H, H, Co, device=DEV)
        run(whole, precise)
        row_bytes = H * H * C * 4
        old = L.lib.ga_debug_set_conv_row_limit(4 * row_bytes + 1)          # sub = 4 rows -> rounded down to 3 (a multiple of rep)
        try:
            run(parts, precise)
        finally:
            L.lib.ga_debug_set_conv_row_limit(old)
        assert torch.equal(whole, parts)
    # and against plain torch (fp32 path)
    xs = x * sc[:, None, None, :] + sh[:, None, None, :]
    ref = F.conv2d(xs.permute(0, 3, 1, 2), w.view(Co, 3, 3, C).permute(0, 3, 1, 2), padding=1).permute(0, 2, 3, 1)
    sg = torch.sigmoid(dact)
    ref = ref * (sg * (1 + dact * (1 - sg))) + add.repeat_interleave(rep, dim=0) + add2
    close(whole, ref.cpu(), 1e-3, 'sub-batched conv vs torch')


def test_halo_tiles_refuse_unsupported_shapes():
    """explicit halo tile codes are never silently rerouted: 8-channel image convs, 1x1 convs and wide images with a prologue
    activation that has no row-segment instantiation are refused"""
    for (cin, k, w_, act) in ((8, 3, 64, 0), (32, 1, 64, 0), (32, 3, 256, 1)):
        x = torch.zeros(1, 4, w_, cin, device=DEV)
        w = torch.zeros(32, k * k * cin, device=DEV)
        hi = w.to(torch.bfloat16)
        y = torch.zeros(1, 4, w_, 32, device=DEV)
        for tile in (5, 6, 7):
            with pytest.raises(L.GaError):
                run_conv(x, w, y, k, pad=k // 2, tile=tile, w_hi=hi, w_lo=hi, pro_act=act)
