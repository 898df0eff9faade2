"""
GPU parity of the StyleGAN2 modulated convolution (first slice; StyleGan_E4E/stylegan2/
generator.py:108-290): output, d/dx and d/dw_latent of
  (a) the bare ModulatedConv2d against the golden produced by the reference's own module (3x3 demodulated and 1x1 plain),
  (b) StyledConv (noise + fused leaky ReLU) and ToRGB's conv + bias against the CPU oracle, chained (StyledConv -> ToRGB, the
      latent feeding both: gradient accumulation into x.g / w.g), at a generator-like width (512 channels, 16x16).
Tolerance 1e-3 absolute relative to the tensor's scale (BASELINE.json north_star); fp32 kernels ~1e-5.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

if not torch.cuda.is_available():
    pytest.skip('needs a GPU', allow_module_level=True)

from gen_adversarial_amd.engine import Engine   # noqa: E402
from gen_adversarial_amd.engine_core import Act   # noqa: E402
from gen_adversarial_amd.stylegan_spec import StyledConvSpec, init_styled_conv_state_dict   # noqa: E402

DEV = 'cuda:0'


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous().to(DEV)


def nchw(t, c=None):
    t = t.permute(0, 3, 1, 2)
    return (t if c is None else t[:, :c]).cpu()


def close(got, ref, tol, what):
    e, s = (got - ref).abs().max().item(), max(1.0, ref.abs().max().item())
    print(f'   {what}: err {e:.2e} of {s:.2e}')
    assert e < tol * s, what


@pytest.mark.parametrize('precision,tol', [('fp32', 2e-5), ('bf16x3', 1e-3)])
def test_modulated_conv_matches_the_reference_golden(precision, tol):
    g = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'stylegan_modconv.npz'))
    cases = {'styled': StyledConvSpec('conv1', 32, 64, 3, 64, 8, True, False),      # bare conv: no tail activation
             'torgb': StyledConvSpec('to_rgb1', 64, 3, 1, 64, 8, False, False)}
    for name, sp in cases.items():
        sd = init_styled_conv_state_dict(sp, int(g['seed']))           # same draws for conv / modulation as the golden script
        sd[f'{sp.prefix}.bias'] = torch.zeros(1, sp.cout, 1, 1)                    # the golden is the conv alone
        x, w = torch.from_numpy(g[f'{name}.x']), torch.from_numpy(g[f'{name}.w'])
        rows = x.shape[0]
        eng = Engine.bare(rows, device=DEV, precision=precision)
        ax, aw = Act(eng, rows, sp.res, sp.res, sp.cin, 'x'), Act(eng, rows, 1, 1, sp.style_dim, 'w')
        out = eng.styled_conv(sd, sp, ax, aw)
        eng.finish()
        ax.t.copy_(nhwc(x))
        aw.t.view(rows, -1).copy_(w.to(DEV))
        eng.forward()
        print(f'{name} [{precision}]')
        close(nchw(out.t, sp.cout), torch.from_numpy(g[f'{name}.y']), tol, 'y')
        out.g.zero_()
        out.g[..., :sp.cout].copy_(nhwc(torch.from_numpy(g[f'{name}.cot'])))
        eng.bwd.run(eng.stream())
        torch.cuda.synchronize()
        close(nchw(ax.g), torch.from_numpy(g[f'{name}.gx']), tol, 'd/dx')
        close(aw.g.view(rows, -1).cpu(), torch.from_numpy(g[f'{name}.gw']), tol, 'd/dw_latent')


@pytest.mark.parametrize('precision,tol', [('fp32', 5e-5), ('bf16x3', 1e-3)])
def test_styled_conv_chain_matches_oracle(precision, tol):
    from oracle import stylegan_oracle as S
    rows, res, D, C = 4, 16, 512, 512
    s1 = StyledConvSpec('convs.1', C, C, 3, D, res, True, True)
    s2 = StyledConvSpec('to_rgbs.0', C, 3, 1, D, res, False, False)
    sd = {**init_styled_conv_state_dict(s1, 3), **init_styled_conv_state_dict(s2, 4)}
    gen = torch.Generator().manual_seed(8)
    x = torch.randn(rows, C, res, res, generator=gen).requires_grad_(True)
    w = torch.randn(rows, D, generator=gen).requires_grad_(True)
    noise = torch.randn(res, res, generator=gen)
    h = S.styled_conv(sd, s1.prefix, x, w, noise)
    img = S.to_rgb_conv(sd, s2.prefix, h, w)
    cot = torch.randn(img.shape, generator=gen)
    gx, gw = torch.autograd.grad((img * cot).sum(), [x, w])

    eng = Engine.bare(rows, device=DEV, precision=precision)
    ax, aw = Act(eng, rows, res, res, C, 'x'), Act(eng, rows, 1, 1, D, 'w')
    ah = eng.styled_conv(sd, s1, ax, aw, noise=noise)
    aimg = eng.styled_conv(sd, s2, ah, aw)
    eng.finish()
    ax.t.copy_(nhwc(x.detach()))
    aw.t.view(rows, -1).copy_(w.detach().to(DEV))
    eng.forward()
    print(f'StyledConv -> ToRGB [{precision}]')
    close(nchw(ah.t), h.detach(), tol, 'hidden')
    close(nchw(aimg.t, 3), img.detach(), tol, 'rgb')
    aimg.g.zero_()
    aimg.g[..., :3].copy_(nhwc(cot))
    eng.bwd.run(eng.stream())
    torch.cuda.synchronize()
    # gradients cross the leaky-ReLU kink of the hidden layer: a pre-activation whose sign differs between the CPU and the
    # GPU summation order (|u| below the ~1e-5 forward difference; ~2 of 524288 here) flips one slope between 1 and 0.2 and
    # moves the 9 x 512 input gradients under it by ~1e-3.  Hence relative L2 against the plain oracle, and the strict bound
    # against the oracle evaluated with the ENGINE's slope mask (same arithmetic, no sign can differ).
    slope = torch.where(nchw(ah.t) > 0, 1.0, 0.2) * 2 ** 0.5
    u = S.modulated_conv(x, w, sd['convs.1.conv.weight'], sd['convs.1.conv.modulation.weight'], sd['convs.1.conv.modulation.bias'])
    u = u + sd['convs.1.noise.weight'] * noise.view(1, 1, res, res) + sd['convs.1.activate.bias'].view(1, -1, 1, 1)
    sgx, sgw = torch.autograd.grad((S.to_rgb_conv(sd, s2.prefix, u * slope, w) * cot).sum(), [x, w])
    for got, ref, strict, what in ((nchw(ax.g), gx, sgx, 'd/dx'), (aw.g.view(rows, -1).cpu(), gw, sgw, 'd/dw_latent')):
        rel = ((got - ref).double().norm() / ref.double().norm()).item()
        print(f'   {what}: relL2 vs oracle {rel:.2e}')
        assert rel < 2e-3, what
        close(got, strict, tol, what + ' (engine slope mask)')


def test_up2_blur_matches_upfirdn2d():
    """ga_up2_blur (ToRGB's skip path) against the oracle's upfirdn2d(up=2, pad=(2,1)) and its autograd adjoint"""
    from gen_adversarial_amd import _lib as L
    from gen_adversarial_amd.engine_core import _ptr
    from oracle import stylegan_oracle as S
    gen = torch.Generator().manual_seed(2)
    lo = torch.randn(3, 4, 6, 10, generator=gen).requires_grad_(True)
    base = torch.randn(3, 4, 12, 20, generator=gen)
    ref = base + S.upfirdn2d(lo, S.make_kernel() * 4, up=2, pad=(2, 1))
    cot = torch.randn(ref.shape, generator=gen)
    (glo,) = torch.autograd.grad((ref * cot).sum(), [lo])
    lo_d, hi_d, cot_d = nhwc(lo.detach()), nhwc(base), nhwc(cot)
    d = L.Up2BlurDesc()
    d.lo_in, d.hi, d.N, d.H, d.W, d.C, d.backward = _ptr(lo_d), _ptr(hi_d), 3, 6, 10, 4, 0
    L.run(d, torch.cuda.current_stream().cuda_stream)
    close(nchw(hi_d), ref.detach(), 1e-6, 'up2 forward')
    out = torch.zeros_like(lo_d)
    b = L.Up2BlurDesc()
    b.hi_in, b.lo, b.N, b.H, b.W, b.C, b.backward = _ptr(cot_d), _ptr(out), 3, 6, 10, 4, 1
    L.run(b, torch.cuda.current_stream().cuda_stream)
    close(nchw(out), glo, 1e-6, 'up2 backward')


@pytest.mark.parametrize('precision,tol', [('fp32', 1e-4), ('bf16x3', 1e-3)])
def test_upsampling_styled_conv_matches_oracle(precision, tol):
    """StyledConv(upsample=True) alone (transposed conv + blur as four parity convs / one 6x6 stride-2 conv); a cotangent on
    the pre-activation side is not available, so the kink is handled as in the chain test (engine slope mask)"""
    from oracle import stylegan_oracle as S
    rows, D = 3, 64
    sp = StyledConvSpec('convs.0', 32, 64, 3, D, 16, True, True, True)
    sd = init_styled_conv_state_dict(sp, 5)
    gen = torch.Generator().manual_seed(9)
    x = torch.randn(rows, 32, 8, 8, generator=gen).requires_grad_(True)
    w = torch.randn(rows, D, generator=gen).requires_grad_(True)
    noise = torch.randn(16, 16, generator=gen)
    eng = Engine.bare(rows, device=DEV, precision=precision)
    ax, aw = Act(eng, rows, 8, 8, 32, 'x'), Act(eng, rows, 1, 1, D, 'w')
    out = eng.styled_conv(sd, sp, ax, aw, noise=noise)
    eng.finish()
    ax.t.copy_(nhwc(x.detach()))
    aw.t.view(rows, -1).copy_(w.detach().to(DEV))
    eng.forward()
    ref = S.styled_conv(sd, sp.prefix, x, w, noise, upsample=True)
    print(f'up-sampling StyledConv [{precision}]')
    close(nchw(out.t), ref.detach(), tol, 'y')
    slope = torch.where(nchw(out.t) > 0, 1.0, 0.2) * 2 ** 0.5
    u = S.modulated_conv(x, w, sd['convs.0.conv.weight'], sd['convs.0.conv.modulation.weight'], sd['convs.0.conv.modulation.bias'],
                         True, True)
    u = u + sd['convs.0.noise.weight'] * noise.view(1, 1, 16, 16) + sd['convs.0.activate.bias'].view(1, -1, 1, 1)
    cot = torch.randn(ref.shape, generator=gen)
    gx, gw = torch.autograd.grad((u * slope * cot).sum(), [x, w])
    out.g.copy_(nhwc(cot))
    eng.bwd.run(eng.stream())
    torch.cuda.synchronize()
    close(nchw(ax.g), gx, tol, 'd/dx')
    close(aw.g.view(rows, -1).cpu(), gw, tol, 'd/dw_latent')


@pytest.mark.parametrize('precision,tol', [('fp32', 1e-4), ('bf16x3', 1e-3)])
def test_generator_matches_oracle(precision, tol):
    """the whole synthesis network at 1/8 width, 64x64 output (10 latents, 4 up-sampling stages): image and d/dlatent"""
    from gen_adversarial_amd.stylegan_spec import build_stylegan_spec, init_stylegan_state_dict
    from oracle import stylegan_oracle as S
    spec = build_stylegan_spec(64, width_div=8, style_dim=128)
    sd = init_stylegan_state_dict(spec, 6)
    rows = 3
    gen = torch.Generator().manual_seed(10)
    lat = torch.randn(rows, spec.n_latent, spec.style_dim, generator=gen).requires_grad_(True)
    img = S.generator_forward(sd, spec, lat)
    cot = torch.randn(img.shape, generator=gen)
    (glat,) = torch.autograd.grad((img * cot).sum(), [lat])

    eng = Engine.bare(rows, device=DEV, precision=precision)
    alat = Act(eng, rows, 1, 1, spec.n_latent * spec.style_dim, 'latent')
    aimg = eng.build_stylegan(sd, spec, alat)
    eng.finish()
    alat.t.view(rows, -1).copy_(lat.detach().reshape(rows, -1).to(DEV))
    eng.forward()
    print(f'generator 64x64 [{precision}]: {len(eng.fwd)} forward ops, {len(eng.bwd)} backward ops')
    close(nchw(aimg.t, 3), img.detach(), tol, 'image')
    assert aimg.t[..., 3].abs().max().item() == 0.0
    aimg.g.zero_()
    aimg.g[..., :3].copy_(nhwc(cot))
    eng.bwd.run(eng.stream())
    torch.cuda.synchronize()
    got = alat.g.view(rows, spec.n_latent, spec.style_dim).cpu()
    rel = ((got - glat).double().norm() / glat.double().norm()).item()
    print(f'   d/dlatent relL2 {rel:.2e} max err {(got - glat).abs().max().item():.2e} of {glat.abs().max().item():.2e}')
    from gradcheck import assert_grad_given_engine_decisions
    assert_grad_given_engine_decisions(eng, lambda t: (S.generator_forward(sd, spec, t) * cot).sum(), lat, got, 1e-3,
                                       f'generator d/dlatent [{precision}]', min_matched=9)
    assert rel < 5e-3                    # secondary: leaky-ReLU kinks of 9 hidden layers (see the chain test)


# ------------------------------------------------------------------------------------------------------------
# against goldens produced by the REFERENCE's own StyledConv / ToRGB / Generator (tests/golden/make_stylegan_full_golden.py)
def _golden():
    z = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'stylegan_full.npz'))
    return {k: z[k] for k in z.files}


@pytest.mark.parametrize('precision,tol', [('fp32', 1e-4), ('bf16x3', 1e-3)])
def test_styled_conv_and_torgb_match_the_reference_golden(precision, tol):
    """reference StyledConv (plain and up-sampling: transposed conv + upfirdn2d blur + noise + FusedLeakyReLU) and ToRGB with
    its upfirdn2d-up-sampled skip: output, d/dx, d/dstyle, d/dskip"""
    from oracle import stylegan_oracle as S
    from gradcheck import assert_grad_given_engine_decisions
    g = _golden()
    layers = {'styled': StyledConvSpec('conv1', 32, 64, 3, 64, 8, True, True),
              'styled_up': StyledConvSpec('convs.0', 32, 64, 3, 64, 16, True, True, True),
              'torgb': StyledConvSpec('to_rgbs.0', 64, 3, 1, 64, 16, False, False)}
    for name, sp in layers.items():
        sd = init_styled_conv_state_dict(sp, 41)
        x, w, cot = (torch.from_numpy(g[f'{name}.{k}']) for k in ('x', 'w', 'cot'))
        rows, rin = x.shape[0], x.shape[2]
        eng = Engine.bare(rows, device=DEV, precision=precision)
        ax, aw = Act(eng, rows, rin, rin, sp.cin, 'x'), Act(eng, rows, 1, 1, sp.style_dim, 'w')
        skip = None
        if sp.activate:
            out = eng.styled_conv(sd, sp, ax, aw, noise=torch.from_numpy(g[f'{name}.noise'])[0, 0])
        else:
            skip = Act(eng, rows, sp.res // 2, sp.res // 2, 4, 'skip')
            out = eng.styled_conv(sd, sp, ax, aw, skip=skip)
        eng.finish()
        ax.t.copy_(nhwc(x))
        aw.t.view(rows, -1).copy_(w.to(DEV))
        if skip is not None:
            skip.t.zero_()
            skip.t[..., :3].copy_(nhwc(torch.from_numpy(g[f'{name}.skip'])))
        eng.forward()
        print(f'{name} vs reference golden [{precision}]')
        close(nchw(out.t, sp.cout), torch.from_numpy(g[f'{name}.y']), tol, 'y')
        out.g.zero_()
        out.g[..., :sp.cout].copy_(nhwc(cot))
        eng.bwd.run(eng.stream())
        torch.cuda.synchronize()
        if sp.activate:          # leaky-ReLU kink: tie mask from the oracle (pinned to this golden at 1e-5 on the CPU)
            noise = torch.from_numpy(g[f'{name}.noise'])
            for i, (got, key) in enumerate(((nchw(ax.g), 'gx'), (aw.g.view(rows, -1).cpu(), 'gw'))):
                other = w if i == 0 else x
                fn = (lambda t: (S.styled_conv(sd, sp.prefix, t, other, noise, upsample=sp.upsample) * cot).sum()) if i == 0 else \
                     (lambda t: (S.styled_conv(sd, sp.prefix, other, t, noise, upsample=sp.upsample) * cot).sum())
                arg = x if i == 0 else w
                ar = arg.clone().requires_grad_(True)
                (g0,) = torch.autograd.grad(fn(ar), [ar])
                ref = torch.from_numpy(g[f'{name}.{key}'])
                assert_grad_given_engine_decisions(eng, fn, arg, got, tol, f'{name}.{key}', golden=(ref, g0))
        else:
            close(nchw(ax.g), torch.from_numpy(g[f'{name}.gx']), tol, 'gx')
            close(aw.g.view(rows, -1).cpu(), torch.from_numpy(g[f'{name}.gw']), tol, 'gw')
            close(nchw(skip.g, 3), torch.from_numpy(g[f'{name}.gskip']), tol, 'gskip')


@pytest.mark.parametrize('precision,tol', [('fp32', 1e-4), ('bf16x3', 1e-3)])
def test_generator_and_mapping_match_the_reference_golden(precision, tol):
    """the reference's Generator(size=32, 512, 8) at full width (512 channels, 8 latents, 3 up-sampling stages):
    forward([latent], input_is_latent=True, randomize_noise=False) image, d/dlatent, and the mapping network"""
    from gen_adversarial_amd.stylegan_spec import build_stylegan_spec, init_stylegan_state_dict
    from oracle import stylegan_oracle as S
    from gradcheck import assert_grad_given_engine_decisions
    g = _golden()
    spec = build_stylegan_spec(int(g['gen_size']))
    sd = init_stylegan_state_dict(spec, int(g['gen_seed']))
    lat, cot = torch.from_numpy(g['gen.latent']), torch.from_numpy(g['gen.cot'])
    rows = lat.shape[0]
    eng = Engine.bare(rows, device=DEV, precision=precision)
    alat = Act(eng, rows, 1, 1, spec.n_latent * spec.style_dim, 'latent')
    aimg = eng.build_stylegan(sd, spec, alat)
    eng.finish()
    alat.t.view(rows, -1).copy_(lat.reshape(rows, -1).to(DEV))
    eng.forward()
    print(f'Generator(32) vs reference golden [{precision}]')
    close(nchw(aimg.t, 3), torch.from_numpy(g['gen.image']), tol, 'image')
    aimg.g.zero_()
    aimg.g[..., :3].copy_(nhwc(cot))
    eng.bwd.run(eng.stream())
    torch.cuda.synchronize()
    got = alat.g.view(rows, spec.n_latent, spec.style_dim).cpu()
    ref = torch.from_numpy(g['gen.glatent'])
    lr = lat.clone().requires_grad_(True)
    (g0,) = torch.autograd.grad((S.generator_forward(sd, spec, lr) * cot).sum(), [lr])
    assert_grad_given_engine_decisions(eng, lambda t: (S.generator_forward(sd, spec, t) * cot).sum(), lat, got, 1e-3, 'd/dlatent',
                                       min_matched=7, golden=(ref, g0))
    z = torch.from_numpy(g['map.z'])
    em = Engine.bare(z.shape[0], device=DEV, precision=precision)
    zb = em.alloc(tuple(z.shape))
    styles = em.build_mapping(sd, zb)
    em.finish()
    zb.copy_(z.to(DEV))
    em.forward()
    close(styles.view(z.shape[0], -1).cpu(), torch.from_numpy(g['map.styles']), tol, 'mapping network')
