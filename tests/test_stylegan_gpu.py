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
