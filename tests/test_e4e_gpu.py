"""
GPU parity of the e4e encoder path (Encoder4Editing, encoding/encoder.py:57-140) — latents and input
gradient — against (a) the golden produced by the reference's own module (full-width IR-SE50, 64x64 input, 10 style heads)
and (b) the CPU oracle on reduced configurations.  Tolerance 1e-3 absolute on the latents (|w| ~ 3).

Input gradients: PReLU (encoder body) and LeakyReLU (style heads) have a kink at 0; a pre-activation whose sign differs
between the CPU and the GPU summation order (|value| below the ~1e-5 forward difference) flips one derivative between
1 and its slope, which moves the gradient by up to ~1 % (the heads end on 1x1 maps with 512 values per row).  Exactness of
the backward arithmetic is therefore shown two ways: against the oracle in relative L2 (kinks included), and — strictly —
against autograd evaluated on the engine's OWN forward activations, head by head (no sign can differ there).
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

if not torch.cuda.is_available():
    pytest.skip('needs a GPU', allow_module_level=True)

from gen_adversarial_amd.e4e_spec import build_e4e_spec, init_e4e_state_dict   # noqa: E402
from gen_adversarial_amd.engine import Engine   # noqa: E402

DEV = 'cuda:0'


eng_last = [None]          # the engine of the last _run (its stored activations feed the decision replay)


def _run(sd, spec, x, cot, precision):
    eng = eng_last[0] = Engine(None, None, (3, x.shape[2], x.shape[3]), sd, spec, rows=x.shape[0], rep=1, alphas=[], device=DEV, precision=precision)
    eng.x_in.copy_(x.to(DEV))
    eng.forward()
    w = eng.logits.view(x.shape[0], spec.style_count, spec.style_dim).cpu()
    eng.dlogits.view(x.shape[0], spec.style_count, spec.style_dim).copy_(cot.to(DEV))
    eng.backward()
    return w, eng.dx.cpu()


@pytest.mark.parametrize('precision,tol', [('fp32', 2e-4), ('bf16x3', 1e-3)])
def test_e4e_matches_the_reference_golden(precision, tol):
    g = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'e4e_ir50_s64.npz'))
    size, seed = int(g['stylegan_size']), int(g['seed'])
    spec, sd = build_e4e_spec(size), init_e4e_state_dict(size, 1, seed)
    w, gx = _run(sd, spec, torch.from_numpy(g['x']), torch.from_numpy(g['cot']), precision)
    e_w = (w - torch.from_numpy(g['w'])).abs().max().item()
    ref_g = torch.from_numpy(g['gx'])
    e_g = (gx - ref_g).abs().max().item()
    print(f'e4e golden [{precision}]: latents err {e_w:.2e} (max {np.abs(g["w"]).max():.2f}) input-grad err {e_g:.2e} (max {ref_g.abs().max().item():.2f})')
    rel = ((gx - ref_g).double().norm() / ref_g.double().norm()).item()
    print(f'   input-grad relL2 {rel:.2e}')
    assert e_w < tol
    # the golden gradient is the reference's own; the tie mask comes from the oracle (pinned to the same golden at 1e-5)
    from gradcheck import assert_grad_given_engine_decisions
    from oracle.e4e_oracle import e4e_encode
    xg, cg = torch.from_numpy(g['x']), torch.from_numpy(g['cot'])
    xr = xg.clone().requires_grad_(True)
    (g0,) = torch.autograd.grad((e4e_encode(sd, spec, xr) * cg).sum(), [xr])
    assert (g0 - ref_g).abs().max().item() < 1e-5 * ref_g.abs().max().item()
    assert_grad_given_engine_decisions(eng_last[0], lambda t: (e4e_encode(sd, spec, t) * cg).sum(), xg, gx, 1e-3,
                                       f'e4e input gradient vs the reference golden [{precision}]', min_matched=20, golden=(ref_g, g0))
    assert rel < 3e-2                        # secondary (PReLU / LeakyReLU near-ties included)


@pytest.mark.parametrize('units,res,size', [((1, 2, 2, 1), 64, 64), ((2, 1, 1, 2), 128, 256)])
def test_reduced_e4e_matches_oracle(units, res, size):
    """quarter-width networks: every unit kind (sub-sampled / conv / identity shortcut), both FPN levels, heads that end on
    1x1 maps early (64-px input) and heads that run their full depth (128-px input)"""
    from oracle.e4e_oracle import e4e_encode
    spec, sd = build_e4e_spec(size, 4, units), init_e4e_state_dict(size, 4, 9, units)
    gen = torch.Generator().manual_seed(1)
    x = torch.rand(2, 3, res, res, generator=gen)
    xr = x.clone().requires_grad_(True)
    ref = e4e_encode(sd, spec, xr)
    cot = torch.randn(ref.shape, generator=gen)
    (gx,) = torch.autograd.grad((ref * cot).sum(), [xr])
    w, g = _run(sd, spec, x, cot, 'bf16x3')
    e_w, e_g = (w - ref.detach()).abs().max().item(), (g - gx).abs().max().item()
    print(f'e4e reduced {units} {res}px: latents err {e_w:.2e} (max {ref.abs().max().item():.2f}) grad err {e_g:.2e} (max {gx.abs().max().item():.2f})')
    rel = ((g - gx).double().norm() / gx.double().norm()).item()
    from gradcheck import assert_grad_given_engine_decisions
    assert_grad_given_engine_decisions(eng_last[0], lambda t: (e4e_encode(sd, spec, t) * cot).sum(), x, g, 1e-3,
                                       f'reduced e4e {units} input gradient', min_matched=8)
    assert e_w < 1e-3 and rel < 3e-2


def test_style_head_backward_is_exact_on_the_engines_activations():
    """every stage gradient of two style heads against autograd run on the engine's own FPN output (full width)"""
    import torch.nn.functional as F
    size, units = 64, (1, 1, 1, 1)
    spec, sd = build_e4e_spec(size, 1, units), init_e4e_state_dict(size, 1, 9, units)
    gen = torch.Generator().manual_seed(1)
    x = torch.rand(2, 3, 64, 64, generator=gen)
    eng = Engine(None, None, (3, 64, 64), sd, spec, rows=2, rep=1, alphas=[], device=DEV, precision='fp32')
    eng.x_in.copy_(x.to(DEV))
    eng.forward()
    full = torch.randn(2, spec.style_count, spec.style_dim, generator=gen)
    for j, src in ((8, 'e4e.latlayer2.sum'), (4, 'e4e.latlayer1.sum')):
        cot = torch.zeros_like(full)
        cot[:, j] = full[:, j]
        eng.dlogits.view_as(cot).copy_(cot.to(DEV))
        eng.backward()
        feat = eng.acts[src].t.permute(0, 3, 1, 2).cpu().clone().requires_grad_(True)
        hs, h = [], feat
        for k in range(spec.style_pools[j]):
            h = F.conv2d(h if k == 0 else F.leaky_relu(h), sd[f'styles.{j}.convs.{2 * k}.weight'], sd[f'styles.{j}.convs.{2 * k}.bias'],
                         stride=2, padding=1)
            h.retain_grad()
            hs.append(h)
        wl = sd[f'styles.{j}.linear.weight']
        out = F.linear(F.leaky_relu(h).view(-1, spec.style_dim), wl / wl.shape[1] ** 0.5, sd[f'styles.{j}.linear.bias'])
        (out * full[:, j]).sum().backward()
        for k, h in enumerate(hs):
            a = eng.acts[f'e4e.styles.{j}.h{k}']
            assert (a.g.permute(0, 3, 1, 2).cpu() - h.grad).abs().max().item() < 1e-5 * max(1.0, h.grad.abs().max().item())
        fg = eng.acts[src].g.permute(0, 3, 1, 2).cpu()
        assert (fg - feat.grad).abs().max().item() < 1e-5 * max(1.0, feat.grad.abs().max().item())
