"""
CPU: the oracle (our restatement) against golden vectors produced by the reference itself
(tests/golden/make_golden.py).  Tolerance 1e-5 as stated in BASELINE.md §3.
"""
import ast
import os

import numpy as np
import pytest
import torch

from conftest import golden_cfg, load_golden
from gen_adversarial_amd.nvae_spec import (DecCellSpec, EncCellSpec, _Rng, _dec_cell, _enc_cell, build_spec,
                                           init_nvae_state_dict)
from gen_adversarial_amd.vgg_spec import build_vgg_spec, init_vgg_state_dict
from oracle import defender_oracle as D
from oracle import nvae_oracle as O

CASES = ['A_cos07', 'A_zero_noise2', 'B_adaptive', 'A_nf2']
TOL = 1e-5


def _t(a):
    return torch.from_numpy(np.asarray(a))


def _setup(g):
    cfg, res = golden_cfg(g)
    spec = build_spec(cfg, res)
    sd = init_nvae_state_dict(cfg, res, int(g['nvae_seed']))
    vspec = build_vgg_spec(int(g['n_classes']), int(g['width_div']))
    vsd = init_vgg_state_dict(int(g['n_classes']), int(g['width_div']), int(g['vgg_seed']))
    alphas = [float(a) * float(g['attenuation']) for a in g['alphas']]
    return spec, sd, vspec, vsd, alphas


@pytest.mark.parametrize('name', CASES)
def test_state_dict_layout_matches_reference(name, golden_cases):
    g = golden_cases[name]
    cfg, res = golden_cfg(g)
    sd = init_nvae_state_dict(cfg, res, 0)
    ref = {str(k): ast.literal_eval(str(s)) for k, s in zip(g['sd_keys'], g['sd_shapes'])}
    assert set(sd.keys()) == set(ref.keys())
    for k, v in sd.items():
        assert tuple(v.shape) == ref[k], k
    vsd = init_vgg_state_dict(int(g['n_classes']), int(g['width_div']), 0)
    vref = {str(k): ast.literal_eval(str(s)) for k, s in zip(g['vgg_keys'], g['vgg_shapes'])}
    assert set(vsd.keys()) == set(vref.keys())
    for k, v in vsd.items():
        assert tuple(v.shape) == vref[k], k


@pytest.mark.parametrize('name', CASES)
def test_purify_matches_reference(name, golden_cases):
    g = golden_cases[name]
    spec, sd, _, _, alphas = _setup(g)
    eps = [_t(g[f'eps_{i}']) for i in range(len(spec.groups))]
    out = O.nvae_purify(sd, spec, _t(g['x']), alphas, eps, 0.6)
    np.testing.assert_allclose(out.numpy(), g['purified_only'], atol=TOL, rtol=0)


@pytest.mark.parametrize('name', CASES)
def test_defender_logits_and_input_grad(name, golden_cases):
    g = golden_cases[name]
    spec, sd, vspec, vsd, alphas = _setup(g)
    eps = [_t(g[f'eps_{i}']) for i in range(len(spec.groups))]
    x = _t(g['x']).clone().requires_grad_(True)
    logits, purified = D.nvae_defender(sd, spec, vsd, vspec, x, alphas, eps, _t(g['input_noise']),
                                       float(g['noise_eps']))
    np.testing.assert_allclose(purified.detach().numpy(), g['purified'], atol=TOL, rtol=0)
    np.testing.assert_allclose(logits.detach().numpy(), g['logits'], atol=TOL, rtol=0)
    (gx,) = torch.autograd.grad((logits * _t(g['cotangent'])).sum(), [x])
    np.testing.assert_allclose(gx.numpy(), g['grad_x'], atol=TOL, rtol=0)


@pytest.mark.parametrize('name', CASES)
def test_eot_wrapper_and_ce_grad(name, golden_cases):
    g = golden_cases[name]
    spec, sd, vspec, vsd, alphas = _setup(g)
    eot = int(g['eot_steps'])
    eps = [_t(g[f'eot_eps_{i}']) for i in range(len(spec.groups))]
    x1 = _t(g['x'][:1]).clone().requires_grad_(True)
    logits, _ = D.eot_defender(sd, spec, vsd, vspec, x1, eot, alphas, eps, _t(g['eot_noise']), float(g['noise_eps']))
    np.testing.assert_allclose(logits.detach().numpy(), g['eot_logits'], atol=TOL, rtol=0)
    loss = torch.nn.functional.cross_entropy(logits, logits.argmax(dim=1))
    (gx,) = torch.autograd.grad(loss, [x1])
    np.testing.assert_allclose(gx.numpy(), g['eot_ce_grad'], atol=TOL, rtol=0)


def test_cells_match_reference_modules():
    g = load_golden('nvae_modules.npz')
    cases = [('enc_same', EncCellSpec('c', 8, 8, False)), ('enc_down', EncCellSpec('c', 8, 16, True)),
             ('dec_same', DecCellSpec('c', 8, 8, False, 6)), ('dec_up', DecCellSpec('c', 8, 4, True, 3))]
    for name, cell in cases:
        sd = {}
        rng = _Rng(int(g[f'{name}_seed']))
        if isinstance(cell, EncCellSpec):
            _enc_cell(sd, rng, cell)
            fn = O.enc_cell
        else:
            _dec_cell(sd, rng, cell)
            fn = O.dec_cell
        x = _t(g[f'{name}_x']).clone().requires_grad_(True)
        y = fn(sd, cell, x)
        np.testing.assert_allclose(y.detach().numpy(), g[f'{name}_y'], atol=TOL, rtol=0, err_msg=name)
        (gx,) = torch.autograd.grad((y * _t(g[f'{name}_cot'])).sum(), [x])
        np.testing.assert_allclose(gx.numpy(), g[f'{name}_gx'], atol=TOL, rtol=0, err_msg=name)


def test_dml_mean_and_normal():
    g = load_golden('nvae_modules.npz')
    lg = _t(g['dml_logits']).clone().requires_grad_(True)
    y = O.disc_mix_logistic_mean(lg, 10)
    np.testing.assert_allclose(y.detach().numpy(), g['dml_mean'], atol=TOL, rtol=0)
    (gl,) = torch.autograd.grad((y * _t(g['dml_cot'])).sum(), [lg])
    np.testing.assert_allclose(gl.numpy(), g['dml_glogits'], atol=TOL, rtol=0)
    mu, sigma = O.normal_mu_sigma(_t(g['normal_mu_in']), _t(g['normal_ls_in']), 0.6)
    np.testing.assert_allclose(mu.numpy(), g['normal_mu'], atol=TOL, rtol=0)
    np.testing.assert_allclose(sigma.numpy(), g['normal_sigma'], atol=1e-5, rtol=1e-6)
    z = mu + _t(g['normal_eps']) * sigma
    np.testing.assert_allclose(z.numpy(), g['normal_z'], atol=1e-5, rtol=1e-6)


def test_e4e_oracle_matches_the_reference_encoder():
    """oracle/e4e_oracle.py against the reference's Encoder4Editing (tests/golden/make_e4e_golden.py): latents and input
    gradient of the full-width IR-SE50 + FPN + 10 style heads on 64x64 inputs"""
    import os
    import numpy as np
    import torch
    from gen_adversarial_amd.e4e_spec import build_e4e_spec, init_e4e_state_dict
    from oracle.e4e_oracle import e4e_encode
    g = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'e4e_ir50_s64.npz'))
    size, seed = int(g['stylegan_size']), int(g['seed'])
    spec = build_e4e_spec(size)
    sd = init_e4e_state_dict(size, 1, seed)
    x = torch.from_numpy(g['x']).requires_grad_(True)
    w = e4e_encode(sd, spec, x)
    (gx,) = torch.autograd.grad((w * torch.from_numpy(g['cot'])).sum(), [x])
    assert (w.detach() - torch.from_numpy(g['w'])).abs().max().item() < 1e-5
    assert (gx - torch.from_numpy(g['gx'])).abs().max().item() < 1e-5 * max(1.0, float(np.abs(g['gx']).max()))


def test_stylegan_modulated_conv_oracle_matches_the_reference():
    """oracle/stylegan_oracle.py against the reference's ModulatedConv2d (tests/golden/make_stylegan_golden.py): output,
    d/dx and d/dstyle of the demodulated 3x3 conv (StyledConv) and of the plain 1x1 conv (ToRGB)"""
    from oracle.stylegan_oracle import modulated_conv
    from gen_adversarial_amd.stylegan_spec import StyledConvSpec, init_styled_conv_state_dict
    g = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'stylegan_modconv.npz'))
    cases = {'styled': StyledConvSpec('conv1', 32, 64, 3, 64, 8, True, True),
             'torgb': StyledConvSpec('to_rgb1', 64, 3, 1, 64, 8, False, False),
             'up': StyledConvSpec('convs.0', 32, 64, 3, 64, 16, True, True, True)}      # transposed conv, blur left out
    for name, sp in cases.items():
        sd = init_styled_conv_state_dict(sp, int(g['seed']))
        x = torch.from_numpy(g[f'{name}.x']).requires_grad_(True)
        w = torch.from_numpy(g[f'{name}.w']).requires_grad_(True)
        y = modulated_conv(x, w, sd[f'{sp.prefix}.conv.weight'], sd[f'{sp.prefix}.conv.modulation.weight'],
                           sd[f'{sp.prefix}.conv.modulation.bias'], sp.demodulate, sp.upsample, blur=False)
        gx, gw = torch.autograd.grad((y * torch.from_numpy(g[f'{name}.cot'])).sum(), [x, w])
        for got, key in ((y, 'y'), (gx, 'gx'), (gw, 'gw')):
            ref = torch.from_numpy(g[f'{name}.{key}'])
            assert (got.detach() - ref).abs().max().item() < 1e-5 * max(1.0, ref.abs().max().item()), (name, key)
