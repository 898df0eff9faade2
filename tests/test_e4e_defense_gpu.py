"""
GPU parity of the composed e4e + StyleGAN2 defender (E4EStyleGanDefenseModel, src/defenses/ours/
models.py:80-132, behind MLVGMDefenseModel.__call__, abstract_models.py:161-193) against the CPU oracle
(oracle/defender_oracle.e4e_defender_call) on a reduced configuration: quarter-width IR-SE encoder, 1/8-width 64-px generator,
face_pool to 32 px, 1/8-width ResNet.  Also the small ops of the path (PixelNorm + mapping MLP, latent mixing, face_pool +
de-normalisation into the classifier's space-to-depth layout).
Tolerance 1e-3 absolute on logits and purified image (BASELINE.json north_star); input gradients in relative L2 (PReLU /
leaky-ReLU / ReLU / max-pool kinks along ~150 layers: see test_e4e_gpu.py and test_stylegan_gpu.py).
"""
import pytest
import torch

pytestmark = pytest.mark.gpu

if not torch.cuda.is_available():
    pytest.skip('needs a GPU', allow_module_level=True)

from test_host_cpu import _small_e4e_defense   # noqa: E402

DEV = 'cuda:0'


def test_mapping_network_matches_oracle():
    from gen_adversarial_amd.engine import Engine
    from gen_adversarial_amd.stylegan_spec import build_stylegan_spec, init_stylegan_state_dict
    from oracle import stylegan_oracle as S
    spec = build_stylegan_spec(16, width_div=8, style_dim=512)
    sd = init_stylegan_state_dict(spec, 2)
    z = torch.randn(36, 512, generator=torch.Generator().manual_seed(1))
    ref = S.mapping_network(sd, z)
    for precision, tol in (('fp32', 2e-5), ('bf16x3', 1e-3)):
        eng = Engine.bare(36, device=DEV, precision=precision)
        zb = eng.alloc((36, 512))
        out = eng.build_mapping(sd, zb)
        eng.finish()
        zb.copy_(z.to(DEV))
        eng.forward()
        e = (out.view(36, 512).cpu() - ref).abs().max().item()
        print(f'mapping network [{precision}]: err {e:.2e} of {ref.abs().max().item():.2e}')
        assert e < tol * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize('precision,tol,share', [('fp32', 2e-4, False), ('bf16x3', 1e-3, False), ('fp32', 2e-4, True), ('bf16x3', 1e-3, True)])
def test_e4e_defender_matches_oracle(precision, tol, share):
    """share: the encoder runs once per image and its EoT replicas read the same codes — compared with the oracle's literal
    x.repeat(eot) path all the same"""
    from oracle import defender_oracle as D
    rows, rep = 4, 2
    eng, (esd, espec, gsd, gspec, avg, csd, cspec, alphas) = _small_e4e_defense(rows, rep, DEV, False, precision, share)
    assert eng.enc_rows == (rows // rep if share else rows)
    gen = torch.Generator().manual_seed(3)
    x = torch.rand(rows // rep, 3, 64, 64, generator=gen)
    z = torch.randn(rows, gspec.n_latent, gspec.style_dim, generator=gen)
    xr = x.clone().requires_grad_(True)
    logits, purified = D.e4e_defender_call(esd, espec, gsd, gspec, avg, csd, cspec, xr.repeat_interleave(rep, dim=0), alphas, z, 32)
    cot = torch.randn(logits.shape, generator=gen)
    (gx,) = torch.autograd.grad((logits * cot).sum(), [xr])

    eng.x_in.copy_(x.to(DEV))
    eng.eps[0].copy_(z.to(DEV))
    eng.forward()
    e_p = (eng.purified_nchw().cpu() - purified.detach()).abs().max().item()
    e_l = (eng.logits.view(rows, -1).cpu() - logits.detach()).abs().max().item()
    eng.dlogits.view(rows, -1).copy_(cot.to(DEV))
    eng.backward()
    rel = ((eng.dx.cpu() - gx).double().norm() / gx.double().norm()).item()
    print(f'e4e defender [{precision}{", shared encoder" if share else ""}]: {len(eng.fwd)} + {len(eng.bwd)} ops; purified err {e_p:.2e} logits err {e_l:.2e} '
          f'(|logits| {logits.abs().max().item():.2f}); input-grad relL2 {rel:.2e} (|g| {gx.abs().max().item():.2e})')
    assert e_p < tol and e_l < tol * max(1.0, logits.abs().max().item())
    from gradcheck import assert_grad_given_engine_decisions
    assert_grad_given_engine_decisions(
        eng, lambda t: (D.e4e_defender_call(esd, espec, gsd, gspec, avg, csd, cspec, t.repeat_interleave(rep, dim=0), alphas, z, 32)[0] * cot).sum(),
        x, eng.dx, 1e-3, f'e4e defender input gradient [{precision}]', min_matched=20)
    assert rel < 3e-2                        # secondary

    # the mixing alphas are device data: changing them needs no re-build, and alpha = 1 everywhere cuts the encoder off
    eng.set_alphas([1.0] * gspec.n_latent)
    eng.forward()
    eng.backward()
    assert eng.dx.abs().max().item() == 0.0
    l1, _ = D.e4e_defender_call(esd, espec, gsd, gspec, avg, csd, cspec, x.repeat_interleave(rep, dim=0), [1.0] * gspec.n_latent, z, 32)
    assert (eng.logits.view(rows, -1).cpu() - l1).abs().max().item() < tol * max(1.0, l1.abs().max().item())


def test_e4e_defender_through_the_reference_api(tmp_path):
    """experiment 'gender', defense_type 'ours' through load(args) (src/experiments/load_defense.py:27-41,124-133): checkpoint
    layouts of loading_utils.py:10-16,37-48 (pSp: 'state_dict' with encoder./decoder. prefixes, 'latent_avg', 'opts'), EoT
    wrapper, autograd to the input, get_purified, purify in the normalised domain, mutable interpolation_alphas"""
    from argparse import Namespace
    import yaml
    from gen_adversarial_amd.experiments.load_defense import load
    from oracle import defender_oracle as D
    _, (esd, espec, gsd, gspec, avg, csd, cspec, alphas) = _small_e4e_defense(dry_run=True, device='cpu')
    ck = {'state_dict': {**{'encoder.' + k: v for k, v in esd.items()}, **{'decoder.' + k: v for k, v in gsd.items()}},
          'latent_avg': avg, 'opts': {'stylegan_size': gspec.size, 'start_from_latent_avg': True, 'encoder_type': 'Encoder4Editing'}}
    torch.save(ck, tmp_path / 'e4e.pt')
    torch.save({'state_dict': csd}, tmp_path / 'resnet.pt')
    with open(tmp_path / 'cfg.yaml', 'w') as f:
        yaml.safe_dump({'classifier_path': str(tmp_path / 'resnet.pt'), 'autoencoder_path': str(tmp_path / 'e4e.pt'),
                        'interpolation_alphas': [a / 0.5 for a in alphas], 'alpha_attenuation': 0.5, 'initial_noise_eps': 0.0,
                        'gaussian_blur_input': False}, f)
    eot = 3
    args, model = load(Namespace(config=str(tmp_path / 'cfg.yaml'), experiment='gender', defense_type='ours', eot_steps=eot, device=DEV))
    gen = torch.Generator().manual_seed(4)
    x = torch.rand(1, 3, 64, 64, generator=gen)
    z = torch.randn(eot, gspec.n_latent, gspec.style_dim, generator=gen)
    xr = x.clone().requires_grad_(True)
    logits, purified = D.e4e_defender_call(esd, espec, gsd, gspec, avg, csd, cspec, xr.repeat(eot, 1, 1, 1), alphas, z, 64)
    mean = logits.mean(dim=0, keepdim=True)
    (gx,) = torch.autograd.grad(mean[0, 1], [xr])

    model.model.fixed_noise([z.to(DEV)], None)
    xd = x.to(DEV).requires_grad_(True)
    out = model(xd)
    assert out.shape == (1, 2)
    assert (out.detach().cpu() - mean.detach()).abs().max().item() < 1e-3
    (g,) = torch.autograd.grad(out[0, 1], [xd])
    rel = ((g.cpu() - gx).double().norm() / gx.double().norm()).item()
    print(f'e4e defender API: EoT-{eot} logits err {(out.detach().cpu() - mean.detach()).abs().max().item():.2e}, input-grad relL2 {rel:.2e}')
    from gradcheck import assert_grad_given_engine_decisions
    assert_grad_given_engine_decisions(
        model.model._engine(eot, eot), lambda t: D.e4e_defender_call(esd, espec, gsd, gspec, avg, csd, cspec, t.repeat(eot, 1, 1, 1),
                                                                    alphas, z, 64)[0].mean(dim=0)[1],
        x, g, 1e-3, 'e4e defender API input gradient', min_matched=20)
    assert rel < 3e-2                        # secondary

    # get_purified: the de-normalised reconstruction of a single draw; purify: the same in the normalised domain
    model.model.fixed_noise([z[:1].to(DEV)], None)
    p = model.get_purified(x.to(DEV))
    assert p.shape == (1, 3, 64, 64) and (p.cpu() - purified[:1].detach()).abs().max().item() < 1e-3
    pn = model.model.purify(x.to(DEV) * 2 - 1)
    assert (pn.cpu() - (purified[:1].detach() * 2 - 1)).abs().max().item() < 2e-3

    # alpha learning overwrites the list in place (src/experiments/alpha_learning/common_utils.py:88)
    model.model.interpolation_alphas[:] = [0.0] * gspec.n_latent
    l0, _ = D.e4e_defender_call(esd, espec, gsd, gspec, avg, csd, cspec, x, [0.0] * gspec.n_latent, z[:1], 64)
    assert (model.model(x.to(DEV)).cpu() - l0).abs().max().item() < 1e-3
    model.model.fixed_noise(None, None)
    a, b = model(x.to(DEV)), model(x.to(DEV))            # alpha = 0: no randomness left
    assert torch.equal(a, b)
    with pytest.raises(ValueError):
        from gen_adversarial_amd.defenses.ours.models import E4EStyleGanDefenseModel
        E4EStyleGanDefenseModel(model.model.classifier, str(tmp_path / 'e4e.pt'), [0.1] * 3, device=DEV)


def test_fullsize_e4e_defender_properties():
    """The reference's sizes — IR-SE50 on 256x256, 18 x 512 latents, StyleGAN2 at 1024x1024, face_pool to 256, ResNet-50 — with
    random weights: too large for the CPU oracle, so size-independent properties: finite outputs in range, EoT replicas with
    equal noise give bitwise equal rows, the shared-encoder plan equals the literal-repeat plan, the backward pass is linear
    in its cotangent."""
    from gen_adversarial_amd.engine import Engine, WeightStore
    from gen_adversarial_amd.e4e_spec import build_e4e_spec, init_e4e_state_dict
    from gen_adversarial_amd.resnet_spec import build_resnet_spec, init_resnet_state_dict
    from gen_adversarial_amd.stylegan_spec import build_stylegan_spec, init_stylegan_state_dict
    espec, esd = build_e4e_spec(1024), init_e4e_state_dict(1024, 1, 0)
    gspec = build_stylegan_spec(1024)
    gsd = init_stylegan_state_dict(gspec, 1)
    cspec, csd = build_resnet_spec(2), init_resnet_state_dict(2, 1, 2)
    assert (gspec.n_latent, gspec.style_dim, gspec.convs[-1].cout) == (18, 512, 32)
    gen = torch.Generator().manual_seed(5)
    avg = 0.1 * torch.randn(18, 512, generator=gen)
    alphas = [0.05 * (j % 5) for j in range(18)]
    store = WeightStore(DEV)
    rows, rep = 2, 2
    x = torch.rand(1, 3, 256, 256, generator=gen)
    z = torch.randn(1, 18, 512, generator=gen).repeat(rows, 1, 1)          # equal noise for both replicas
    out = {}
    for share in (False, True):
        eng = Engine.bare(rows, device=DEV, store=store, rep=rep, resolution=(3, 256, 256), alphas=alphas, share_encoder=share)
        eng.build_e4e_defense(esd, espec, gsd, gspec, avg, csd, cspec, pool_to=256)
        eng.x_in.copy_(x.to(DEV))
        eng.eps[0].copy_(z.to(DEV))
        eng.forward()
        logits, purified = eng.logits.view(rows, -1).clone(), eng.purified_nchw()
        assert torch.isfinite(logits).all() and torch.isfinite(purified).all() and purified.shape == (rows, 3, 256, 256)
        assert torch.equal(logits[0], logits[1]) and torch.equal(purified[0], purified[1])
        grads = []
        a = torch.randn(rows, 2, generator=gen).to(DEV)
        b = torch.randn(rows, 2, generator=gen).to(DEV)
        for c in (a, b, a + b):
            eng.dlogits.view(rows, -1).copy_(c)
            eng.backward()
            grads.append(eng.dx.clone())
        scale = grads[2].abs().max().item()
        lin = (grads[2] - grads[0] - grads[1]).abs().max().item()
        print(f'full-size e4e defender (share={share}): logits {logits[0].tolist()}, |dx| {scale:.2e}, backward linearity {lin:.2e}')
        assert torch.isfinite(grads[2]).all() and scale > 0 and lin < 1e-4 * scale
        out[share] = (logits, purified, grads[0], a)
        del eng
        torch.cuda.empty_cache()
    l0, p0, _, _ = out[False]
    l1, p1, _, _ = out[True]
    # the two plans run the encoder at different row counts (other tiles / split-K orders): rounding-level differences in the
    # latents pass through 150 random-weight layers, hence the image tolerance of the path (1e-3), not bitwise equality
    assert (l0 - l1).abs().max().item() < 1e-4 * max(1.0, l0.abs().max().item()) and (p0 - p1).abs().max().item() < 1e-3


@pytest.mark.parametrize('noise_eps,blur', [(4.0, False), (0.0, True)])
def test_e4e_defender_preprocessing_configs(tmp_path, noise_eps, blur):
    """configs/ours_*_noise_gender.yaml (initial_noise_eps 4.0: the encoder can no longer be shared by the EoT replicas) and
    ours_*_blur_gender.yaml (gaussian_blur_input) through load(args), against the oracle's pre-processing helpers
    (abstract_models.py:129-159) in front of its e4e defender"""
    from argparse import Namespace
    import yaml
    from gen_adversarial_amd.experiments.load_defense import load
    from oracle import defender_oracle as D
    _, (esd, espec, gsd, gspec, avg, csd, cspec, alphas) = _small_e4e_defense(dry_run=True, device='cpu')
    ck = {'state_dict': {**{'encoder.' + k: v for k, v in esd.items()}, **{'decoder.' + k: v for k, v in gsd.items()}},
          'latent_avg': avg, 'opts': {'stylegan_size': gspec.size, 'start_from_latent_avg': True, 'encoder_type': 'Encoder4Editing'}}
    torch.save(ck, tmp_path / 'e4e.pt')
    torch.save({'state_dict': csd}, tmp_path / 'resnet.pt')
    with open(tmp_path / 'cfg.yaml', 'w') as f:
        yaml.safe_dump({'classifier_path': str(tmp_path / 'resnet.pt'), 'autoencoder_path': str(tmp_path / 'e4e.pt'),
                        'interpolation_alphas': list(alphas), 'alpha_attenuation': 1.0, 'initial_noise_eps': noise_eps,
                        'gaussian_blur_input': blur}, f)
    eot = 2
    args, model = load(Namespace(config=str(tmp_path / 'cfg.yaml'), experiment='gender', defense_type='ours', eot_steps=eot, device=DEV))
    gen = torch.Generator().manual_seed(7)
    x = torch.rand(1, 3, 64, 64, generator=gen)
    z = torch.randn(eot, gspec.n_latent, gspec.style_dim, generator=gen)
    noise = torch.randn(eot, 3, 64, 64, generator=gen)
    xr = x.clone().requires_grad_(True)
    pre = D.apply_gaussian_blur(xr) if blur else xr
    pre = D.add_gaussian_noise(pre.repeat(eot, 1, 1, 1), noise, noise_eps)
    logits, _ = D.e4e_defender_call(esd, espec, gsd, gspec, avg, csd, cspec, pre, alphas, z, 64)
    mean = logits.mean(dim=0, keepdim=True)
    (gx,) = torch.autograd.grad(mean[0, 0], [xr])
    model.model.fixed_noise([z.to(DEV)], noise.to(DEV) if noise_eps else None)
    xd = x.to(DEV).requires_grad_(True)
    out = model(xd)
    (g,) = torch.autograd.grad(out[0, 0], [xd])
    eng = model.model._engine(eot, eot)
    assert eng.share_encoder == (noise_eps == 0.0)
    e = (out.detach().cpu() - mean.detach()).abs().max().item()
    rel = ((g.cpu() - gx).double().norm() / gx.double().norm()).item()
    print(f'e4e defender, noise_eps {noise_eps}, blur {blur}: logits err {e:.2e}, input-grad relL2 {rel:.2e}')
    from gradcheck import assert_grad_given_engine_decisions

    def loss(t):
        p_ = D.apply_gaussian_blur(t) if blur else t
        p_ = D.add_gaussian_noise(p_.repeat(eot, 1, 1, 1), noise, noise_eps)
        return D.e4e_defender_call(esd, espec, gsd, gspec, avg, csd, cspec, p_, alphas, z, 64)[0].mean(dim=0)[0]
    assert_grad_given_engine_decisions(eng, loss, x, g, 1e-3, f'e4e defender input gradient (noise_eps {noise_eps}, blur {blur})',
                                       min_matched=20)
    assert e < 1e-3 and rel < 3e-2


@pytest.mark.parametrize('precision,tol', [('fp32', 2e-4), ('bf16x3', 1e-3)])
def test_e4e_defender_matches_the_reference_purify_golden(precision, tol):
    """tests/golden/e4e_purify.npz: the REFERENCE's E4EStyleGanDefenseModel.__call__(x, preds_only=False) (full-width IR-SE50 +
    Generator(32), latent_avg, alphas x attenuation, recorded torch.normal draw): codes, purified image, and the input gradient
    through the returned purified image (src/defenses/ours/models.py:105-132, abstract_models.py:161-193)"""
    import numpy as np
    import os
    from gen_adversarial_amd.engine import Engine
    from gen_adversarial_amd.e4e_spec import build_e4e_spec, init_e4e_state_dict
    from gen_adversarial_amd.resnet_spec import build_resnet_spec, init_resnet_state_dict
    from gen_adversarial_amd.stylegan_spec import build_stylegan_spec, init_stylegan_state_dict
    from gradcheck import assert_grad_given_engine_decisions
    from oracle import defender_oracle as D
    z_ = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'e4e_purify.npz'))
    g = {k: z_[k] for k in z_.files}
    size = int(g['size'])
    espec, esd = build_e4e_spec(size), init_e4e_state_dict(size, 1, int(g['enc_seed']))
    gspec = build_stylegan_spec(size)
    gsd = init_stylegan_state_dict(gspec, int(g['gen_seed']))
    cspec, csd = build_resnet_spec(2, 8, (1, 1, 1, 1)), init_resnet_state_dict(2, 8, 5, (1, 1, 1, 1))   # any classifier: not compared
    x, z, avg = (torch.from_numpy(g[f'purify.{k}']) for k in ('x', 'z', 'latent_avg'))
    alphas = [float(a) for a in g['purify.alphas']]
    rows = x.shape[0]
    eng = Engine.bare(rows, device=DEV, precision=precision, rep=1, resolution=(3, x.shape[2], x.shape[3]), alphas=alphas)
    eng.build_e4e_defense(esd, espec, gsd, gspec, avg, csd, cspec, pool_to=size)
    eng.x_in.copy_(x.to(DEV))
    eng.eps[0].copy_(z.to(DEV))
    eng.forward()
    ref = torch.from_numpy(g['purify.purified32'])
    e_p = (eng.purified_nchw().cpu() - ref).abs().max().item()
    codes = eng.acts['sg.latent'].t.view(rows, gspec.n_latent, -1).cpu()
    print(f'e4e defender vs reference purify golden [{precision}]: purified err {e_p:.2e} (range {ref.min().item():.2f}..{ref.max().item():.2f})')
    assert e_p < tol * max(1.0, ref.abs().max().item())
    cot = torch.from_numpy(g['purify.cot'])
    eng.dpurified.copy_(cot.to(DEV))
    eng.backward(from_logits=False, from_purified=True)
    gx = torch.from_numpy(g['purify.gx'])
    xr = x.clone().requires_grad_(True)
    (g0,) = torch.autograd.grad((D.e4e_purify(esd, espec, gsd, gspec, avg, xr, alphas, z, size) * cot).sum(), [xr])
    assert (g0 - gx).abs().max().item() < 5e-5 * gx.abs().max().item()
    assert_grad_given_engine_decisions(eng, lambda t: (D.e4e_purify(esd, espec, gsd, gspec, avg, t, alphas, z, size) * cot).sum(), x, eng.dx,
                                       1e-3, 'input gradient through the returned purified image', min_matched=20, golden=(gx, g0))
    # both cotangents at once = the sum of the two backward passes (the logits leg starts in the classifier)
    cl = torch.randn(rows, 2, generator=torch.Generator().manual_seed(0)).to(DEV)
    eng.dlogits.view(rows, -1).copy_(cl)
    eng.backward()
    d_logits = eng.dx.clone()
    eng.dpurified.copy_(cot.to(DEV))
    eng.backward(from_logits=True, from_purified=True)
    both = eng.dx.clone()
    eng.dpurified.copy_(cot.to(DEV))
    eng.backward(from_logits=False, from_purified=True)
    s_ = both.abs().max().item()
    assert (both - d_logits - eng.dx).abs().max().item() < 1e-4 * s_
    del codes


def test_latent_avg_is_estimated_when_the_checkpoint_has_none(tmp_path):
    """pSp.__load_latent_avg (psp.py:117-125): start_from_latent_avg without a stored 'latent_avg' -> mean of the mapping
    network over 10000 random latents; compared with the oracle's mapping network on the same draw and, statistically, with
    an independent draw"""
    from gen_adversarial_amd.defenses.loading_utils import estimate_latent_avg, load_E4EStyleGan
    from oracle import stylegan_oracle as S
    _, (esd, espec, gsd, gspec, avg, csd, cspec, alphas) = _small_e4e_defense(dry_run=True, device='cpu')
    est = estimate_latent_avg(gsd, gspec, DEV, n_latent=4096, seed=3)
    z = torch.empty(4096, gspec.style_dim, device=DEV).normal_(generator=torch.Generator(device=DEV).manual_seed(3)).cpu()
    ref = S.mapping_network(gsd, z).double().mean(dim=0, keepdim=True).float()
    assert est.shape == (1, gspec.style_dim)
    assert (est.cpu() - ref).abs().max().item() < 1e-3 * max(1.0, ref.abs().max().item())
    ck = {'state_dict': {**{'encoder.' + k: v for k, v in esd.items()}, **{'decoder.' + k: v for k, v in gsd.items()}},
          'opts': {'stylegan_size': gspec.size, 'start_from_latent_avg': True, 'encoder_type': 'Encoder4Editing'}}
    torch.save(ck, tmp_path / 'e4e_noavg.pt')
    w = load_E4EStyleGan(str(tmp_path / 'e4e_noavg.pt'), DEV)
    assert w.latent_avg.shape == (gspec.n_latent, gspec.style_dim)
    assert torch.equal(w.latent_avg[0], w.latent_avg[-1])                       # one mean latent, repeated over the indices
    spread = S.mapping_network(gsd, z).std(dim=0).max().item() / 100.0          # standard error of a 10000-sample mean
    assert (w.latent_avg[0] - ref[0]).abs().max().item() < 6 * spread + 1e-3


def test_bpda_on_the_e4e_defender():
    """Engine.backward(identity_purifier=True) for the e4e defender (classifier input in space-to-depth form): dx[image] = sum
    over its EoT replicas of the classifier's input gradient at the purified image; needs equal input and purified sizes"""
    from gen_adversarial_amd.engine import Engine
    from gen_adversarial_amd.e4e_spec import build_e4e_spec, init_e4e_state_dict
    from gen_adversarial_amd.resnet_spec import build_resnet_spec, init_resnet_state_dict
    from gen_adversarial_amd.stylegan_spec import build_stylegan_spec, init_stylegan_state_dict
    size = res = 64
    espec, esd = build_e4e_spec(size, 4, (1, 1, 1, 1)), init_e4e_state_dict(size, 4, 3, (1, 1, 1, 1))
    gspec = build_stylegan_spec(size, width_div=8, style_dim=espec.style_dim)
    gsd = init_stylegan_state_dict(gspec, 4)
    cspec, csd = build_resnet_spec(2, 8, (1, 1, 1, 1)), init_resnet_state_dict(2, 8, 5, (1, 1, 1, 1))
    rows, rep = 6, 3
    eng = Engine.bare(rows, device=DEV, rep=rep, resolution=(3, res, res), alphas=[0.1] * gspec.n_latent, share_encoder=True)
    eng.build_e4e_defense(esd, espec, gsd, gspec, None, csd, cspec, pool_to=size)
    assert eng.bpda is not None
    g = torch.Generator().manual_seed(8)
    eng.x_in.copy_(torch.rand(rows // rep, 3, res, res, generator=g).to(DEV))
    eng.eps[0].copy_(torch.randn(eng.eps[0].shape, generator=g).to(DEV))
    eng.forward()
    eng.dlogits.view(rows, -1).copy_(torch.randn(rows, 2, generator=g).to(DEV))
    eng.backward()                                            # full white-box gradient: also leaves d loss / d purified in place
    full = eng.dx.clone()
    t = eng.purified_s2d.g
    n, h2, w2, _ = t.shape
    dp = t.view(n, h2, w2, 2, 2, 8)[..., :3].permute(0, 5, 1, 3, 2, 4).reshape(n, 3, 2 * h2, 2 * w2)      # as purified_nchw() reads .t
    eng.backward(identity_purifier=True)
    want = dp.view(rows // rep, rep, 3, res, res).sum(dim=1)
    assert torch.allclose(eng.dx, want, atol=1e-6 * max(1.0, want.abs().max().item()))
    assert not torch.allclose(eng.dx, full, atol=1e-3 * full.abs().max().item())       # BPDA is not the white-box gradient
    small = Engine.bare(2, device=DEV, rep=1, resolution=(3, res, res), alphas=[0.1] * gspec.n_latent)
    small.build_e4e_defense(esd, espec, gsd, gspec, None, csd, cspec, pool_to=32)       # purified smaller than the input: no identity
    assert small.bpda is None
    with pytest.raises(RuntimeError):
        small.backward(identity_purifier=True)
