"""
GPU: the drop-in surface (load(args) -> EoTWrapper(NVAEDefenseModel(CelebaIdentityClassifier))) driven the way the
reference's driver and attacks drive it (src/experiments/test_defense.py:133-135,225; src/attacks/untargeted.py:146,
529-535), checked against the oracle with explicit noise.
"""
from argparse import Namespace

import numpy as np
import pytest
import torch
import yaml

pytestmark = pytest.mark.gpu

if not torch.cuda.is_available():
    pytest.skip('needs a GPU', allow_module_level=True)

from gen_adversarial_amd.experiments.load_defense import load   # noqa: E402
from gen_adversarial_amd.nvae_spec import build_spec, nvae_checkpoint   # noqa: E402
from gen_adversarial_amd.vgg_spec import build_vgg_spec, init_vgg_state_dict   # noqa: E402
from oracle import defender_oracle as D   # noqa: E402

DEV = 'cuda:0'
CFG = {'initial_channels': 8, 'num_pre-post_process_blocks': 1, 'num_pre-post_process_cells': 2, 'num_scales': 3,
       'num_groups_per_scale': 2, 'is_adaptive': False, 'min_groups_per_scale': 1, 'num_cells_per_group': 1,
       'num_latent_per_group': 4, 'num_logistic_mixtures': 10, 'num_nf_cells': None}
RES = (3, 64, 64)
EOT = 4


@pytest.fixture(scope='module')
def setup(tmp_path_factory):
    d = tmp_path_factory.mktemp('ckpt')
    ck = nvae_checkpoint(CFG, RES, seed=5)
    torch.save(ck, d / 'nvae.pt')
    vsd = init_vgg_state_dict(100, 16, seed=6)
    torch.save({'state_dict': vsd}, d / 'vgg.pt')
    n_groups = len(build_spec(CFG, RES).groups)
    alphas = [round(i / (n_groups - 1), 3) for i in range(n_groups)]
    y = {'classifier_path': str(d / 'vgg.pt'), 'autoencoder_path': str(d / 'nvae.pt'), 'interpolation_alphas': alphas,
         'alpha_attenuation': 0.7, 'initial_noise_eps': 2.0, 'gaussian_blur_input': False}
    with open(d / 'cfg.yaml', 'w') as f:
        yaml.safe_dump(y, f)
    args = Namespace(config=str(d / 'cfg.yaml'), experiment='ids', defense_type='ours', eot_steps=EOT, device=DEV)
    args, model = load(args)
    return args, model, ck, vsd, [a * 0.7 for a in alphas]


def _oracle(setup, x, eps, noise, rep):
    _, _, ck, vsd, alphas = setup
    spec = build_spec(CFG, RES)
    sd = ck['state_dict_temp=0.6']
    return D.nvae_defender(sd, spec, vsd, build_vgg_spec(100, 16), x.repeat_interleave(rep, dim=0), alphas, eps, noise, 2.0)


def test_eot_logits_grad_and_retain_graph(setup):
    args, model, *_ = setup
    assert args.image_size == 64 and set(args.attacks) == {'deepfool', 'c&w', 'autoattack'}
    spec = build_spec(CFG, RES)
    g = torch.Generator().manual_seed(0)
    x = torch.rand(1, *RES, generator=g)
    eps = [torch.randn(EOT, 4, gs.res, gs.res, generator=g) for gs in spec.groups]
    noise = torch.randn(EOT, *RES, generator=g)
    xr = x.clone().requires_grad_(True)
    lo, _ = _oracle(setup, xr, eps, noise, EOT)
    lo = lo.mean(dim=0, keepdim=True)
    g0, g3 = (torch.autograd.grad(lo[0, k], [xr], retain_graph=True)[0] for k in (0, 3))

    model.model.fixed_noise([e.to(DEV) for e in eps], noise.to(DEV))
    xd = x.to(DEV).requires_grad_(True)
    out = model(xd)
    assert out.shape == (1, 100)
    assert (out.cpu() - lo).abs().max().item() < 2e-4
    # per-class backward passes on one forward, as DeepFool / FAB do
    h0 = torch.autograd.grad(out[0, 0], [xd], retain_graph=True)[0]
    h3 = torch.autograd.grad(out[0, 3], [xd], retain_graph=True)[0]
    assert (h0.cpu() - g0).abs().max().item() < 2e-4 * max(1.0, g0.abs().max().item())
    assert (h3.cpu() - g3).abs().max().item() < 2e-4 * max(1.0, g3.abs().max().item())
    # a second forward in between must not corrupt a later backward of the first graph
    with torch.no_grad():
        model(torch.rand(1, *RES, device=DEV))
    model.model.fixed_noise([e.to(DEV) for e in eps], noise.to(DEV))
    h0b = torch.autograd.grad(out[0, 0], [xd])[0]
    assert (h0b - h0).abs().max().item() < 1e-6
    model.model.fixed_noise(None, None)


def test_get_purified_and_preds(setup):
    args, model, *_ = setup
    x = torch.rand(2, *RES, device=DEV)
    logits, purified = model.model(x, preds_only=False)
    assert logits.shape == (2, 100) and purified.shape == (2, *RES)
    assert 0.0 <= purified.min().item() and purified.max().item() <= 1.0
    p = model.get_purified(x)
    assert p.shape == (2, *RES)
    # purify() alone has no input-noise stage (eps=2 would move the image visibly)
    spec = build_spec(CFG, RES)
    eps = [torch.randn(2, 4, gs.res, gs.res, device=DEV) for gs in spec.groups]
    model.model.fixed_noise(eps, None)
    a = model.model.purify(x)
    b = model.model.purify(x)
    assert torch.equal(a, b)
    model.model.fixed_noise(None, None)


def test_alpha_overwrite_is_picked_up(setup):
    """alpha learning assigns a new list to .interpolation_alphas (alpha_learning/common_utils.py:88)."""
    args, model, *_ = setup
    spec = build_spec(CFG, RES)
    x = torch.rand(1, *RES, device=DEV)
    eps = [torch.randn(EOT, 4, gs.res, gs.res, device=DEV) for gs in spec.groups]
    noise = torch.randn(EOT, *RES, device=DEV)
    model.model.fixed_noise(eps, noise)
    old = list(model.model.interpolation_alphas)
    with torch.no_grad():
        l1 = model(x).clone()
        model.model.interpolation_alphas = [0.0 for _ in old]
        l2 = model(x).clone()
        model.model.interpolation_alphas = old
        l3 = model(x).clone()
    assert not torch.equal(l1, l2) and torch.equal(l1, l3)
    model.model.fixed_noise(None, None)


def test_pgd_attack_protocol(setup):
    args, model, *_ = setup
    x = torch.rand(1, *RES, device=DEV)
    with torch.no_grad():
        label = model(x).argmax(dim=1)
    atk = args.pgd
    atk.steps = 3
    success, bound, adv = atk(x, label, model)
    assert isinstance(success, bool) and isinstance(bound, float) and adv.shape == x.shape
    assert bound <= 8.0 / 255.0 + 1e-6 and 0.0 <= adv.min().item() and adv.max().item() <= 1.0


def test_batched_pgd_through_the_driver(setup):
    """experiments/test_defense.evaluate_shard with batch_images > 1: several images x EoT rows per defender call"""
    from gen_adversarial_amd.attacks.pgd import PGDLinf
    from gen_adversarial_amd.experiments.test_defense import FAILED, evaluate_shard
    args, model, *_ = setup
    x = torch.rand(5, *RES, device=DEV, generator=torch.Generator(device=DEV).manual_seed(3))
    with torch.no_grad():
        y = model(x).argmax(dim=1)
    table = evaluate_shard(model, {'pgd': PGDLinf(eps=8.0 / 255.0, step_size=2.0 / 255.0, steps=3)}, x, y, batch_images=4)
    assert table.shape == (5, 2) and set(table[:, 0].tolist()) <= {0.0, 1.0}
    d = table[:, 1]
    assert bool(((d == FAILED) | ((d >= 0) & (d <= 8.0 / 255.0 + 1e-6))).all())


def test_reference_attacks_drive_the_hip_defender(setup):
    """DeepFool (per-class backward on one forward), APGD and FGSM against the stochastic HIP defender: protocol and
    invariants (the numbers themselves are pinned on the CPU against the reference, tests/test_attacks_cpu.py)."""
    from gen_adversarial_amd.attacks.l2_attacks import APGDAttack, DeepFool, FGSM
    args, model, *_ = setup
    x = torch.rand(1, *RES, device=DEV)
    with torch.no_grad():
        label = model(x).argmax(dim=1)
    s, b, adv = DeepFool(num_classes=4, overshoot=0.02, max_iter=3)(x, label, model)
    assert isinstance(s, bool) and adv.shape == x.shape and torch.isfinite(adv).all()
    s, b, adv = APGDAttack(n_iter=3, rho=0.75, max_bound=0.5, ce_loss=True)(x, label, model)
    assert b <= 0.5 + 1e-4 and 0.0 <= adv.min().item() and adv.max().item() <= 1.0
    s, b, adv = APGDAttack(n_iter=2, rho=0.75, max_bound=0.5, ce_loss=False)(x, label, model)
    assert b <= 0.5 + 1e-4
    s, b, adv = FGSM(l2_bound=2.0)(x, label, model)
    assert abs((adv - x).flatten().norm().item()) <= 2.0 + 1e-3


def test_base_classifier_path(setup, tmp_path):
    """defense_type 'base' (configs/no_defense_ids.yaml): classifier only, differentiable."""
    args, _, ck, vsd, _ = setup
    y = {'classifier_path': yaml.safe_load(open(args.config))['classifier_path']}
    with open(tmp_path / 'base.yaml', 'w') as f:
        yaml.safe_dump(y, f)
    a2, clf = load(Namespace(config=str(tmp_path / 'base.yaml'), experiment='ids', defense_type='base', eot_steps=1, device=DEV))
    g = torch.Generator().manual_seed(1)
    x = torch.rand(3, *RES, generator=g)
    xr = x.clone().requires_grad_(True)
    ref = D.classifier_call(vsd, build_vgg_spec(100, 16), xr)
    (gr,) = torch.autograd.grad(ref[:, 7].sum(), [xr])
    xd = x.to(DEV).requires_grad_(True)
    out = clf(xd)
    assert (out.cpu() - ref).abs().max().item() < 2e-4
    (gd,) = torch.autograd.grad(out[:, 7].sum(), [xd])
    diff = (gd.cpu() - gr)
    from gradcheck import assert_grad_given_engine_decisions
    assert_grad_given_engine_decisions(clf._engine(3, 1), lambda t: D.classifier_call(vsd, build_vgg_spec(100, 16), t)[:, 7].sum(),
                                       x, gd, 1e-3, 'classifier input gradient', min_matched=8)
    assert (diff.norm() / gr.norm()).item() < 2e-2          # secondary (tie-dependent elements included)
    assert clf.get_purified(x) is x
    with pytest.raises(NotImplementedError):
        load(Namespace(config=args.config, experiment='imagenet', defense_type='ours', eot_steps=1, device=DEV))


def test_ablation_defenders_and_alpha_objective(setup, tmp_path):
    """defense_type 'ablation' (configs/ablation_{noise,blur}_ids.yaml) and the alpha-learning objective."""
    from gen_adversarial_amd.experiments.alpha_learning.common_utils import AlphaEvaluator, get_cosine_alphas, random_search
    args, model, ck, vsd, _ = setup
    cpath = yaml.safe_load(open(args.config))
    x = torch.rand(2, *RES, generator=torch.Generator().manual_seed(5))
    for kind in ('noise', 'blur'):
        with open(tmp_path / f'{kind}.yaml', 'w') as f:
            yaml.safe_dump({'classifier_path': cpath['classifier_path'], 'type': kind}, f)
        a, m = load(Namespace(config=str(tmp_path / f'{kind}.yaml'), experiment='ids', defense_type='ablation', eot_steps=2, device=DEV))
        p = m.get_purified(x.to(DEV))
        assert p.shape == x.shape and 0 <= p.min().item() and p.max().item() <= 1
        if kind == 'blur':
            assert (p.cpu() - D.apply_gaussian_blur(x)).abs().max().item() < 1e-5
            ref = D.classifier_call(vsd, build_vgg_spec(100, 16), D.apply_gaussian_blur(x))
            assert (m.model(x.to(DEV)).cpu() - ref).abs().max().item() < 2e-4
        else:
            assert abs((p.cpu() - x).flatten(1).norm(dim=1).max().item() - 2.0) < 0.2      # L2 = eps before the clamp
        xg = x[:1].to(DEV).requires_grad_(True)
        (g,) = torch.autograd.grad(m(xg)[0, 0], [xg])
        assert torch.isfinite(g).all() and g.abs().max().item() > 0
    assert get_cosine_alphas(4)[-1] == pytest.approx(1.0)
    ev_args = Namespace(classifier_type='vgg-11', classifier_path=cpath['classifier_path'],
                        autoencoder_path=cpath['autoencoder_path'], initial_alphas=[0.] * 6, eot_steps=2)
    imgs = torch.rand(5, *RES)
    ev = AlphaEvaluator(ev_args, DEV, images=imgs, labels=torch.zeros(5, dtype=torch.long), batch_images=2)
    acc = ev.objective_function(torch.full((6,), 0.5))
    assert 0.0 <= acc <= 1.0
    al, ac = random_search(ev, 2, seed=1)
    assert al.shape == (2, 6) and ac.shape == (2, 1)


def test_batched_deepfool_and_fab_equal_the_one_image_protocol(setup):
    """SURVEY.md §8 row f1 on the HIP defender: DeepFool / FAB over B = 3 images in one call (3 x EoT rows per defender run,
    one backward pass per class rank for all images) against the reference's one-image-at-a-time protocol.  The defender is
    made deterministic for the comparison (alpha = 0 everywhere: no latent noise enters; the input noise is fixed), so both
    runs see the same function."""
    from gen_adversarial_amd.attacks.l2_attacks import DeepFool, FABAttack
    args, model, ck, vsd, alphas = setup
    old = list(model.model.interpolation_alphas)
    g = torch.Generator().manual_seed(3)
    x = torch.rand(3, *RES, generator=g).to(DEV)
    spec = build_spec(CFG, RES)
    try:
        model.model.interpolation_alphas[:] = [0.0] * len(old)

        def fix(rows):          # equal input noise for every image: row r of any call uses draw r % EOT
            n = torch.randn(EOT, *RES, generator=torch.Generator().manual_seed(9)).to(DEV)
            eps = [torch.zeros(rows, 4, gs.res, gs.res, device=DEV) for gs in spec.groups]
            model.model.fixed_noise(eps, n.repeat(rows // EOT, 1, 1, 1))
        fix(3 * EOT)
        with torch.no_grad():
            labels = model(x).argmax(dim=1)
        for mk in (lambda: DeepFool(num_classes=4, overshoot=0.02, max_iter=4), lambda: FABAttack(n_iter=3, alpha_max=0.1, eta=1.05, beta=0.9)):
            fix(3 * EOT)
            s, b, a = mk()(x, labels, model)
            for i in range(3):
                fix(EOT)
                s1, b1, a1 = mk()(x[i:i + 1], labels[i:i + 1], model)
                assert bool(s[i]) == bool(s1), (type(mk()).__name__, i)
                if bool(s1):
                    assert abs(float(b[i]) - float(b1)) <= 2e-3 * max(1.0, float(b1)), (type(mk()).__name__, i, float(b[i]), float(b1))
                assert (a[i:i + 1] - a1).abs().max().item() < 2e-3
    finally:
        model.model.interpolation_alphas[:] = old
        model.model.fixed_noise(None, None)


def test_bpda_gradient_is_the_classifier_gradient_at_the_purified_image(setup):
    """BPDA (north_star "PGD-40 + BPDA"; new code on the reference's protocol): forward exact, backward with the purifier
    counted as the identity: dx = sum over the EoT replicas of d loss / d purified — checked against the oracle's classifier
    gradient at the oracle's purified images; and PGD-Linf(bpda=True) runs on it"""
    from gen_adversarial_amd.attacks.pgd import PGDLinf, bpda
    args, model, ck, vsd, alphas = setup
    spec = build_spec(CFG, RES)
    g = torch.Generator().manual_seed(4)
    x = torch.rand(2, *RES, generator=g)
    eps = [torch.randn(2 * EOT, 4, gs.res, gs.res, generator=g) for gs in spec.groups]
    noise = torch.randn(2 * EOT, *RES, generator=g)
    with torch.no_grad():
        _, purified = _oracle(setup, x, eps, noise, EOT)
    pr = purified.clone().requires_grad_(True)
    logits = D.classifier_call(vsd, build_vgg_spec(100, 16), pr).view(2, EOT, -1).mean(dim=1)
    loss = torch.nn.functional.cross_entropy(logits, logits.argmax(dim=1).detach(), reduction='sum')
    (gp,) = torch.autograd.grad(loss, [pr])
    ref = gp.view(2, EOT, *RES).sum(dim=1)

    model.model.fixed_noise([e.to(DEV) for e in eps], noise.to(DEV))
    xd = x.to(DEV).requires_grad_(True)
    with bpda(model):
        out = model(xd)
        l2 = torch.nn.functional.cross_entropy(out, out.argmax(dim=1).detach(), reduction='sum')
        (gd,) = torch.autograd.grad(l2, [xd])
    assert model.model.bpda is False
    assert (out.detach().cpu() - logits.detach()).abs().max().item() < 1e-3
    rel = ((gd.cpu() - ref).norm() / ref.norm()).item()
    print(f'BPDA gradient vs oracle classifier gradient at the purified image: relL2 {rel:.2e}')
    assert rel < 2e-2
    from gradcheck import assert_grad_given_engine_decisions
    eng = model.model._engine(2 * EOT, EOT)

    def lossfn(p_):
        lg = D.classifier_call(vsd, build_vgg_spec(100, 16), p_).view(2, EOT, -1).mean(dim=1)
        return torch.nn.functional.cross_entropy(lg, logits.argmax(dim=1).detach(), reduction='sum')
    with torch.no_grad():
        eng_purified = eng.purified.cpu()
    gfull = eng.acts['purified_nhwc'].g[..., :3].permute(0, 3, 1, 2)
    assert_grad_given_engine_decisions(eng, lossfn, eng_purified, gfull, 1e-3, 'classifier gradient at the purified image', min_matched=8)
    assert torch.allclose(gd, gfull.view(2, EOT, *RES).sum(dim=1), atol=1e-6)
    model.model.fixed_noise(None, None)
    s, b, adv = PGDLinf(eps=8 / 255, step_size=2 / 255, steps=3, bpda=True)(x.to(DEV), out.argmax(dim=1), model)
    assert adv.shape == x.shape and float(torch.as_tensor(b).max()) <= 8 / 255 + 1e-6
    with pytest.raises(TypeError):
        with bpda(torch.nn.Linear(2, 2)):
            pass


def test_alpha_objective_verdicts_equal_the_oracles_under_fixed_noise(setup):
    """SURVEY.md §8 row f2 (AlphaEvaluator.objective_function, src/experiments/alpha_learning/common_utils.py:81-103): with the
    latent noise fixed, every image's verdict (EoT-mean prediction == label) and the accuracy equal the oracle's"""
    from gen_adversarial_amd.experiments.alpha_learning.common_utils import AlphaEvaluator
    args, model, ck, vsd, _ = setup
    cpath = yaml.safe_load(open(args.config))
    spec = build_spec(CFG, RES)
    eot, bi, n = 2, 2, 6
    g = torch.Generator().manual_seed(11)
    imgs = torch.rand(n, *RES, generator=g)
    eps = [torch.randn(bi * eot, 4, gs.res, gs.res, generator=g) for gs in spec.groups]
    cand = torch.rand(len(spec.groups), generator=g)
    alphas = [float(a) * 0.7 for a in cand]
    sd = ck['state_dict_temp=0.6']
    vspec = build_vgg_spec(100, 16)
    mean_logits = []
    for i in range(0, n, bi):
        x = imgs[i:i + bi].repeat_interleave(eot, dim=0)
        lg, _ = D.nvae_defender(sd, spec, vsd, vspec, x, alphas, eps, torch.ones_like(x), 0.0)     # a draw is made even at eps 0 (:132)
        mean_logits.append(lg.view(bi, eot, -1).mean(dim=1))
    mean_logits = torch.cat(mean_logits)
    labels = mean_logits.argmax(dim=1).clone()
    labels[::2] = (labels[::2] + 1) % 100                      # half of the set "stays fooled"
    ev_args = Namespace(classifier_type='vgg-11', classifier_path=cpath['classifier_path'], autoencoder_path=cpath['autoencoder_path'],
                        initial_alphas=[0.] * len(spec.groups), eot_steps=eot)
    ev = AlphaEvaluator(ev_args, DEV, images=imgs, labels=labels, batch_images=bi)
    ev.defense_model.model.fixed_noise([e.to(DEV) for e in eps], None)
    hits = ev.per_image_verdicts(cand)
    ref = mean_logits.argmax(dim=1) == labels
    top2 = mean_logits.topk(2, dim=1).values
    assert (top2[:, 0] - top2[:, 1]).min().item() > 1e-3          # no verdict sits on a tie: the comparison is exact
    assert hits.cpu().tolist() == ref.tolist()
    assert ev.objective_function(cand) == pytest.approx(ref.float().mean().item())


def test_noise_ablation_with_injected_noise_matches_the_oracle(setup, tmp_path):
    """SURVEY.md §8 row f4 (GaussianNoiseDefenseModel, src/defenses/ablations/models.py:13-39): purified image and logits with
    the N(0,1) draw injected, against MLVGMDefenseModel.add_gaussian_noise's restatement + the classifier oracle"""
    args, model, ck, vsd, _ = setup
    cpath = yaml.safe_load(open(args.config))
    with open(tmp_path / 'noise.yaml', 'w') as f:
        yaml.safe_dump({'classifier_path': cpath['classifier_path'], 'type': 'noise'}, f)
    eot = 3
    a, m = load(Namespace(config=str(tmp_path / 'noise.yaml'), experiment='ids', defense_type='ablation', eot_steps=eot, device=DEV))
    g = torch.Generator().manual_seed(6)
    x = torch.rand(2, *RES, generator=g)
    noise = torch.randn(2 * eot, *RES, generator=g)
    xr = x.clone().requires_grad_(True)
    noisy = D.add_gaussian_noise(xr.repeat_interleave(eot, dim=0), noise, 2.0)
    ref = D.classifier_call(vsd, build_vgg_spec(100, 16), noisy).view(2, eot, -1).mean(dim=1)
    m.model.fixed_noise(None, noise.to(DEV))
    xd = x.to(DEV).requires_grad_(True)
    out = m(xd)
    assert (out.detach().cpu() - ref.detach()).abs().max().item() < 2e-4
    eng = m.model._engine(2 * eot, eot)
    assert (eng.input_image_nchw().cpu() - noisy.detach()).abs().max().item() < 1e-5
    (gd,) = torch.autograd.grad(out[:, 3].sum(), [xd])
    from gradcheck import assert_grad_given_engine_decisions
    assert_grad_given_engine_decisions(
        eng, lambda t: D.classifier_call(vsd, build_vgg_spec(100, 16), D.add_gaussian_noise(t.repeat_interleave(eot, dim=0), noise, 2.0))
        .view(2, eot, -1).mean(dim=1)[:, 3].sum(), x, gd, 1e-3, 'noise ablation input gradient', min_matched=8)
    m.model.fixed_noise(None, None)
    p = m.get_purified(x[:1].to(DEV))
    assert abs((p.cpu() - x[:1]).flatten(1).norm(dim=1).item() - 2.0) < 0.2


def test_class_jacobian_k_cotangent_plan_equals_one_backward_per_class(setup, monkeypatch):
    """SURVEY.md §8 row f1, the one-pass multi-class VJP: the per-class input gradients DeepFool / FAB need
    (src/attacks/untargeted.py:526-560, :605-635: one `.backward(retain_graph=True)` per class) from the engine's K-cotangent
    backward plan — ONE forward, ceil(columns / K) backward replays — against one autograd backward per class on the same
    defender under the same noise.  With input noise configured (eps 2.0: clamp masks per replica, no shared encoder)."""
    import math
    from gen_adversarial_amd.attacks.l2_attacks import ClassJacobian
    from gen_adversarial_amd.engine import Engine
    args, model, ck, vsd, alphas = setup
    spec = build_spec(CFG, RES)
    B = 2
    g = torch.Generator().manual_seed(17)
    x = torch.rand(B, *RES, generator=g).to(DEV)
    eps = [torch.randn(B * EOT, 4, gs.res, gs.res, generator=g).to(DEV) for gs in spec.groups]
    noise = torch.randn(B * EOT, *RES, generator=g).to(DEV)
    classes = torch.stack([torch.randperm(100, generator=g)[:5] for _ in range(B)]).to(DEV)
    calls = {'n': 0}
    real = Engine.backward

    def counting(self, *a, **k):
        calls['n'] += 1
        return real(self, *a, **k)
    monkeypatch.setattr(Engine, 'backward', counting)
    try:
        for cols, n_cols in ((classes, 5), (None, 100)):
            model.model.fixed_noise(eps, noise)
            calls['n'] = 0
            fast = ClassJacobian(model, x, cols)
            assert fast._fast is not None, 'the HIP defender must offer its K-cotangent plan'
            K = fast._fast.eng.cot_rep
            assert K == min(n_cols, 512 // (B * EOT))
            g_fast = fast.grads()
            assert calls['n'] == math.ceil(n_cols / K), (calls['n'], n_cols, K)       # <= n_cols / K backward replays
            monkeypatch.setattr(type(model.model), 'jacobian_cot_rows', 0)            # no K-cotangent plan: autograd per class
            calls['n'] = 0
            slow = ClassJacobian(model, x, cols)
            assert slow._fast is None
            g_slow = slow.grads()
            assert calls['n'] == n_cols
            monkeypatch.undo()
            monkeypatch.setattr(Engine, 'backward', counting)
            assert g_fast.shape == g_slow.shape == (B, n_cols, *RES)
            e_l = (fast.logits - slow.logits).abs().max().item()
            scale = g_slow.abs().amax(dim=(2, 3, 4), keepdim=True).clamp_min(1e-30)
            e_g = ((g_fast - g_slow).abs() / scale).max().item()
            print(f'   K-cotangent plan (K = {K}, {n_cols} columns): logits {e_l:.2e}, gradients {e_g:.2e} of each column\'s max')
            assert e_l < 1e-5 and e_g < 1e-4
    finally:
        model.model.fixed_noise(None, None)


@pytest.mark.parametrize('share', [False, True])
def test_k_cotangent_engine_equals_repeated_backward(setup, share):
    """Engine(cot_rep = K) against K backward replays of the plain engine, without input noise: share = True runs the encoder once
    per image (its cotangents then meet in ga_rep_sum with K folded into the row), share = False the literal repeat; dense random
    cotangents (not only one-hot class seeds), gradient from the logits and from the purified image."""
    from gen_adversarial_amd.engine import Engine, WeightStore
    _, _, ck, vsd, alphas = setup
    spec = build_spec(CFG, RES)
    sd = ck['state_dict_temp=0.6']
    vspec = build_vgg_spec(100, 16)
    rows, rep, K = 8, 4, 3
    store = WeightStore(DEV)
    g = torch.Generator().manual_seed(23)
    imgs = torch.rand(rows // rep, *RES, generator=g).to(DEV)
    eps = [torch.randn(rows, 4, gs.res, gs.res, generator=g).to(DEV) for gs in spec.groups]
    cot = torch.randn(rows, K, 100, generator=g).to(DEV)
    cot_img = torch.randn(rows, K, *RES, generator=g).to(DEV)
    engs = {k: Engine(sd, CFG, RES, vsd, vspec, rows=rows, rep=rep, alphas=alphas, device=DEV, store=store, share_encoder=share,
                      cot_rep=k) for k in (1, K)}
    for e in engs.values():
        e.x_in.copy_(imgs)
        for dst, src in zip(e.eps, eps):
            dst.copy_(src)
        e.forward()
    assert torch.equal(engs[1].logits, engs[K].logits)
    engs[K].dlogits.view(rows, K, 100).copy_(cot)
    engs[K].backward()
    got = engs[K].dx.view(rows // rep, K, *RES).clone()
    engs[K].dpurified.view(rows, K, *RES).copy_(cot_img)
    engs[K].backward(from_logits=False, from_purified=True)
    got_img = engs[K].dx.view(rows // rep, K, *RES).clone()
    for k in range(K):
        engs[1].dlogits.view(rows, 100).copy_(cot[:, k])
        engs[1].backward()
        ref = engs[1].dx.clone()
        e = (got[:, k] - ref).abs().max().item() / ref.abs().max().item()
        engs[1].dpurified.copy_(cot_img[:, k])
        engs[1].backward(from_logits=False, from_purified=True)
        ref_img = engs[1].dx.clone()
        e_img = (got_img[:, k] - ref_img).abs().max().item() / ref_img.abs().max().item()
        print(f'   cotangent {k}: from logits {e:.2e}, from the purified image {e_img:.2e} (relative to max |g|)')
        assert e < 1e-5 and e_img < 1e-5
