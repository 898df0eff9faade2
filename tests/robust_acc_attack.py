"""
Robust accuracy under one of the REFERENCE's attacks (APGD-CE at a fixed L2 bound, src/attacks/untargeted.py:37-243), HIP path
against the CPU oracle, with a paired confidence interval (VERDICT r03 "next round" #3, weak #1; BASELINE.json: "robust-accuracy
within +-0.1 % of reference").  TEST INFRASTRUCTURE: imports the oracle as the checker; used by tests/, tools/robust_acc_delta.py
and bench.py's robust_accuracy_delta leg, never by the product path.

Protocol: N images x EoT `eot` through the reduced NVAE + VGG defender of tests/robust_acc.py.  The SAME attack object
(gen_adversarial_amd.attacks.l2_attacks.APGDAttack, batched over images, pinned against the reference's class in
tests/test_attacks_cpu.py) drives (i) the HIP defender behind the drop-in API and (ii) oracle.defender_oracle.EoTDefenderOracle.
Every random draw is pinned and equal on both sides: the attack's start noise, and for the defender a fresh latent draw PER CALL
(call t of an attack run uses draw t on both sides: the reference draws fresh noise in every forward, models.py:206,250).
Labels = the oracle's clean prediction.  Verdict of an image: robust = the attack's own success flag is False.  Each implementation
follows its own trajectory.  Reported: both robust accuracies, the paired difference with its 95 % interval
(delta = (n_hip_only - n_oracle_only) / N, se = sqrt(n_d - (n_hip_only - n_oracle_only)^2 / N) / N with n_d discordant pairs), and
the same-input verdicts (each implementation judges the OTHER's adversarial examples under one more pinned draw).
"""
import math
import os
import time

import torch

from robust_acc import CFG, RES


class _Scheduled(torch.nn.Module):
    """a defender whose latent noise for call t is draw t of a seeded CPU stream (identical on both sides)"""

    def __init__(self, inner, set_noise, spec, rows, seed):
        super().__init__()
        self.inner, self.set_noise, self.spec, self.rows, self.seed, self.t = inner, set_noise, spec, rows, seed, 0

    def draw(self, t):
        g = torch.Generator().manual_seed(self.seed * 100003 + t)
        return [torch.randn(self.rows, self.spec.num_latent, gs.res, gs.res, generator=g) for gs in self.spec.groups]

    def forward(self, x):
        self.set_noise(self.draw(self.t))
        self.t += 1
        return self.inner(x)


def robust_accuracy_under_attack(device='cuda:0', n_images=256, eot=2, n_iter=5, bound=2.0, seed=0, chunk_images=64, threads=None,
                                 tmpdir=None, progress=False, model_seed=0):
    """seed: images, start noise and latent draws; model_seed: the random weights (runs that are pooled share the model: another
    random model has other margins — the one with model_seed 1 is robust on every image at this bound, which adds nothing)"""
    import tempfile
    from argparse import Namespace
    import yaml
    from gen_adversarial_amd.attacks.l2_attacks import APGDAttack
    from gen_adversarial_amd.experiments.load_defense import load
    from gen_adversarial_amd.nvae_spec import build_spec, nvae_checkpoint
    from gen_adversarial_amd.vgg_spec import build_vgg_spec, init_vgg_state_dict
    from oracle import defender_oracle as D
    if threads is None:
        try:
            threads = min(16, len(os.sched_getaffinity(0)))
        except AttributeError:
            threads = min(16, os.cpu_count() or 1)
    torch.set_num_threads(threads)
    spec = build_spec(CFG, RES)
    ng = len(spec.groups)
    alphas = [i / (ng - 1) for i in range(ng)]
    d = tmpdir or tempfile.mkdtemp(prefix='ga_racc_')
    ck = nvae_checkpoint(CFG, RES, seed=model_seed + 5)
    vsd = init_vgg_state_dict(100, 16, seed=model_seed + 6)
    torch.save(ck, os.path.join(d, 'nvae.pt'))
    torch.save({'state_dict': vsd}, os.path.join(d, 'vgg.pt'))
    with open(os.path.join(d, 'cfg.yaml'), 'w') as f:
        yaml.safe_dump({'classifier_path': os.path.join(d, 'vgg.pt'), 'autoencoder_path': os.path.join(d, 'nvae.pt'),
                        'interpolation_alphas': alphas, 'alpha_attenuation': 0.7, 'initial_noise_eps': 0.0,
                        'gaussian_blur_input': False}, f)
    _, model = load(Namespace(config=os.path.join(d, 'cfg.yaml'), experiment='ids', defense_type='ours', eot_steps=eot, device=device))
    chunk_images = min(chunk_images, n_images)
    assert n_images % chunk_images == 0
    rows = chunk_images * eot
    oracle = D.EoTDefenderOracle(ck['state_dict_temp=0.6'], spec, vsd, build_vgg_spec(100, 16), eot, [a * 0.7 for a in alphas],
                                 None, None, noise_eps=0.0)

    def set_cpu(eps):
        oracle.eps = eps

    def set_hip(eps):
        model.model.fixed_noise([e.to(device) for e in eps], None)

    g = torch.Generator().manual_seed(seed)
    keep = {k: [] for k in ('hip', 'cpu', 'cpu_on_hip', 'hip_on_cpu', 'l2_hip', 'l2_cpu')}
    t_cpu = t_hip = 0.0
    for c in range(n_images // chunk_images):
        x = torch.rand(chunk_images, *RES, generator=g)
        init = torch.randn(chunk_images, *RES, generator=g)
        s_cpu = _Scheduled(oracle, set_cpu, spec, rows, seed * 1000 + c)
        s_hip = _Scheduled(model, set_hip, spec, rows, seed * 1000 + c)
        with torch.no_grad():
            set_cpu(s_cpu.draw(10 ** 6))
            labels = oracle(x).argmax(dim=1)
        t = time.time()
        ok_c, b_c, adv_c = APGDAttack(n_iter=n_iter, rho=0.75, max_bound=bound, ce_loss=True)(x, labels, s_cpu, init_noise=init)
        t_cpu += time.time() - t
        t = time.time()
        ok_h, b_h, adv_h = APGDAttack(n_iter=n_iter, rho=0.75, max_bound=bound, ce_loss=True)(x.to(device), labels.to(device), s_hip,
                                                                                           init_noise=init.to(device))
        torch.cuda.synchronize()
        t_hip += time.time() - t
        keep['cpu'].append(~torch.as_tensor(ok_c).view(-1))
        keep['hip'].append(~torch.as_tensor(ok_h).view(-1).cpu())
        keep['l2_cpu'].append(torch.as_tensor(b_c).view(-1))
        keep['l2_hip'].append(torch.as_tensor(b_h).view(-1).cpu())
        # the two implementations as judges of the same adversarial examples under one more pinned draw
        judge = s_cpu.draw(10 ** 6 + 1)
        with torch.no_grad():
            set_cpu(judge)
            set_hip(judge)
            keep['cpu_on_hip'].append((oracle(adv_h.cpu()).argmax(dim=1) == labels, model(adv_h).argmax(dim=1).cpu() == labels))
            keep['hip_on_cpu'].append((model(adv_c.to(device)).argmax(dim=1).cpu() == labels, oracle(adv_c).argmax(dim=1) == labels))
        if progress:
            import sys
            print(f'[robust_acc_attack] {(c + 1) * chunk_images} / {n_images} images, oracle {t_cpu:.0f} s, hip {t_hip:.0f} s', file=sys.stderr, flush=True)
    model.model.fixed_noise(None, None)
    hip, cpu = torch.cat(keep['hip']), torch.cat(keep['cpu'])
    n = hip.numel()
    n10, n01 = int((hip & ~cpu).sum()), int((~hip & cpu).sum())
    delta = (n10 - n01) / n
    se = math.sqrt(max(n10 + n01 - (n10 - n01) ** 2 / n, 0.0)) / n
    same = sum(int((a != b).sum()) for pair in keep['cpu_on_hip'] + keep['hip_on_cpu'] for a, b in [pair])
    l2d = (torch.cat(keep['l2_hip']) - torch.cat(keep['l2_cpu'])).abs().max().item()
    return {'attack': 'apgd-ce', 'l2_bound': bound, 'n_iter': n_iter, 'images': n, 'eot': eot,
            'robust_acc_hip': hip.float().mean().item(), 'robust_acc_oracle': cpu.float().mean().item(),
            'delta': delta, 'ci95_halfwidth': 1.96 * se, 'ci95': [delta - 1.96 * se, delta + 1.96 * se],
            'within_0.1_percent': abs(delta) + 1.96 * se <= 1e-3,
            'discordant_pairs': n10 + n01, 'robust_on_hip_only': n10, 'robust_on_oracle_only': n01,
            'same_input_verdicts_differing': same, 'same_input_verdict_pairs': 2 * n,
            'max_abs_l2_difference': l2d, 'oracle_seconds': t_cpu, 'hip_seconds': t_hip, 'cpu_threads': threads,
            'what': f'{n} images x EoT {eot}, APGD-CE (the reference\'s class restated, {n_iter} iterations, L2 bound {bound}) through the '
                    'reduced NVAE + VGG defender behind load(args); start noise and per-call latent noise pinned and equal on both sides; '
                    'labels = the oracle\'s clean prediction; verdict = the attack\'s own success flag; paired 95 % interval on the '
                    'difference of the two robust accuracies'}


class _EngineFn(torch.autograd.Function):
    """EoT-mean logits of a bare Engine as a differentiable function of the image batch (the latent noise is whatever the caller put
    into eng.eps: pinned for the whole run)"""

    @staticmethod
    def forward(ctx, x, eng, eot):
        eng.x_in.copy_(x.detach())
        eng.forward()
        ctx.eng, ctx.eot, ctx.x = eng, eot, x.detach().clone()
        return eng.logits.view(x.shape[0], eot, -1).mean(dim=1).clone()

    @staticmethod
    def backward(ctx, dl):
        eng, eot = ctx.eng, ctx.eot
        eng.x_in.copy_(ctx.x)
        eng.forward()                                          # (another forward may have run on this engine since)
        eng.dlogits.view(dl.shape[0], eot, -1).copy_((dl / eot).unsqueeze(1).expand(-1, eot, -1))
        eng.backward()
        return eng.dx.clone(), None, None


class _EngineNet(torch.nn.Module):
    def __init__(self, eng, eot):
        super().__init__()
        self.eng, self.eot = eng, eot

    def forward(self, x):
        return _EngineFn.apply(x, self.eng, self.eot)


def fullsize_same_input_verdicts(device='cuda:0', n_images=8, eot=2, n_iter=3, bound=2.0, seed=0, threads=None):
    """VERDICT r03 "next round" #3, last clause: same-input verdicts on the FULL-SIZE model (the assumed NVAE configuration of the
    bench + VGG-11) for a small N.  The HIP engine produces adversarial examples with the reference's APGD-CE (pinned latent noise,
    pinned start noise); the CPU oracle then judges the clean and the adversarial images under the same noise, as does the engine:
    verdicts (argmax of the EoT-mean logits) and logits are compared on identical inputs."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import build_model
    from gen_adversarial_amd.attacks.l2_attacks import APGDAttack
    from gen_adversarial_amd.nvae_spec import ASSUMED_NVAE_CONFIG, ASSUMED_NVAE_RESOLUTION, build_spec
    from oracle import defender_oracle as D
    if threads is None:
        threads = min(16, len(os.sched_getaffinity(0)))
    torch.set_num_threads(threads)
    rows = n_images * eot
    eng, (sd, vsd, vspec, alphas) = build_model(device, rows, eot, seed=0)
    spec = build_spec(ASSUMED_NVAE_CONFIG, ASSUMED_NVAE_RESOLUTION)
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(n_images, 3, 64, 64, generator=g)
    eps = [torch.randn(rows, spec.num_latent, gs.res, gs.res, generator=g) for gs in spec.groups]
    init = torch.randn(n_images, 3, 64, 64, generator=g)
    for dst, src in zip(eng.eps, eps):
        dst.copy_(src.to(device))
    net = _EngineNet(eng, eot)

    def oracle(imgs):
        with torch.no_grad():
            lg, _ = D.nvae_defender(sd, spec, vsd, vspec, imgs.repeat_interleave(eot, dim=0), alphas, eps, torch.ones(rows, 3, 64, 64), 0.0)
        return lg.view(n_images, eot, -1).mean(dim=1)
    t = time.time()
    lo_clean = oracle(x)
    labels = lo_clean.argmax(dim=1)
    with torch.no_grad():
        lh_clean = net(x.to(device)).cpu()
    t_h = time.time()
    ok, b, adv = APGDAttack(n_iter=n_iter, rho=0.75, max_bound=bound, ce_loss=True)(x.to(device), labels.to(device), net, init_noise=init.to(device))
    torch.cuda.synchronize()
    t_h = time.time() - t_h
    with torch.no_grad():
        lh_adv = net(adv).cpu()
    lo_adv = oracle(adv.cpu())
    t_all = time.time() - t
    vh = torch.cat([lh_clean.argmax(dim=1) == labels, lh_adv.argmax(dim=1) == labels])
    vo = torch.cat([lo_clean.argmax(dim=1) == labels, lo_adv.argmax(dim=1) == labels])
    top2 = torch.cat([lo_clean, lo_adv]).topk(2, dim=1).values
    return {'model': 'full size (assumed NVAE configuration C=32, 3x8 groups, 20 latents + VGG-11), random weights',
            'images': n_images, 'eot': eot, 'attack': f'apgd-ce, {n_iter} iterations, L2 bound {bound}, run on the HIP engine',
            'verdict_pairs': int(vh.numel()), 'verdicts_differing': int((vh != vo).sum()),
            'robust_acc_hip_judged': float(vh[n_images:].float().mean()), 'robust_acc_oracle_judged': float(vo[n_images:].float().mean()),
            'max_abs_logit_err_clean': float((lh_clean - lo_clean).abs().max()), 'max_abs_logit_err_adversarial': float((lh_adv - lo_adv).abs().max()),
            'smallest_oracle_decision_margin': float((top2[:, 0] - top2[:, 1]).min()),
            'mean_l2_of_adversarial': float(torch.as_tensor(b).float().mean()), 'hip_attack_seconds': t_h, 'total_seconds': t_all,
            'cpu_threads': threads}
