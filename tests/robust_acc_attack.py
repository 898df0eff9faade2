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
                                 tmpdir=None, progress=False):
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
    ck = nvae_checkpoint(CFG, RES, seed=seed + 5)
    vsd = init_vgg_state_dict(100, 16, seed=seed + 6)
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
