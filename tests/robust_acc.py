"""
Robust-accuracy delta of the HIP path against the CPU oracle (BASELINE.json: "robust-accuracy within +-0.1 % of reference";
VERDICT r02 weak #8: the bar needs a sample size that can resolve it).  TEST INFRASTRUCTURE: imports the oracle as the checker;
used by tests/test_engine_gpu.py and by bench.py's cpu_baseline leg, never by the product path.

Protocol: N images x EoT `eot`, PGD-Linf (`steps` iterations, eps 8/255, step 2/255) through purifier + classifier, the SAME
latent noise in both implementations at every step (drawn on the CPU from one seeded generator), labels = the oracle's clean
prediction.  Each implementation follows its OWN trajectory (sign of its own gradient); the verdict of an image is whether the
final adversarial example is still classified as its label.  Reported: both robust accuracies, their difference, the number of
images whose verdicts differ, and the number whose CLEAN predictions already differ.

Model: a reduced NVAE (C = 4, 2 scales x 2 groups, 4 latents) + VGG-11 at 1/16 width with seeded random weights — the oracle
needs ~10 ms per row and pass here, the full-size model ~1.5 s.
"""
import os
import time

import torch

CFG = {'initial_channels': 4, 'num_pre-post_process_blocks': 1, 'num_pre-post_process_cells': 2, 'num_scales': 2,
       'num_groups_per_scale': 2, 'is_adaptive': False, 'min_groups_per_scale': 1, 'num_cells_per_group': 1,
       'num_latent_per_group': 4, 'num_logistic_mixtures': 10, 'num_nf_cells': None}
RES = (3, 64, 64)


def robust_accuracy_delta(device='cuda:0', n_images=512, eot=4, steps=6, eps=8.0 / 255.0, step=2.0 / 255.0, seed=0,
                          chunk_images=128, precision='bf16x3', threads=None, logit_scale=1.0):
    from gen_adversarial_amd.engine import Engine
    from gen_adversarial_amd.nvae_spec import build_spec, init_nvae_state_dict
    from gen_adversarial_amd.vgg_spec import build_vgg_spec, init_vgg_state_dict
    from oracle import defender_oracle as D
    spec = build_spec(CFG, RES)
    sd = init_nvae_state_dict(CFG, RES, seed + 5)
    vspec = build_vgg_spec(100, 16)
    vsd = init_vgg_state_dict(100, 16, seed + 6)
    ng = len(spec.groups)
    alphas = [0.7 * i / (ng - 1) for i in range(ng)]
    if threads is None:
        try:
            threads = min(16, len(os.sched_getaffinity(0)))
        except AttributeError:
            threads = min(16, os.cpu_count() or 1)
    torch.set_num_threads(threads)
    chunk_images = min(chunk_images, n_images)
    assert n_images % chunk_images == 0
    rows = chunk_images * eot
    eng = Engine(sd, CFG, RES, vsd, vspec, rows=rows, rep=eot, alphas=alphas, temperature=0.6, noise_eps=0.0, device=device,
                 precision=precision, share_encoder=False)
    g = torch.Generator().manual_seed(seed)
    t_cpu = t_gpu = 0.0
    keep = {'hip': [], 'cpu': [], 'clean_agree': [], 'cpu_on_hip': [], 'hip_on_cpu': []}

    dummy_noise = torch.ones(rows, *RES)          # scaled by eps = 0 (abstract_models.py:132-138 divides by its norm: must not be 0)

    def draw():
        return [torch.randn(rows, spec.num_latent, gs.res, gs.res, generator=g) for gs in spec.groups]

    def cpu_logits(x, e, grad):
        xr = x.clone().requires_grad_(grad)
        with torch.set_grad_enabled(grad):
            lg, _ = D.nvae_defender(sd, spec, vsd, vspec, xr.repeat_interleave(eot, dim=0), alphas, e, dummy_noise, 0.0)
            mean = lg.view(-1, eot, lg.shape[-1]).mean(dim=1)
        return xr, mean

    def hip_logits(x, e):
        eng.x_in.copy_(x.to(device))
        for dst, src in zip(eng.eps, e):
            dst.copy_(src.to(device))
        eng.forward()
        return eng.logits.view(-1, eot, eng.logits.shape[-1]).mean(dim=1)

    for c in range(n_images // chunk_images):
        x0 = torch.rand(chunk_images, *RES, generator=g)
        e0 = draw()
        t = time.time()
        _, m0 = cpu_logits(x0, e0, False)
        t_cpu += time.time() - t
        labels = m0.argmax(dim=1)
        h0 = hip_logits(x0, e0).argmax(dim=1).cpu()
        keep['clean_agree'].append(h0 == labels)
        xc, xh = x0.clone(), x0.clone().to(device)
        lab_d = labels.to(device)
        for s in range(steps):
            e = draw()
            t = time.time()
            xr, mean = cpu_logits(xc, e, True)
            (gc,) = torch.autograd.grad(torch.nn.functional.cross_entropy(mean, labels, reduction='sum'), [xr])
            xc = torch.min(torch.max(xc + step * gc.sign(), x0 - eps), x0 + eps).clamp_(0.0, 1.0)
            t_cpu += time.time() - t
            t = time.time()
            mh = hip_logits(xh, e)
            p = torch.softmax(mh, dim=1)
            p[torch.arange(chunk_images, device=device), lab_d] -= 1.0
            eng.dlogits.view(-1, eot, p.shape[-1]).copy_((p / eot).unsqueeze(1).expand(-1, eot, -1))
            eng.backward()
            x0d = x0.to(device)
            xh = torch.min(torch.max(xh + step * eng.dx.sign(), x0d - eps), x0d + eps).clamp_(0.0, 1.0)
            torch.cuda.synchronize()
            t_gpu += time.time() - t
        ef = draw()
        t = time.time()
        _, mc = cpu_logits(xc, ef, False)
        t_cpu += time.time() - t
        keep['cpu'].append(mc.argmax(dim=1) == labels)
        keep['hip'].append(hip_logits(xh, ef).argmax(dim=1).cpu() == labels)
        # the two implementations as JUDGES of the same adversarial examples (same noise): this isolates the parity of purifier +
        # classifier at the adversarial points from the divergence of two sign-gradient trajectories
        t = time.time()
        _, mx = cpu_logits(xh.cpu(), ef, False)
        t_cpu += time.time() - t
        keep['cpu_on_hip'].append(mx.argmax(dim=1) == labels)
        keep['hip_on_cpu'].append(hip_logits(xc, ef).argmax(dim=1).cpu() == labels)
    hip, cpu, clean, cpu_on_hip, hip_on_cpu = (torch.cat(keep[k]) for k in ('hip', 'cpu', 'clean_agree', 'cpu_on_hip', 'hip_on_cpu'))
    acc_h, acc_c = hip.float().mean().item(), cpu.float().mean().item()
    return {'images': n_images, 'eot': eot, 'pgd_steps': steps, 'eps': eps, 'step': step,
            'robust_acc_hip': acc_h, 'robust_acc_oracle': acc_c, 'delta': abs(acc_h - acc_c),
            'differing_verdicts': int((hip != cpu).sum()), 'clean_predictions_differing': int((~clean).sum()),
            'oracle_acc_on_hip_examples': cpu_on_hip.float().mean().item(), 'hip_acc_on_oracle_examples': hip_on_cpu.float().mean().item(),
            'same_input_verdicts_differing': int((cpu_on_hip != hip).sum()) + int((hip_on_cpu != cpu).sum()),
            'oracle_seconds': t_cpu, 'hip_seconds': t_gpu, 'cpu_threads': threads,
            'what': f'{n_images} images x EoT {eot}, PGD-Linf {steps} steps (eps 8/255, step 2/255) through the reduced NVAE + VGG defender, '
                    'identical latent noise per step in both implementations, labels = the oracle\'s clean prediction, each implementation '
                    'on its own trajectory; verdict = final adversarial example still classified as its label.  same_input_verdicts_differing: '
                    'each implementation also judges the OTHER one\'s adversarial examples (2 x N verdict pairs on identical inputs and noise)'}
