"""GPU tool: the full-size Style-Transformer + StyleGAN2-512 defender attack step (BASELINE.json configs[4]):
128x128 image -> blur -> resize 256 / crop -> IR-SE50 trunk at 192x256 -> 3 transformer decoder layers over 16 queries -> 16 x 512
latents mixed with mapped N(0, 0.8) noise -> StyleGAN2 512x512 synthesis -> pool / -1 band / resize to 128 -> ResNeXt-50 32x4d logits,
and the input gradient.  Random weights of the reference architecture.  Prints the step time, conv TFLOP/s and the time by op kind.
    python tools/trans_defense_bench.py [rows=32] [eot=32] [tune.json]      (tune.json: autotune the conv shapes missing from the table)"""
import collections
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import yaml
from gen_adversarial_amd import _lib as L
from bench import build_trans_defender, conv_algorithmic_flops

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 32
eot = int(sys.argv[2]) if len(sys.argv) > 2 else 32
tune_out = sys.argv[3] if len(sys.argv) > 3 else None
eng, y = build_trans_defender('cuda:0', rows, eot, 'bf16x3')
print(f'built: {len(eng.fwd)} + {len(eng.bwd)} ops, {eng.bytes / 1e9:.1f} GB engine buffers, {eng.store.bytes / 1e9:.2f} GB weights', flush=True)
eng.x_in.uniform_()
eng.eps[0].normal_().mul_(0.8)
eng.forward(); eng.dlogits.normal_(); eng.backward(); torch.cuda.synchronize()
assert torch.isfinite(eng.logits).all() and torch.isfinite(eng.dx).all()
print('logits', eng.logits.flatten()[:4].tolist(), '|dx|', eng.dx.abs().max().item(), flush=True)
if tune_out:
    cache = eng.autotune(reps=3, save=tune_out, verbose=False)
    print('tuned entries', len(cache), flush=True)
s = eng.stream()
f_ms, fc_ms, fn = eng.fwd.time(s, iters=3, per_conv=True)
b_ms, bc_ms, bn = eng.bwd.time(s, iters=3, per_conv=True)
fl = conv_algorithmic_flops(eng.fwd) + conv_algorithmic_flops(eng.bwd)
print(f'step: fwd {f_ms:.1f} ms + bwd {b_ms:.1f} ms = {f_ms + b_ms:.1f} ms for {rows} rows -> {rows / (f_ms + b_ms) * 1e3:.0f} rows/s; '
      f'convs {fl / 1e9 / rows:.0f} GFLOP/row at {fl / ((fc_ms + bc_ms) / 1e3) / 1e12:.0f} TFLOP/s ({int(fn + bn)} launches)')
for plan, tag in ((eng.fwd, 'fwd'), (eng.bwd, 'bwd')):
    ms = plan.profile(s)
    ms = plan.profile(s)
    groups = collections.OrderedDict()
    for d, nm, t in zip(plan.descs, plan.names, ms):
        sec = 'trunk' if nm.startswith('e4e.') else 'transformer' if nm.startswith('trans.transformerlayer') else \
              'classifier' if nm.startswith('resnet') else 'generator' if (nm.startswith('conv') or nm.startswith('to_rgb') or nm.startswith('sg.')) else 'glue'
        g = groups.setdefault((sec, type(d).__name__), [0, 0.0])
        g[0] += 1; g[1] += t
    tot = sum(ms)
    print(f'==== {tag}: {tot:.1f} ms')
    for k, g in sorted(groups.items(), key=lambda kv: -kv[1][1])[:16]:
        print(f'   {k[0]:12s} {k[1]:18s} n {g[0]:4d} {g[1]:8.2f} ms {100 * g[1] / tot:5.1f}%')
