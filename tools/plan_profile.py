"""GPU tool: per-op device time of the full-size NVAE + VGG plans (512-row chunk by default), grouped by layer kind and
shape — where a chunk's forward + backward time goes.  python tools/plan_profile.py [rows]"""
import collections
import os
import re
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gen_adversarial_amd import _lib as L
from bench import build_model, conv_algorithmic_flops

R = int(sys.argv[1]) if len(sys.argv) > 1 else 512
eng, _ = build_model('cuda:0', R, 32)
eng.x_in.uniform_()
for e in eng.eps:
    e.normal_()
eng.forward(); eng.dlogits.normal_(); eng.backward(); torch.cuda.synchronize()
s = eng.stream()


def kind(name):
    n = re.sub(r'\d+', '#', name)
    return n


tot = {}
for plan, tag in ((eng.fwd, 'fwd'), (eng.bwd, 'bwd')):
    ms = plan.profile(s)
    ms = plan.profile(s)
    groups = collections.OrderedDict()
    for d, nm, t in zip(plan.descs, plan.names, ms):
        if isinstance(d, L.ConvDesc):
            key = (tag, 'conv', kind(nm.split('.')[-1] if not nm.startswith('vgg') else nm), d.N * d.Ho * d.Wo, d.C1 + d.C2, d.Cout, d.KH)
            fl = 2.0 * (d.N * d.Ho * d.Wo if d.sd == 1 else d.N * d.Hi * d.Wi) * d.KH * d.KW * (d.C1 + d.C2) * d.Cout
            by = 4.0 * (d.N * d.Hi * d.Wi * (d.C1 + d.C2) + d.N * d.Ho * d.Wo * d.Cout * (1 + bool(d.dact_x) + bool(d.addend) + bool(d.addend2)))
        else:
            key = (tag, type(d).__name__, kind(nm.split('.')[-1]), getattr(d, 'N', 0) * getattr(d, 'H', 1) * getattr(d, 'W', 1), getattr(d, 'C', 0), 0, 0)
            fl, by = 0.0, 0.0
        g = groups.setdefault(key, [0, 0.0, 0.0, 0.0])
        g[0] += 1; g[1] += t; g[2] += fl; g[3] += by
    total = sum(ms)
    tot[tag] = total
    print(f'==== {tag}: {total:.2f} ms, {len(ms)} ops')
    for key, g in sorted(groups.items(), key=lambda kv: -kv[1][1])[:45]:
        tf = g[2] / g[1] / 1e9 if g[2] else 0
        tb = g[3] / g[1] / 1e9 if g[3] else 0
        print(f'{str(key[1:]):70s} n {g[0]:3d} {g[1]:8.3f} ms {100 * g[1] / total:5.1f}%  {tf:7.1f} TF/s  {tb:6.2f} TB/s(min)')
print('total', tot, 'rows/s single plan', R / (sum(tot.values()) / 1e3))
