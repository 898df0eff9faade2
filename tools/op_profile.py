"""GPU tool: per-op time of the non-conv ops in the full-size plans, grouped by kind and shape, with achieved GB/s."""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gen_adversarial_amd import _lib as L
from bench import build_model

R = int(sys.argv[1]) if len(sys.argv) > 1 else 256
eng, _ = build_model('cuda:0', R, 32)
eng.x_in.uniform_()
for e in eng.eps: e.normal_()
eng.forward(); eng.dlogits.normal_(); eng.backward(); torch.cuda.synchronize()
s = eng.stream()

def nbytes(d):
    if isinstance(d, L.DwDesc):
        hi = d.N * d.H * d.W * d.C * 4
        i = hi // 4 if d.up2 else hi
        o = hi // 4 if d.pool2 else hi
        extra = o if d.dact_x else 0
        return i + o + extra
    if isinstance(d, L.ReduceDesc):
        return d.N * d.P * d.C * 4 * (2 if d.b else 1)
    if isinstance(d, L.SeApplyDesc):
        n = d.N * d.H * d.W * d.C * 4
        return 3 * n if d.skip_mode == 0 else 2 * n + n // 4
    return 0

for plan, tag in ((eng.fwd, 'fwd'), (eng.bwd, 'bwd')):
    ms = plan.profile(s); ms = plan.profile(s)
    groups = collections.OrderedDict()
    for d, nm, t in zip(plan.descs, plan.names, ms):
        if isinstance(d, L.ConvDesc):
            continue
        if isinstance(d, L.DwDesc): key = ('dw', d.H, d.C, d.up2, d.pool2)
        elif isinstance(d, L.ReduceDesc): key = ('reduce', d.P, d.C, int(bool(d.b)), 0)
        elif isinstance(d, L.SeApplyDesc): key = ('se_apply', d.H, d.C, d.skip_mode, 0)
        elif isinstance(d, L.SeExciteDesc): key = ('se_excite', d.C, d.Hd, d.backward, 0)
        else: key = (type(d).__name__, 0, 0, 0, 0)
        g = groups.setdefault(key, [0, 0.0, 0])
        g[0] += 1; g[1] += t; g[2] += nbytes(d)
    print(f'==== {tag}')
    for key, g in sorted(groups.items(), key=lambda kv: -kv[1][1]):
        gbs = g[2] / g[1] / 1e6 if g[2] else 0
        print(f'{str(key):45s} n {g[0]:3d}  {g[1]:8.3f} ms  {gbs:8.0f} GB/s')
