"""GPU tool: per-launch time of every conv in the full-size plans, grouped by shape."""
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
for plan, tag in ((eng.fwd, 'fwd'), (eng.bwd, 'bwd')):
    ms = plan.profile(s); ms = plan.profile(s)
    groups = collections.OrderedDict()
    other = collections.Counter()
    for d, nm, t in zip(plan.descs, plan.names, ms):
        if isinstance(d, L.ConvDesc):
            M = d.N * d.Ho * d.Wo
            pix = M if d.sd == 1 else d.N * d.Hi * d.Wi
            fl = 2 * pix * d.KH * d.KW * (d.C1 + d.C2) * d.Cout
            key = (M, d.Cout, d.C1 + d.C2, d.KH, d.sn, d.sd, d.pro_act, int(bool(d.dact_x)))
            g = groups.setdefault(key, [0, 0.0, 0.0, nm])
            g[0] += 1; g[1] += t; g[2] += fl
        else:
            other[type(d).__name__] += t
    tot = sum(g[1] for g in groups.values())
    print(f'==== {tag}: conv total {tot:.2f} ms, others ' + ', '.join(f'{k} {v:.2f}' for k, v in other.items()))
    print(f'{"M":>8} {"Cout":>6} {"Cin":>6} K sn sd act dact  {"n":>3} {"ms":>8} {"TF/s":>7}  example')
    for key, g in sorted(groups.items(), key=lambda kv: -kv[1][1]):
        M, co, ci, K, sn, sd, pa, da = key
        print(f'{M:8d} {co:6d} {ci:6d} {K} {sn:2d} {sd:2d} {pa:3d} {da:4d}  {g[0]:3d} {g[1]:8.3f} {g[2] / g[1] / 1e9:7.1f}  {g[3]}')
