"""GPU tool: A/B of the ga_dec_cell variants (4 waves per workgroup vs 8) on the bench shapes, interleaved rounds in one process
(cdna_hip_programming.md §5.4 rule 24), with a bitwise comparison of their results.

    python tools/dec_cell_ab.py [rows]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from gen_adversarial_amd import _lib as L

DEV = 'cuda:0'
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 512


def split(w):
    hi = w.to(torch.bfloat16)
    return hi.contiguous(), (w - hi.float()).to(torch.bfloat16).contiguous()


def setup(N, H, C, Hd):
    g = torch.Generator(device=DEV).manual_seed(1)
    r = lambda *s, sc=1.0: torch.randn(*s, device=DEV, generator=g) * sc                          # noqa: E731
    x, dout = r(N, H, H, C), r(N, H, H, C)
    w1, w2 = r(Hd, C, sc=C ** -0.5), r(C, Hd, sc=Hd ** -0.5)
    wd = r(25, Hd, sc=0.2)
    b1, bd, b2 = r(Hd, sc=0.3), r(Hd, sc=0.3), r(C, sc=0.3)
    ps, pb = r(N, C).abs() * 0.1 + 0.05, r(N, C, sc=0.01)
    keep = [x, dout, w1, w2, wd, b1, bd, b2, ps, pb, wd.flip(0).contiguous()]
    w1h, w1l = split(w1)
    w2h, w2l = split(w2)
    w2th, w2tl = split(w2.t().contiguous())
    keep += [w1h, w1l, w2h, w2l, w2th, w2tl]

    def desc(backward, variant, y):
        d = L.DecCellDesc()
        d.x, d.w1_hi, d.w1_lo, d.b1 = x.data_ptr(), w1h.data_ptr(), w1l.data_ptr(), b1.data_ptr()
        d.wd, d.wd_bwd, d.bd = wd.data_ptr(), keep[10].data_ptr(), bd.data_ptr()
        if backward:
            d.w2_hi, d.w2_lo = w2th.data_ptr(), w2tl.data_ptr()
            d.dout, d.pro_scale, d.pro_shift = dout.data_ptr(), ps.data_ptr(), pb.data_ptr()
        else:
            d.w2_hi, d.w2_lo, d.b2 = w2h.data_ptr(), w2l.data_ptr(), b2.data_ptr()
        d.y = y.data_ptr()
        d.N, d.H, d.W, d.C, d.Hd, d.backward, d.variant = N, H, H, C, Hd, backward, variant
        return d
    return desc, keep


def timed(d, reps=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        L.run(d)
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for (H, C, Hd) in ((16, 128, 768),):
    desc, keep = setup(rows, H, C, Hd)
    for backward in (0, 1):
        ys = [torch.zeros(rows, H, H, Hd if backward else C, device=DEV) for _ in range(2)]
        ds = [desc(backward, v, ys[v]) for v in (0, 1)]
        for d in ds:
            L.run(d)
        torch.cuda.synchronize()
        same = torch.equal(ys[0], ys[1])
        diff = (ys[0] - ys[1]).abs().max().item()
        t = np.array([[timed(d) for d in ds] for _ in range(5)])
        print(f'{H}x{H}x{C} hidden {Hd} rows {rows} {"bwd" if backward else "fwd"}: 4 waves {np.median(t[:, 0]):7.1f} us (min {t[:, 0].min():.1f}), '
              f'8 waves {np.median(t[:, 1]):7.1f} us (min {t[:, 1].min():.1f}); bitwise equal {same} (max diff {diff:.1e})', flush=True)
