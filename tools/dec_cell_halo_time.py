"""GPU tool: time of the ga_dec_cell_halo launches at the bench shapes (512 rows: 64 x 64 x 32 hidden 96, 32 x 32 x 64 hidden 192)."""
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


for H, C, Hd in ((64, 32, 96), (32, 64, 192)):
    g = torch.Generator(device=DEV).manual_seed(1)
    r = lambda *s, sc=1.0: torch.randn(*s, device=DEV, generator=g) * sc                          # noqa: E731
    x, dout, add = r(rows, H, H, C), r(rows, H, H, C), r(rows, H, H, C)
    w1, w2 = r(Hd, C, sc=C ** -0.5), r(C, Hd, sc=Hd ** -0.5)
    wd, b1, bd, b2 = r(25, Hd, sc=0.2), r(Hd, sc=0.3), r(Hd, sc=0.3), r(C, sc=0.3)
    ps, pb = r(rows, C).abs() * 0.1 + 0.05, r(rows, C, sc=0.01)
    wdb = wd.flip(0).contiguous()
    keep = [split(w1), split(w2), split(w2.t().contiguous()), split(w1.t().contiguous())]
    y, dx = torch.empty(rows, H, H, C, device=DEV), torch.empty(rows, H, H, C, device=DEV)
    ds = []
    for bwd in (0, 1):
        d = L.DecCellHaloDesc()
        d.x, d.b1, d.wd, d.wd_bwd, d.bd, d.b2 = x.data_ptr(), b1.data_ptr(), wd.data_ptr(), wdb.data_ptr(), bd.data_ptr(), b2.data_ptr()
        d.w1_hi, d.w1_lo = (t.data_ptr() for t in keep[0])
        d.w2_hi, d.w2_lo = (t.data_ptr() for t in keep[2 if bwd else 1])
        d.w1t_hi, d.w1t_lo = (t.data_ptr() for t in keep[3])
        d.dout, d.pro_scale, d.pro_shift, d.addend = dout.data_ptr(), ps.data_ptr(), pb.data_ptr(), add.data_ptr()
        d.y = (dx if bwd else y).data_ptr()
        d.N, d.H, d.W, d.Cin, d.Cout, d.Hd, d.backward, d.up = rows, H, H, C, C, Hd, bwd, 0
        L.run(d)
        ds.append(d)
    torch.cuda.synchronize()

    def timed(d, reps=10):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            L.run(d)
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3
    t = np.array([[timed(d) for d in ds] for _ in range(5)])
    mb = rows * H * H * C * 4 / 1e6
    print(f'{H}x{H}x{C} hidden {Hd} rows {rows}: fwd {np.median(t[:, 0]):7.1f} us ({2 * mb / np.median(t[:, 0]):.2f} TB/s of x + t3), '
          f'bwd {np.median(t[:, 1]):7.1f} us ({4 * mb / np.median(t[:, 1]):.2f} TB/s of x + dout + addend + dx)', flush=True)
