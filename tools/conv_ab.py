"""GPU tool: A/B of ga_conv2d tile codes on one 3x3 shape (interleaved rounds in one process), with a bitwise comparison.

    python tools/conv_ab.py N H Cin Cout [act] [tiles...]        e.g.  512 16 128 128 1 5 8
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from gen_adversarial_amd import _lib as L
from gen_adversarial_amd.engine_core import WeightStore

N, H, Cin, Cout = (int(v) for v in sys.argv[1:5])
act = int(sys.argv[5]) if len(sys.argv) > 5 else 1
tiles = [int(v) for v in sys.argv[6:]] or [5, 8]
dev = 'cuda:0'
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(N, H, H, Cin, device=dev, generator=g)
if os.environ.get('GA_AB_ZERO'):
    x.zero_()          # clock check: zero operands draw less power (cdna_hip_programming.md rule 25)
w = torch.randn(Cout, 9 * Cin, device=dev, generator=g) * (9 * Cin) ** -0.5
if os.environ.get('GA_AB_ZERO'):
    w.zero_()
b = torch.randn(Cout, device=dev, generator=g) * 0.1
store = WeightStore(dev)
hi, lo = store.split(w)
frag = store.frag3(w)
frag16 = store.frag3(w, m16=True)
frag_thin = store.frag_thin(w) if Cin in (32, 64) else None
dact = torch.randn(N, H, H, Cout, device=dev, generator=g)
add = torch.randn(N, H, H, Cout, device=dev, generator=g)
ys, ds = [], []
for t in tiles:
    y = torch.zeros(N, H, H, Cout, device=dev)
    d = L.ConvDesc()
    d.x, d.ldx, d.C1, d.w, d.bias, d.y, d.ldy, d.Cout = x.data_ptr(), Cin, Cin, w.data_ptr(), b.data_ptr(), y.data_ptr(), Cout, Cout
    d.N, d.Hi, d.Wi, d.Ho, d.Wo, d.KH, d.KW, d.sn, d.sd, d.pad, d.tile, d.pro_act = N, H, H, H, H, 3, 3, 1, 1, 1, t, act
    d.w_hi, d.w_lo, d.w_frag = hi.data_ptr(), lo.data_ptr(), (frag16 if t in (9, 10) else frag_thin if t == 11 else frag).data_ptr()
    if os.environ.get('GA_AB_BWD'):     # the epilogue of a backward conv: act' of a saved activation and an identity-skip addend
        d.dact_x, d.lddact, d.dact_act, d.addend, d.ldadd = dact.data_ptr(), Cout, 1, add.data_ptr(), Cout
    L.run(d)
    ys.append(y)
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
fl = 2 * N * H * H * 9 * Cin * Cout
ref = None if os.environ.get('GA_AB_BWD') else torch.nn.functional.conv2d((x * torch.sigmoid(x) if act == 1 else x).permute(0, 3, 1, 2)[:8], w.view(Cout, 3, 3, Cin).permute(0, 3, 1, 2), b, padding=1)
for i, tl in enumerate(tiles):
    err = float('nan') if ref is None else (ys[i][:8].permute(0, 3, 1, 2) - ref).abs().max().item()
    print(f'N{N} {H}x{H} {Cin}->{Cout} act{act} tile {tl}: {np.median(t[:, i]):7.1f} us (min {t[:, i].min():.1f}) = {fl / np.median(t[:, i]) / 1e6:6.1f} TF/s; '
          f'vs torch {err:.1e}; bitwise equal to tile {tiles[0]}: {torch.equal(ys[i], ys[0])}; max diff to tile {tiles[0]}: {(ys[i] - ys[0]).abs().max().item():.1e}', flush=True)
