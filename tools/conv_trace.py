"""GPU tool (debug build only): per-workgroup phase timeline of one conv_bf3 launch.

    make -C gen_adversarial_amd/csrc trace
    GA_OPS_LIB=gen_adversarial_amd/libga_ops_trace.so python tools/conv_trace.py N H Cin Cout K tile splits aff act

Phases (shader clocks, thread 0 of every workgroup of split 0): setup | first tile staged | K loop | epilogue.
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from gen_adversarial_amd import _lib as L

N, H, Cin, Cout, K = (int(v) for v in sys.argv[1:6])
tile = int(sys.argv[6]) if len(sys.argv) > 6 else 1
splits = int(sys.argv[7]) if len(sys.argv) > 7 else 1
aff = int(sys.argv[8]) if len(sys.argv) > 8 else 0
act = int(sys.argv[9]) if len(sys.argv) > 9 else 0
odd_ld = int(sys.argv[10]) if len(sys.argv) > 10 else 0      # 1: output pitch Cout+1 -> scalar (direct-from-accumulator) epilogue
x = torch.randn(N, H, H, Cin, device='cuda')
w = torch.randn(Cout, K * K * Cin, device='cuda') * 0.05
y = torch.empty(N, H, H, Cout + (1 if len(sys.argv) > 10 and int(sys.argv[10]) else 0), device='cuda')
ws = torch.empty(max(1, splits * N * H * H * Cout), device='cuda')
sc = torch.rand(Cin, device='cuda') + 0.5
sh = torch.randn(Cin, device='cuda') * 0.1
d = L.ConvDesc()
d.x, d.ldx, d.C1, d.w, d.y, d.ldy, d.Cout = x.data_ptr(), Cin, Cin, w.data_ptr(), y.data_ptr(), y.shape[3], Cout
d.N, d.Hi, d.Wi, d.Ho, d.Wo, d.KH, d.KW, d.sn, d.sd, d.pad, d.tile, d.pro_act = N, H, H, H, H, K, K, 1, 1, K // 2, tile, act
d.splits = splits
if splits > 1:
    d.ws, d.ws_floats = ws.data_ptr(), ws.numel()
if aff:
    d.pro_scale, d.pro_shift = sc.data_ptr(), sh.data_ptr()
hi = w.to(torch.bfloat16)
lo = (w - hi.float()).to(torch.bfloat16)
d.w_hi, d.w_lo = hi.data_ptr(), lo.data_ptr()
if tile == 8:
    from gen_adversarial_amd.engine_core import WeightStore
    _store = WeightStore('cuda')
    _frag = _store.frag3(w)
    d.w_frag = _frag.data_ptr()
for _ in range(3):
    L.run(d)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = 20
e0.record()
for _ in range(reps):
    L.run(d)
e1.record()
e1.synchronize()
us = e0.elapsed_time(e1) / reps * 1e3
L.run(d)
torch.cuda.synchronize()
bm, bn = {1: (128, 128), 2: (128, 64), 3: (64, 64), 4: (128, 32), 5: (128, 128), 6: (128, 64), 8: (128, 128)}[tile]
M = N * H * H
nwg = min(8192, -(-M // bm) * -(-Cout // bn))
buf = np.zeros(8 * 8192, dtype=np.uint64)
lib = C.CDLL(os.environ['GA_OPS_LIB'])
rc = (lib.ga_debug_trace_read_halo if tile >= 5 else lib.ga_debug_trace_read)(buf.ctypes.data_as(C.c_void_p), C.c_int(buf.size))
assert rc == 0, rc
t = buf.reshape(8192, 8)[:nwg].astype(np.int64)
t0 = t[:, 0].min()
ph = np.diff(t[:, :5], axis=1)
fl = 2 * M * K * K * Cin * Cout
print(f'conv N{N} H{H} {Cin}->{Cout} k{K} tile{tile} splits{splits} aff{aff} act{act}: {us:.1f} us/launch back-to-back '
      f'({fl / us / 1e6:.1f} TF/s), {nwg} workgroups x {splits} splits, K steps per workgroup {K * K * -(-Cin // 32) // splits}')
print(f'grid span (first start -> last end): {(t[:, 4].max() - t0)} clk;  starts spread over {(t[:, 0].max() - t0)} clk')
for i, nm in enumerate(('setup', 'first tile', 'K loop', 'epilogue')):
    print(f'  {nm:10s} mean {ph[:, i].mean():9.0f}  min {ph[:, i].min():8d}  max {ph[:, i].max():8d} clk')
if tile in (5, 6, 7):
    nst = K * K * -(-Cin // 32) // splits
    inner = t[:, 5:8].astype(np.float64)
    print(f'  K-loop anatomy, wave 0, clocks per step ({nst} steps): MFMA part {inner[:, 0].mean() / nst:7.0f}  '
          f'patch staging at chunk ends {inner[:, 1].mean() / nst:7.0f}  step barrier {inner[:, 2].mean() / nst:7.0f}   '
          f'(MFMA floor alone on the SIMD: {24 * 32 if tile == 5 else 0} clk per step at 128x128)')
# each XCD has its own counter; workgroup i is dispatched to XCD i % 8 -> starts relative to the XCD's first workgroup
rel = np.zeros(nwg, dtype=np.int64)
for xcd in range(8):
    sel = np.arange(nwg) % 8 == xcd
    if sel.any():
        rel[sel] = t[sel, 0] - t[sel, 0].min()
print('start offset within its XCD (clk) percentiles 0/25/50/75/100:', np.percentile(rel, [0, 25, 50, 75, 100]).astype(int).tolist(),
      ' workgroup lifetime mean', int((t[:, 4] - t[:, 0]).mean()))
order = np.argsort(t[:, 0])
print('start offsets (clk) of every 32nd workgroup in start order:', (t[order[::max(1, nwg // 16)], 0] - t0).tolist())
