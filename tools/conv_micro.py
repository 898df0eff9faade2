"""GPU micro-benchmark: one conv shape, repeated; for rocprofv3 --pmc runs and A/B timing."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gen_adversarial_amd import _lib as L

N, H, Cin, Cout, K = (int(v) for v in sys.argv[1:6])
tile = int(sys.argv[6]) if len(sys.argv) > 6 else 1
pro = int(sys.argv[7]) if len(sys.argv) > 7 else 0
reps = int(sys.argv[8]) if len(sys.argv) > 8 else 20
bf3 = int(sys.argv[9]) if len(sys.argv) > 9 else 0
x = torch.randn(N, H, H, Cin, device='cuda')
w = torch.randn(Cout, K * K * Cin, device='cuda') * 0.05
y = torch.empty(N, H, H, Cout, device='cuda')
d = L.ConvDesc()
d.x, d.ldx, d.C1, d.w, d.y, d.ldy, d.Cout = x.data_ptr(), Cin, Cin, w.data_ptr(), y.data_ptr(), Cout, Cout
d.N, d.Hi, d.Wi, d.Ho, d.Wo, d.KH, d.KW, d.sn, d.sd, d.pad, d.tile, d.pro_act = N, H, H, H, H, K, K, 1, 1, K // 2, tile, pro
if bf3:
    hi = w.to(torch.bfloat16); lo = (w - hi.float()).to(torch.bfloat16)
    d.w_hi, d.w_lo = hi.data_ptr(), lo.data_ptr()
for _ in range(3):
    L.run(d)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    L.run(d)
e1.record(); e1.synchronize()
ms = e0.elapsed_time(e1) / reps
fl = 2 * N * H * H * K * K * Cin * Cout
print(f'conv N{N} H{H} {Cin}->{Cout} k{K} tile{tile} pro{pro} bf3={bf3}: {ms * 1e3:.1f} us  {fl / ms / 1e9:.1f} TF/s')
