"""GPU tool: wall time per forward+backward, eager plan replay vs HIP graph launch, at a given row count."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import build_model
R = int(sys.argv[1]) if len(sys.argv) > 1 else 32
eng, _ = build_model('cuda:0', R, 32)
eng.x_in.uniform_()
for e in eng.eps: e.normal_()
def run(n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        eng.forward(); eng.backward()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
run(3); e = run(10)
eng.enable_graphs(); run(3); g = run(10)
print(f'R={R}: eager {e:.2f} ms/step, graphs {g:.2f} ms/step ({R / g * 1e3:.0f} rows/s)')
