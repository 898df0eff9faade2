"""GPU experiment: K independent row chunks, each with its own engine + HIP stream, run concurrently vs one stream."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import build_model
from gen_adversarial_amd.engine import Engine, WeightStore
from gen_adversarial_amd.nvae_spec import ASSUMED_NVAE_CONFIG, ASSUMED_NVAE_RESOLUTION

R = int(sys.argv[1]); K = int(sys.argv[2]); iters = int(sys.argv[3]) if len(sys.argv) > 3 else 6
eng0, (sd, vsd, vspec, alphas) = build_model('cuda:0', R, 32)
engs = [eng0]
for _ in range(K - 1):
    engs.append(Engine(sd, ASSUMED_NVAE_CONFIG, ASSUMED_NVAE_RESOLUTION, vsd, vspec, rows=R, rep=32, alphas=alphas,
                       temperature=0.6, noise_eps=0.0, device='cuda:0', store=eng0.store))
streams = [torch.cuda.Stream() for _ in engs]
x = torch.rand(R // 32, 3, 64, 64, device='cuda')

def step():
    for e, s in zip(engs, streams):
        with torch.cuda.stream(s):
            e.x_in.copy_(x)
            for z in e.eps: z.normal_()
            e.forward()
            e.dlogits.normal_()
            e.backward()

for _ in range(2): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(iters): step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / iters
print(f'R={R} x {K} streams: {dt * 1e3:.1f} ms per round, {R * K / dt:.0f} rows/s', flush=True)
