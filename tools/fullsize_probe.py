"""GPU probe: build the full-size (assumed-config) engine at R rows, time forward/backward and the conv share."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gen_adversarial_amd.engine import Engine
from gen_adversarial_amd.nvae_spec import ASSUMED_NVAE_CONFIG, ASSUMED_NVAE_RESOLUTION, init_nvae_state_dict, build_spec
from gen_adversarial_amd.vgg_spec import build_vgg_spec, init_vgg_state_dict

R = int(sys.argv[1]) if len(sys.argv) > 1 else 256
wd = int(sys.argv[2]) if len(sys.argv) > 2 else 1
t0 = time.time()
sd = init_nvae_state_dict(ASSUMED_NVAE_CONFIG, ASSUMED_NVAE_RESOLUTION, 0)
print('nvae init', time.time() - t0, flush=True); t0 = time.time()
vs = build_vgg_spec(100, wd); vsd = init_vgg_state_dict(100, wd, 1)
print('vgg init', time.time() - t0, flush=True); t0 = time.time()
spec = build_spec(ASSUMED_NVAE_CONFIG, ASSUMED_NVAE_RESOLUTION)
alphas = [0.7 * i / (len(spec.groups) - 1) for i in range(len(spec.groups))]
eng = Engine(sd, ASSUMED_NVAE_CONFIG, ASSUMED_NVAE_RESOLUTION, vsd, vs, rows=R, rep=32, alphas=alphas, device='cuda:0')
torch.cuda.synchronize()
print('engine build', time.time() - t0, 'bytes GB', eng.bytes / 1e9, 'ops', len(eng.fwd), len(eng.bwd), flush=True)
eng.x_in.copy_(torch.rand_like(eng.x_in))
for e in eng.eps: e.normal_()
s = eng.stream()
for it in range(2):
    eng.forward(); eng.dlogits.normal_(); eng.backward()
torch.cuda.synchronize()
f_ms, fc_ms, fn = eng.fwd.time(s, iters=3, per_conv=True)
b_ms, bc_ms, bn = eng.bwd.time(s, iters=3, per_conv=True)
print(f'fwd {f_ms:.2f} ms (conv {fc_ms:.2f} ms in {fn} launches)  bwd {b_ms:.2f} ms (conv {bc_ms:.2f} ms in {bn} launches)')
print(f'rows/s fwd+bwd: {R / ((f_ms + b_ms) / 1e3):.1f}')
print('logits finite', torch.isfinite(eng.logits).all().item(), 'dx finite', torch.isfinite(eng.dx).all().item(), eng.dx.abs().max().item())
