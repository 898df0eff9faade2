"""GPU tool: e4e encoder (row a14) attack step = image -> 18x512 latents + input gradient at 256x256."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gen_adversarial_amd.engine import Engine
from gen_adversarial_amd.e4e_spec import build_e4e_spec, init_e4e_state_dict
from bench import conv_algorithmic_flops

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 16
tune_out = sys.argv[2] if len(sys.argv) > 2 else None
spec = build_e4e_spec(1024)
sd = init_e4e_state_dict(1024, 1, 0)
eng = Engine(None, None, (3, 256, 256), sd, spec, rows=rows, rep=1, alphas=[], device='cuda:0')
eng.x_in.uniform_()
eng.forward(); eng.dlogits.normal_(); eng.backward(); torch.cuda.synchronize()
if tune_out:
    eng.autotune(reps=5, save=tune_out, verbose=False)
s = eng.stream()
f_ms, fc_ms, fn = eng.fwd.time(s, iters=3, per_conv=True)
b_ms, bc_ms, bn = eng.bwd.time(s, iters=3, per_conv=True)
fl = conv_algorithmic_flops(eng.fwd) + conv_algorithmic_flops(eng.bwd)
print(json.dumps({'rows': rows, 'fwd_ms': f_ms, 'bwd_ms': b_ms, 'rows_per_s': rows / (f_ms + b_ms) * 1e3,
                  'conv_tflops': fl / (fc_ms + bc_ms) / 1e9, 'gflop_per_row': fl / rows / 1e9, 'gb': eng.bytes / 1e9,
                  'launches': int(fn + bn), 'ops': len(eng.fwd) + len(eng.bwd)}))
