"""GPU tool: ResNet-50 classifier (CelebaGenderClassifier path) attack step = forward + input gradient at 256x256."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gen_adversarial_amd.engine import Engine
from gen_adversarial_amd.resnet_spec import build_resnet_spec, init_resnet_state_dict
from bench import conv_algorithmic_flops

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 64
tune_out = sys.argv[2] if len(sys.argv) > 2 and sys.argv[2] != '-' else None
next50 = len(sys.argv) > 3 and sys.argv[3] == 'resnext'         # resnext50_32x4d at the cars resolution
res = 128 if next50 else 256
spec = build_resnet_spec(4, 1, (3, 4, 6, 3), 32, 4) if next50 else build_resnet_spec(2)
sd = init_resnet_state_dict(4, 1, 0, (3, 4, 6, 3), 32, 4) if next50 else init_resnet_state_dict(2, 1, 0)
eng = Engine(None, None, (3, res, res), sd, spec, rows=rows, rep=1, alphas=[], device='cuda:0')
eng.x_in.uniform_()
eng.forward(); eng.dlogits.normal_(); eng.backward(); torch.cuda.synchronize()
if tune_out:
    eng.autotune(reps=5, save=tune_out, verbose=False)
s = eng.stream()
f_ms, fc_ms, fn = eng.fwd.time(s, iters=5, per_conv=True)
b_ms, bc_ms, bn = eng.bwd.time(s, iters=5, per_conv=True)
fl = conv_algorithmic_flops(eng.fwd) + conv_algorithmic_flops(eng.bwd)
print(json.dumps({'net': 'resnext50_32x4d@128' if next50 else 'resnet50@256', 'rows': rows, 'fwd_ms': f_ms, 'bwd_ms': b_ms, 'rows_per_s': rows / (f_ms + b_ms) * 1e3,
                  'conv_tflops': fl / (fc_ms + bc_ms) / 1e9, 'gflop_per_row': fl / rows / 1e9, 'gb_activations': eng.bytes / 1e9,
                  'launches': int(fn + bn)}))
