"""GPU tool: slowest ops of the ResNet-50 classifier plans."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gen_adversarial_amd.engine import Engine
from gen_adversarial_amd.resnet_spec import build_resnet_spec, init_resnet_state_dict
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 64
next50 = len(sys.argv) > 2 and sys.argv[2] == 'resnext'
spec = build_resnet_spec(4, 1, (3, 4, 6, 3), 32, 4) if next50 else build_resnet_spec(2)
sd = init_resnet_state_dict(4, 1, 0, (3, 4, 6, 3), 32, 4) if next50 else init_resnet_state_dict(2, 1, 0)
res = 128 if next50 else 256
eng = Engine(None, None, (3, res, res), sd, spec, rows=rows, rep=1, alphas=[], device='cuda:0')
eng.x_in.uniform_(); eng.forward(); eng.dlogits.normal_(); eng.backward(); torch.cuda.synchronize()
s = eng.stream()
for plan, tag in ((eng.fwd, 'fwd'), (eng.bwd, 'bwd')):
    ms = plan.profile(s); ms = plan.profile(s)
    print(f'==== {tag}: total {sum(ms):.2f} ms')
    for t, nm in sorted(zip(ms, plan.names), reverse=True)[:14]:
        print(f'  {t:8.3f} ms  {nm}')
