"""GPU tool: the reference's protocol (ONE image x EoT 32 per defender call, src/experiments/test_defense.py:116) as PGD attack steps on
the full-size NVAE + VGG engine: the literal x.repeat(eot) path against the API default (encoder once per image when no input noise is
configured).  Prints one JSON line.   python tools/ref_protocol.py [images=1] [eot=32]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import AttackStep, build_model, _time_steps

images = int(sys.argv[1]) if len(sys.argv) > 1 else 1
eot = int(sys.argv[2]) if len(sys.argv) > 2 else 32
dev = 'cuda:0'
out = {'images': images, 'eot': eot}
store = None
for share in (False, True):
    e, _ = build_model(dev, images * eot, eot, seed=0, share_encoder=share, store=store)
    store = e.store
    x = torch.rand(images, 3, 64, 64, device=dev)
    st = AttackStep([e], [torch.cuda.Stream(device=dev)], torch.zeros(images, dtype=torch.long, device=dev), x)
    t = _time_steps(st, 20, warm=3)
    out['shared_encoder' if share else 'literal_repeat'] = {'rows_per_s': images * eot / t, 'ms_per_step': t * 1e3,
                                                           'launches': len(e.fwd) + len(e.bwd)}
    del st, e
print(json.dumps(out))
