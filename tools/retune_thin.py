"""GPU tool (round 4): tile 11 (csrc/conv_thin3.hip) is a new candidate for the 3x3 layers with 32 (later also 64) input channels — time those shapes
again on every engine the bench builds (NVAE + VGG at 1024 / 512 / 256 / 32 rows, literal and shared encoder; the configs[2] and
configs[4] defenders) and write the merged table.   python tools/retune_thin.py [out.json]"""
import gc
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import build_e4e_defender, build_model, build_trans_defender
from gen_adversarial_amd.engine_core import conv_key, tune_cache

out = sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/r04_tune_thin.json'
cache = dict(tune_cache())
dev = 'cuda:0'


def retune(eng, what):
    only = int(os.environ.get('GA_RETUNE_C1', '0'))              # e.g. 64: only the shapes the 64-channel form of tile 11 takes
    keys = {conv_key(d) for d in eng._conv_descs() if id(d) in eng._thin_ok and (not only or d.C1 == only)}
    before = {k: cache.get(k) for k in keys}
    for k in keys:
        cache.pop(k, None)
    eng.autotune(cache=cache, reps=5, verbose=False)
    for k in sorted(keys):
        print(f'{what}: {k}: {before[k]} -> {cache[k]}', flush=True)


for rows, share in ((1024, False), (512, False), (256, False), (32, False), (32, True), (512, True)):
    eng, _ = build_model(dev, rows, 32, share_encoder=share)
    eng.x_in.uniform_()
    for e in eng.eps:
        e.normal_()
    eng.forward(); eng.dlogits.normal_(); eng.backward(); torch.cuda.synchronize()
    retune(eng, f'nvae {rows} rows' + (' shared encoder' if share else ''))
    del eng
    gc.collect(); torch.cuda.empty_cache()
for name, build, rows in (('configs[2] e4e', build_e4e_defender, 32), ('configs[4] trans', build_trans_defender, 64)):
    eng, _ = build(dev, rows, 32, 'bf16x3')
    eng.x_in.uniform_()
    eng.eps[0].normal_()
    if eng.noise is not None:
        eng.noise.normal_()
        eng.noise_coef.copy_(eng.noise_eps / eng.noise.flatten(1).norm(dim=1))
    eng.forward(); eng.dlogits.normal_(); eng.backward(); torch.cuda.synchronize()
    retune(eng, name)
    del eng
    gc.collect(); torch.cuda.empty_cache()
with open(out, 'w') as f:
    json.dump(cache, f, indent=0, sort_keys=True)
print('entries', len(cache))
