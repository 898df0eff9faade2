"""GPU tool: autotune (tile, split-K) of every conv shape of the bench engine; writes the table under gpurun_out/."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import build_model

R = int(sys.argv[1]) if len(sys.argv) > 1 else 256
out = sys.argv[2] if len(sys.argv) > 2 else 'gpurun_out/conv_tune_gfx950.json'
share = len(sys.argv) > 4 and sys.argv[4] == 'share'
eng, _ = build_model('cuda:0', R, int(os.environ.get('GA_TUNE_REP', '32')), share_encoder=share)
eng.x_in.uniform_()
for e in eng.eps: e.normal_()
eng.forward(); eng.dlogits.normal_(); eng.backward(); torch.cuda.synchronize()
s = eng.stream()
f0, fc0, _ = eng.fwd.time(s, iters=3, per_conv=True); b0, bc0, _ = eng.bwd.time(s, iters=3, per_conv=True)
print(f'before: fwd {f0:.2f} (conv {fc0:.2f})  bwd {b0:.2f} (conv {bc0:.2f})', flush=True)
fresh = len(sys.argv) > 3 and sys.argv[3] == 'fresh'
start = {} if fresh else None
if len(sys.argv) > 3 and sys.argv[3] == 'retune3x3':       # new halo tile codes: time this engine's 3x3 / stride-1 shapes again
    from gen_adversarial_amd.engine_core import conv_key, tune_cache
    start = dict(tune_cache())
    for d in eng._conv_descs():
        if d.KH == 3 and d.KW == 3 and d.sn == 1 and d.sd == 1 and d.C2 == 0:
            start.pop(conv_key(d), None)
if len(sys.argv) > 3 and sys.argv[3] == 'retune_thin':     # tile 11 (conv_thin3) is new: time the shapes it takes again
    from gen_adversarial_amd.engine_core import conv_key, tune_cache
    start = dict(tune_cache())
    for d in eng._conv_descs():
        if id(d) in eng._thin_ok:
            start.pop(conv_key(d), None)
cache = eng.autotune(cache=start, reps=5, save=out, verbose=True)     # default: add missing shapes
f1, fc1, _ = eng.fwd.time(s, iters=3, per_conv=True); b1, bc1, _ = eng.bwd.time(s, iters=3, per_conv=True)
print(f'after : fwd {f1:.2f} (conv {fc1:.2f})  bwd {b1:.2f} (conv {bc1:.2f})  entries {len(cache)}')
