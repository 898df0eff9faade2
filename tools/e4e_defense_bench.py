"""GPU tool: the full-size e4e + StyleGAN2 defender attack step (rows a14-a17 + a13): 256x256 image -> IR-SE50 encoder -> 18x512
latents mixed with mapped noise -> StyleGAN2 1024x1024 synthesis -> face_pool 256 -> ResNet-50 logits, and the input gradient.
Random weights of the reference architecture.  Prints the step time, conv TFLOP/s and the time by plan section / op kind.
    python tools/e4e_defense_bench.py [rows=8] [eot=4] [tune.json]"""
import sys, os, json, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gen_adversarial_amd import _lib as L
from gen_adversarial_amd.engine import Engine
from gen_adversarial_amd.e4e_spec import build_e4e_spec, init_e4e_state_dict
from gen_adversarial_amd.resnet_spec import build_resnet_spec, init_resnet_state_dict
from gen_adversarial_amd.stylegan_spec import build_stylegan_spec, init_stylegan_state_dict
from bench import conv_algorithmic_flops

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 8
eot = int(sys.argv[2]) if len(sys.argv) > 2 else 4
tune_out = sys.argv[3] if len(sys.argv) > 3 else None
SIZE = int(os.environ.get('GA_STYLEGAN_SIZE', '1024'))
espec, esd = build_e4e_spec(SIZE), init_e4e_state_dict(SIZE, 1, 0)
gspec = build_stylegan_spec(SIZE)
gsd = init_stylegan_state_dict(gspec, 1)
cspec, csd = build_resnet_spec(2), init_resnet_state_dict(2, 1, 2)
avg = torch.zeros(gspec.n_latent, gspec.style_dim)
alphas = [0.3] * gspec.n_latent
share = os.environ.get('GA_SHARE_ENCODER', '1') == '1'          # exact without input noise (the API's default)
eng = Engine.bare(rows, device='cuda:0', rep=eot, resolution=(3, 256, 256), alphas=alphas, share_encoder=share)
eng.build_e4e_defense(esd, espec, gsd, gspec, avg, csd, cspec, pool_to=256)
print(f'built: {len(eng.fwd)} + {len(eng.bwd)} ops, {eng.bytes / 1e9:.1f} GB engine buffers, {eng.store.bytes / 1e9:.2f} GB weights', flush=True)
eng.x_in.uniform_()
eng.eps[0].normal_()
eng.forward(); eng.dlogits.normal_(); eng.backward(); torch.cuda.synchronize()
assert torch.isfinite(eng.logits).all() and torch.isfinite(eng.dx).all()
print('logits', eng.logits.flatten()[:4].tolist(), '|dx|', eng.dx.abs().max().item(), flush=True)
if tune_out:
    from gen_adversarial_amd.engine_core import conv_key, tune_cache
    cache = dict(tune_cache())
    if os.environ.get('GA_RETUNE_3X3', '0') == '1':          # new halo tiles: time this engine's 3x3 / stride-1 shapes again
        for d in eng._conv_descs():
            if d.KH == 3 and d.KW == 3 and d.sn == 1 and d.sd == 1 and d.C2 == 0:
                cache.pop(conv_key(d), None)
    eng.autotune(cache=cache, reps=3, save=tune_out, verbose=False)
    print('tuned', flush=True)
s = eng.stream()
f_ms, fc_ms, fn = eng.fwd.time(s, iters=2, per_conv=True)
b_ms, bc_ms, bn = eng.bwd.time(s, iters=2, per_conv=True)
fl = conv_algorithmic_flops(eng.fwd) + conv_algorithmic_flops(eng.bwd)
print(json.dumps({'rows': rows, 'eot': eot, 'encoder_shared_by_eot_replicas': bool(eng.share_encoder), 'fwd_ms': f_ms, 'bwd_ms': b_ms, 'rows_per_s': rows / (f_ms + b_ms) * 1e3,
                  'conv_ms': fc_ms + bc_ms, 'conv_tflops': fl / (fc_ms + bc_ms) / 1e9, 'gflop_per_row': fl / rows / 1e9,
                  'gb': eng.bytes / 1e9, 'ops': len(eng.fwd) + len(eng.bwd)}), flush=True)


def section(name):
    if name.startswith('e4e.'): return 'encoder'
    if name.startswith('resnet.'): return 'classifier'
    if name.startswith('sg.mapping'): return 'mapping'
    if name.split('.')[0] in ('conv1', 'to_rgb1', 'convs', 'to_rgbs'): return 'generator'
    return 'glue'


for plan, tag in ((eng.fwd, 'fwd'), (eng.bwd, 'bwd')):
    ms = plan.profile(s); ms = plan.profile(s)
    by_sec, by_kind = collections.OrderedDict(), collections.OrderedDict()
    for d, nm, t in zip(plan.descs, plan.names, ms):
        by_sec[section(nm)] = by_sec.get(section(nm), 0.0) + t
        if section(nm) == 'generator':
            k = type(d).__name__ + ('' if not isinstance(d, L.ConvDesc) else f' K{d.KH}s{d.sn}')
            e = by_kind.setdefault(k, [0, 0.0]); e[0] += 1; e[1] += t
    print(f'==== {tag}: ' + '  '.join(f'{k} {v:.2f} ms' for k, v in by_sec.items()))
    for k, (n, t) in sorted(by_kind.items(), key=lambda kv: -kv[1][1]):
        print(f'   generator {k:28s} n {n:3d} {t:9.3f} ms')
    top = sorted(zip(ms, plan.names), reverse=True)[:12]
    print('   top ops: ' + ', '.join(f'{n} {t:.2f}' for t, n in top))
    enc = collections.OrderedDict()                              # the encoder by op kind and shape
    for d, nm, t in zip(plan.descs, plan.names, ms):
        if section(nm) != 'encoder':
            continue
        k = type(d).__name__
        if isinstance(d, L.ConvDesc):
            k += f' K{d.KH}s{d.sn}d{d.sd} {d.N * d.Ho * d.Wo}x{d.C1 + d.C2}->{d.Cout}'
        e = enc.setdefault(k, [0, 0.0]); e[0] += 1; e[1] += t
    for k, (n, t) in sorted(enc.items(), key=lambda kv: -kv[1][1])[:14]:
        print(f'   encoder {k:44s} n {n:3d} {t:9.3f} ms')

if os.environ.get('GA_GRAPHS', '0') == '1':                 # eager plan replay vs HIP graphs, wall clock per attack step
    def wall(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(n):
            eng.forward(); eng.backward()
        e1.record(); e1.synchronize()
        return e0.elapsed_time(e1) / n
    wall(1)
    eager = wall(4)
    eng.enable_graphs()
    wall(1)
    graph = wall(4)
    print(json.dumps({'step_ms_eager': eager, 'step_ms_graphs': graph}), flush=True)
