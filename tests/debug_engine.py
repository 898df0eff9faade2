"""Debug helper (GPU): compares engine intermediates against the oracle, cell by cell."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
from conftest import load_golden, golden_cfg
from gen_adversarial_amd.engine import Engine
from gen_adversarial_amd.nvae_spec import init_nvae_state_dict, build_spec
from gen_adversarial_amd.vgg_spec import build_vgg_spec, init_vgg_state_dict
from oracle import nvae_oracle as O, defender_oracle as D

name = sys.argv[1] if len(sys.argv) > 1 else 'A_cos07'
g = load_golden(f'nvae_{name}.npz')
cfg, res = golden_cfg(g)
sd = init_nvae_state_dict(cfg, res, int(g['nvae_seed']))
spec = build_spec(cfg, res)
vs = build_vgg_spec(int(g['n_classes']), int(g['width_div'])); vsd = init_vgg_state_dict(int(g['n_classes']), int(g['width_div']), int(g['vgg_seed']))
alphas = [float(a) * float(g['attenuation']) for a in g['alphas']]
x = torch.from_numpy(g['x'])
eps = [torch.from_numpy(g[f'eps_{i}']) for i in range(len(spec.groups))]

rec = {}
_enc, _dec = O.enc_cell, O.dec_cell
def enc(sd_, cell, x_):
    y = _enc(sd_, cell, x_); rec[cell.prefix + '.out'] = y; return y
def dec(sd_, cell, x_):
    y = _dec(sd_, cell, x_); rec[cell.prefix + '.out'] = y; return y
O.enc_cell, O.dec_cell = enc, dec
out, latents, logits = O.nvae_purify(sd, spec, x, alphas, eps, 0.6, return_latents=True)
for i, gs in enumerate(spec.groups):
    rec['z0' if i == 0 else f'z_{gs.s}:{gs.g}'] = latents[i]
rec['mix_logits'] = logits

eng = Engine(sd, cfg, res, vsd, vs, rows=x.shape[0], rep=1, alphas=alphas, device='cuda:0')
eng.x_in.copy_(x.cuda())
for b, e in zip(eng.eps, eps): b.copy_(e.cuda())
eng.forward(); torch.cuda.synchronize()
for k, a in eng.acts.items():
    if k in rec:
        r = rec[k]
        e = (a.t.permute(0, 3, 1, 2).cpu() - r).abs().max().item()
        print(f'{k:60s} err {e:.3e}  ref max {r.abs().max():.3e}')
print('purified err', (eng.purified.cpu() - out).abs().max().item())

# ---- detailed check of the first down cell
import torch.nn.functional as F
cell = spec.pre_cells[1]
p = cell.prefix
xin = eng.acts[spec.pre_cells[0].prefix + '.out'].t.permute(0, 3, 1, 2).cpu()
r = F.silu(O.bn_eval(sd, f'{p}.residual.0', xin))
r1 = O.wn_conv(sd, f'{p}.residual.2', r, stride=2, padding=1)
t1 = O.bn_eval(sd, f'{p}.residual.3', r1)
t2 = O.wn_conv(sd, f'{p}.residual.5', F.silu(t1), stride=1, padding=1)
sk = O.wn_conv(sd, f'{p}.skip_connection.conv', F.silu(xin), stride=2)
for nm, ref in (('.t1', t1), ('.t2', t2), ('.skip', sk)):
    a = eng.acts[p + nm].t.permute(0, 3, 1, 2).cpu()
    d = (a - ref).abs()
    print(nm, 'err', d.max().item(), 'argmax', np.unravel_index(d.argmax().item(), d.shape), 'shape', tuple(ref.shape))
    if nm == '.t1':
        print('per-channel max err', d.amax(dim=(0, 2, 3)))
        print('per-row max err', d.amax(dim=(0, 1, 3)))

i = eng.fwd.names.index(p + '.conv1')
dd = eng.fwd.descs[i]
print({f[0]: getattr(dd, f[0]) for f in dd._fields_ if f[1].__name__ in ('c_int',)})
from gen_adversarial_amd import folding as FO
w = FO.fold_enc_cell(sd, cell)
W = w['w1'].reshape(16, 3, 3, 8).permute(0, 3, 1, 2)
ref2 = F.conv2d(F.silu(xin * w['pro_scale'].view(1, -1, 1, 1) + w['pro_shift'].view(1, -1, 1, 1)), W, w['b1'], stride=2, padding=1)
print('folded-ref vs oracle', (ref2 - t1).abs().max().item())
a = eng.acts[p + '.t1'].t.permute(0, 3, 1, 2).cpu()
print('engine vs folded-ref', (a - ref2).abs().max().item())
