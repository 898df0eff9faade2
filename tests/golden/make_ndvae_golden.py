"""
Golden vectors for the ND-VAE competitor defender, produced by IMPORTING THE REFERENCE (read-only at
/root/reference) in the build container.  Only the .npz travels.

    python tests/golden/make_ndvae_golden.py          (~10 s)

What runs is the reference's own Python: `Defence_NVAE` (src/defenses/competitors/nd_vae/modules/models/NVAE.py:639-720) with its
cells, towers and samplers, `DiscMixLogistic.mean` (NVAE_utils.py) and `NDVaeDefenseModel.purify`
(src/defenses/competitors/nd_vae/purification_model.py:18-26).
Shims: name-only stand-ins for `tkinter`, `turtle` (NVAE.py:1-2 imports them and uses nothing) and `torchvision` (absent; only
the commented-out training function would use it).  The random draws are made explicit without touching the arithmetic:
  * `Normal.sample` returns `self.sample_given_eps(eps)` — the reference's own method (NVAE.py:100-101) — for the next eps of a
    recorded list instead of drawing inside the scripted `sample_normal_jit`;
  * `torch.randn_like` is patched while `purify` runs (the input noise);
  * `Decoder_tower.h`, an unregistered random tensor (ndvae_spec.py header), is set to a recorded draw.
Weights: gen_adversarial_amd.ndvae_spec.init_ndvae_state_dict (primary keys; strict=False because every cell also lists its
layers under alias names that share the same tensors — checked below).
"""
import contextlib
import io
import os
import re
import sys

sys.dont_write_bytecode = True     # importing the reference must not leave __pycache__ in /root/reference (read-only tree)
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference'
sys.path.insert(0, REPO)
sys.path.insert(0, REF)                      # the reference's `src` package shadows the repo's import shim of the same name
for _name in ('tkinter', 'turtle'):
    _m = types.ModuleType(_name)
    _m.W = _m.forward = None
    sys.modules[_name] = _m
_tv = types.ModuleType('torchvision')
_tv.datasets, _tv.transforms = types.ModuleType('torchvision.datasets'), types.ModuleType('torchvision.transforms')
sys.modules.update({'torchvision': _tv, 'torchvision.datasets': _tv.datasets, 'torchvision.transforms': _tv.transforms})

with contextlib.redirect_stdout(io.StringIO()):
    import src.defenses.competitors.nd_vae.modules.models.NVAE as ref
    from src.defenses.competitors.nd_vae.purification_model import NDVaeDefenseModel

from gen_adversarial_amd.ndvae_spec import build_ndvae_spec, init_ndvae_h, init_ndvae_state_dict   # noqa: E402

CASES = {
    # two latent scales (the gender / cars layout at reduced width), 32-px input
    'A': ({'x_channels': 3, 'encoding_channels': 4, 'pre_proc_groups': 2, 'scales': 2, 'groups': 2, 'cells': 2, 'input_dim': 32}, 0.1, 11),
    # one latent scale (the ids layout: configs/competitor_ndvae_ids.yaml), 32-px input -> h is 8 x 8
    'B': ({'x_channels': 3, 'encoding_channels': 8, 'pre_proc_groups': 2, 'scales': 1, 'groups': 2, 'cells': 1, 'input_dim': 32}, 0.05, 12),
}


def build(cfg, seed):
    with contextlib.redirect_stdout(io.StringIO()):
        m = ref.Defence_NVAE(cfg['x_channels'], cfg['encoding_channels'], cfg['pre_proc_groups'], cfg['scales'], cfg['groups'],
                             cfg['cells'], cfg['input_dim'])
    sd = init_ndvae_state_dict(cfg, seed)
    r = m.load_state_dict(sd, strict=False)
    assert not r.unexpected_keys
    alias = re.compile(r'\.cell\.\d+|^pre_proc\.tower\.|num_batches_tracked$')
    assert all(alias.search(k) for k in r.missing_keys), [k for k in r.missing_keys if not alias.search(k)][:5]
    full = m.state_dict()
    assert all(torch.equal(full[k], v) for k, v in sd.items())
    h = init_ndvae_h(cfg, seed + 100)
    m.decoder.h = h.unsqueeze(0)
    return m.eval(), sd, h


def main():
    out = {}
    for name, (cfg, noise_std, seed) in CASES.items():
        spec = build_ndvae_spec(cfg)
        m, sd, h = build(cfg, seed)
        g = torch.Generator().manual_seed(seed + 1)
        B, D = 3, cfg['input_dim']
        x = torch.rand(B, 3, D, D, generator=g).requires_grad_(True)
        noise = torch.randn(B, 3, D, D, generator=g)
        eps = [torch.randn(B, c, r, r, generator=g) for c, r in spec.latent_shapes]
        it = iter(eps)
        ref.Normal.sample = lambda self: (self.sample_given_eps(next(it)), None)
        real = torch.randn_like
        torch.randn_like = lambda t, **kw: noise
        try:
            model = NDVaeDefenseModel(lambda t: t, m, noise_std)
            pur = model.purify(x)
        finally:
            torch.randn_like = real
        assert next(it, None) is None, 'a sampler drew more / fewer eps than the spec lists'
        cot = torch.randn(pur.shape, generator=g)
        (gx,) = torch.autograd.grad((pur * cot).sum(), [x])
        # the mixture logits of the same forward, without the input noise (Defence_NVAE.forward alone)
        it = iter(eps)
        logits = m(x.detach())[0]
        out.update({f'{name}.cfg_keys': np.array(list(cfg.keys())), f'{name}.cfg_vals': np.array(list(cfg.values())),
                    f'{name}.noise_std': np.float32(noise_std), f'{name}.seed': np.int64(seed),
                    f'{name}.x': x.detach().numpy(), f'{name}.noise': noise.numpy(), f'{name}.h': h.numpy(),
                    f'{name}.purified': pur.detach().numpy(), f'{name}.cot': cot.numpy(), f'{name}.gx': gx.numpy(),
                    f'{name}.logits_clean': logits.detach().numpy()})
        for i, e in enumerate(eps):
            out[f'{name}.eps{i}'] = e.numpy()
        print(f'case {name}: purified {tuple(pur.shape)} in [{float(pur.min()):.3f}, {float(pur.max()):.3f}], |gx| max {float(gx.abs().max()):.3e}, '
              f'{len(eps)} samplers, latent shapes {spec.latent_shapes}')
    np.savez_compressed(os.path.join(HERE, 'ndvae.npz'), **out)
    print('wrote', os.path.join(HERE, 'ndvae.npz'), f'{os.path.getsize(os.path.join(HERE, "ndvae.npz")) / 1024:.0f} KB')


if __name__ == '__main__':
    main()
