"""
Golden vectors for the A-VAE competitor defender, produced by IMPORTING THE REFERENCE (read-only at
/root/reference) in the build container.  Only the .npz travels.

    python tests/golden/make_avae_golden.py          (~20 s)

What runs is the reference's own Python: `StyledGenerator(64)` (src/defenses/competitors/a_vae/model.py:108-141) with its Encoder,
Generator, StyledConvBlocks, equal-lr hooks, FusedUpsample, Blur and AdaIN (modules.py), and `AVaeDefenseModel.purify`
(purification_model.py:16-20).  `purification_model.py` uses `torch` without importing it: the name is injected
into the module — no arithmetic is touched.  The random draws are made explicit: `torch.randn_like` is patched while `purify`
runs (the latent sample eps, model.py:82) and the per-block noise images are passed through the `noise=` argument of
`StyledGenerator.forward` (its default draws them with `.cuda()`, model.py:131-135) by wrapping the purifier call.
Weights: gen_adversarial_amd.avae_spec.init_avae_state_dict, load_state_dict(strict=True).
"""
import os
import sys

sys.dont_write_bytecode = True     # importing the reference must not leave __pycache__ in /root/reference (read-only tree)

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, '/root/reference')        # the reference's `src` package shadows the repo's import shim of the same name

from src.defenses.competitors.a_vae.model import StyledGenerator                      # noqa: E402
import importlib.util                                                                   # noqa: E402
_spec = importlib.util.spec_from_file_location('ref_avae_purification', '/root/reference/src/defenses/competitors/a_vae/purification_model.py')
_mod = importlib.util.module_from_spec(_spec)
_mod.torch = torch                           # the file's missing `import torch`
_spec.loader.exec_module(_mod)
AVaeDefenseModel = _mod.AVaeDefenseModel

from gen_adversarial_amd.avae_spec import build_avae_spec, init_avae_state_dict        # noqa: E402

SIZE, KERNEL, SEED, B = 64, 2, 41, 2


def main():
    spec = build_avae_spec(SIZE)
    m = StyledGenerator(SIZE)
    m.load_state_dict(init_avae_state_dict(SIZE, SEED), strict=True)
    m.eval()
    g = torch.Generator().manual_seed(SEED + 1)
    x = torch.rand(B, 3, SIZE, SIZE, generator=g).requires_grad_(True)
    eps = torch.randn(B, spec.c512, 4, 4, generator=g)
    noise = [torch.randn(B, 1, 4 * 2 ** i, 4 * 2 ** i, generator=g) for i in range(len(spec.blocks))]
    purifier = lambda t, inference=False: m(t, noise=noise, inference=inference)          # noqa: E731
    model = AVaeDefenseModel(lambda t: t, purifier, KERNEL)
    real = torch.randn_like
    torch.randn_like = lambda t, **kw: eps
    try:
        pur = model.purify(x)
    finally:
        torch.randn_like = real
    cot = torch.randn(pur.shape, generator=g)
    (gx,) = torch.autograd.grad((pur * cot).sum(), [x])
    out = {'size': np.int64(SIZE), 'kernel_size': np.int64(KERNEL), 'seed': np.int64(SEED), 'x': x.detach().numpy(), 'eps': eps.numpy(),
           'purified': pur.detach().numpy(), 'cot': cot.numpy(), 'gx': gx.numpy()}
    for i, n in enumerate(noise):
        out[f'noise{i}'] = n.numpy()
    print(f'purified {tuple(pur.shape)} in [{float(pur.min()):.3f}, {float(pur.max()):.3f}], |gx| max {float(gx.abs().max()):.3e}')
    np.savez_compressed(os.path.join(HERE, 'avae.npz'), **out)
    print('wrote', os.path.join(HERE, 'avae.npz'), f'{os.path.getsize(os.path.join(HERE, "avae.npz")) / 1024:.0f} KB')


if __name__ == '__main__':
    main()
