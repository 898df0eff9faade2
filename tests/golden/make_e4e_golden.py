"""
Golden vectors for the e4e encoder, produced by IMPORTING THE REFERENCE's `Encoder4Editing`
(read-only at /root/reference) in the build container.  Only the .npz travels.

    python tests/golden/make_e4e_golden.py          (run from anywhere; ~1 min on 8 cores)

Container-only shim: `src.mlvgms_autoencoders.StyleGan_E4E.stylegan2.op` is replaced by a stub module BEFORE the
reference is imported — importing the real one JIT-compiles CUDA sources and would write into the reference tree
(SURVEY.md §0.4).  The encoder never calls those ops (its EqualLinear has activation=None), so the stub bodies raise.
Weights: gen_adversarial_amd.e4e_spec's seeded initialiser, loaded with load_state_dict(strict=True), which also pins
our key names and shapes.  The reference fixes the channel widths (64..512), so the golden runs the full-width IR-SE50
on a 64x64 input with stylegan_size=64 (10 style heads: every head kind, both FPN levels).
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference'
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
sys.path.insert(1, REPO)


def _absent(*a, **k):
    raise RuntimeError('stylegan2.op is stubbed: the encoder does not use it')


op = types.ModuleType('src.mlvgms_autoencoders.StyleGan_E4E.stylegan2.op')
op.fused_leaky_relu = _absent
op.upfirdn2d = _absent


class FusedLeakyReLU(torch.nn.Module):
    def __init__(self, *a, **k):
        super().__init__()

    forward = _absent


op.FusedLeakyReLU = FusedLeakyReLU
sys.modules['src.mlvgms_autoencoders.StyleGan_E4E.stylegan2.op'] = op

from src.mlvgms_autoencoders.StyleGan_E4E.encoding.encoder import Encoder4Editing      # noqa: E402
from gen_adversarial_amd.e4e_spec import init_e4e_state_dict                             # noqa: E402

assert not any(f.endswith('.hip') for _, _, fs in os.walk(REF) for f in fs), 'reference tree was modified'

if __name__ == '__main__':
    torch.set_num_threads(8)
    SIZE, SEED, RES, B = 64, 21, 64, 2
    enc = Encoder4Editing(50, 'ir_se', types.SimpleNamespace(stylegan_size=SIZE))
    enc.load_state_dict(init_e4e_state_dict(SIZE, 1, SEED), strict=True)
    enc.eval()
    g = torch.Generator().manual_seed(5)
    x = torch.rand(B, 3, RES, RES, generator=g).requires_grad_(True)
    w = enc(x)
    cot = torch.randn(w.shape, generator=g)
    (gx,) = torch.autograd.grad((w * cot).sum(), [x])
    np.savez_compressed(os.path.join(HERE, 'e4e_ir50_s64.npz'), stylegan_size=SIZE, seed=SEED, x=x.detach().numpy(),
                        w=w.detach().numpy(), cot=cot.numpy(), gx=gx.numpy())
    print('e4e golden: w', tuple(w.shape), 'max', float(w.abs().max()), '|gx|', float(gx.abs().max()))
