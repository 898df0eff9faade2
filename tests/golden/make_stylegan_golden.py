"""
Golden vectors for the StyleGAN2 modulated convolution (first slice), produced by IMPORTING THE
REFERENCE's `ModulatedConv2d` (read-only at /root/reference) in the build container.  Only the .npz travels.

    python tests/golden/make_stylegan_golden.py

Container-only shim, as in make_e4e_golden.py: `stylegan2.op` is replaced by a stub BEFORE the import (the real module
JIT-compiles CUDA sources into the reference tree).  The non-resampling ModulatedConv2d path never calls those ops.
Three cases: the 3x3 demodulated conv of a StyledConv, the 1x1 non-demodulated conv of a ToRGB, and the transposed conv of
an up-sampling StyledConv (before its blur); forward, d/dx and d/dstyle under a fixed cotangent.
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
    raise RuntimeError('stylegan2.op is stubbed: this path does not use it')


op = types.ModuleType('src.mlvgms_autoencoders.StyleGan_E4E.stylegan2.op')
op.fused_leaky_relu = _absent
op.upfirdn2d = _absent
op.FusedLeakyReLU = type('FusedLeakyReLU', (torch.nn.Module,), {'forward': _absent})
sys.modules['src.mlvgms_autoencoders.StyleGan_E4E.stylegan2.op'] = op

from src.mlvgms_autoencoders.StyleGan_E4E.stylegan2.generator import ModulatedConv2d     # noqa: E402
from gen_adversarial_amd.stylegan_spec import StyledConvSpec, init_styled_conv_state_dict   # noqa: E402

CASES = {'styled': StyledConvSpec('conv1', 32, 64, 3, 64, 8, True, True),
         'torgb': StyledConvSpec('to_rgb1', 64, 3, 1, 64, 8, False, False),
         # the up-sampling conv with its Blur module replaced by the identity (the blur is the stubbed CUDA op): the
         # transposed convolution of generator.py:178-188 alone, output (2*8+1)^2
         'up': StyledConvSpec('convs.0', 32, 64, 3, 64, 16, True, True, True)}
SEED, ROWS = 31, 3

if __name__ == '__main__':
    out = {}
    for name, sp in CASES.items():
        sd = init_styled_conv_state_dict(sp, SEED)
        m = ModulatedConv2d(sp.cin, sp.cout, sp.kernel, sp.style_dim, demodulate=sp.demodulate, upsample=sp.upsample)
        if sp.upsample:
            m.blur = torch.nn.Identity()
        m.load_state_dict({'weight': sd[f'{sp.prefix}.conv.weight'], 'modulation.weight': sd[f'{sp.prefix}.conv.modulation.weight'],
                           'modulation.bias': sd[f'{sp.prefix}.conv.modulation.bias']}, strict=not sp.upsample)
        g = torch.Generator().manual_seed(7)
        r_in = sp.res // 2 if sp.upsample else sp.res
        x = torch.randn(ROWS, sp.cin, r_in, r_in, generator=g).requires_grad_(True)
        w = torch.randn(ROWS, sp.style_dim, generator=g).requires_grad_(True)
        y = m(x, w)
        cot = torch.randn(y.shape, generator=g)
        gx, gw = torch.autograd.grad((y * cot).sum(), [x, w])
        for k, v in (('x', x), ('w', w), ('y', y), ('cot', cot), ('gx', gx), ('gw', gw)):
            out[f'{name}.{k}'] = v.detach().numpy()
        print(name, 'y', tuple(y.shape), float(y.abs().max()), '|gx|', float(gx.abs().max()), '|gw|', float(gw.abs().max()))
    np.savez_compressed(os.path.join(HERE, 'stylegan_modconv.npz'), seed=SEED, **out)
