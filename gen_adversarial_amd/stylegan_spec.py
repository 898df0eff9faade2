"""
Parameter layout of the StyleGAN2 synthesis layers the e4e defender decodes with
(src/mlvgms_autoencoders/StyleGan_E4E/stylegan2/generator.py).  This round covers ONE layer family: the modulated
convolution without resampling — `StyledConv(upsample=False)` (generator.py:229-265: ModulatedConv2d, NoiseInjection,
FusedLeakyReLU) and the `ToRGB` convolution (generator.py:268-290, demodulate=False, 1x1, plain bias).  The upsampling
variant (transposed conv + blur), the mapping MLP and the generator's wiring are the next rows.

State-dict keys follow the reference module (prefix = the layer's name inside Generator, e.g. 'conv1' / 'convs.1'):
  {p}.conv.weight [1,Cout,Cin,k,k]   {p}.conv.modulation.weight [Cin,D]   {p}.conv.modulation.bias [Cin]
  {p}.noise.weight [1]               {p}.activate.bias [Cout]
and for ToRGB:  {p}.conv.weight [1,3,Cin,1,1]  {p}.conv.modulation.*  {p}.bias [1,3,1,1]
"""
from __future__ import annotations

from dataclasses import dataclass

import torch


@dataclass(frozen=True)
class StyledConvSpec:
    prefix: str
    cin: int
    cout: int
    kernel: int          # 3 (StyledConv) or 1 (ToRGB)
    style_dim: int
    res: int             # feature-map side
    demodulate: bool     # False for ToRGB
    activate: bool       # noise + FusedLeakyReLU (StyledConv) or plain bias (ToRGB)


def init_styled_conv_state_dict(spec: StyledConvSpec, seed: int = 0) -> dict:
    """random parameters with the reference's initial distributions, but a non-zero noise strength / biases so that
    every term of the layer is exercised"""
    g = torch.Generator().manual_seed(seed)
    p = spec.prefix
    sd = {f'{p}.conv.weight': torch.randn(1, spec.cout, spec.cin, spec.kernel, spec.kernel, generator=g),
          f'{p}.conv.modulation.weight': torch.randn(spec.cin, spec.style_dim, generator=g),
          f'{p}.conv.modulation.bias': 1.0 + 0.1 * torch.randn(spec.cin, generator=g)}
    if spec.activate:
        sd[f'{p}.noise.weight'] = 0.3 * torch.randn(1, generator=g)
        sd[f'{p}.activate.bias'] = 0.2 * torch.randn(spec.cout, generator=g)
    else:
        sd[f'{p}.bias'] = 0.2 * torch.randn(1, spec.cout, 1, 1, generator=g)
    return sd
