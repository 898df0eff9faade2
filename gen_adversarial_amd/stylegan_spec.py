"""
Parameter layout of the StyleGAN2 synthesis layers the e4e defender decodes with
(src/mlvgms_autoencoders/StyleGan_E4E/stylegan2/generator.py): `StyledConv` with and without up-sampling
(generator.py:229-265: ModulatedConv2d, NoiseInjection, FusedLeakyReLU), `ToRGB` (generator.py:268-290: 1x1 modulated conv
without demodulation, bias, up-sampled skip) and the synthesis network's wiring for `input_is_latent=True,
randomize_noise=False` (generator.py:399-470 as called by E4EStyleGanDefenseModel.decode, src/defenses/ours/models.py:346).
The mapping MLP (`style.*`, generator.py:306-317) turns the defender's Gaussian noise into the styles it mixes with the
encoder's codes (src/defenses/ours/models.py:118-124): forward only.

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
    upsample: bool = False   # StyledConv(upsample=True): stride-2 transposed conv + [1,3,3,1] blur; res = OUTPUT side


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


N_MLP, LR_MLP = 8, 0.01          # pSp builds Generator(size, 512, 8, channel_multiplier=2) (psp.py:25); lr_mlp default 0.01


@dataclass(frozen=True)
class StyleGanSpec:
    size: int                 # output side (power of two >= 8)
    style_dim: int
    n_latent: int             # log2(size) * 2 - 2 (generator.py:378)
    const_channels: int
    conv1: StyledConvSpec
    to_rgb1: StyledConvSpec
    convs: tuple              # (up, plain) per resolution, flattened: convs.0, convs.1, ...
    to_rgbs: tuple


def build_stylegan_spec(size: int, channel_multiplier: int = 2, width_div: int = 1, style_dim: int = 512) -> StyleGanSpec:
    """channel table of Generator.__init__ (generator.py:322-332), optionally divided (tests); layer order of :334-376"""
    import math
    log_size = int(math.log2(size))
    assert 2 ** log_size == size and size >= 8
    table = {4: 512, 8: 512, 16: 512, 32: 512, 64: 256 * channel_multiplier, 128: 128 * channel_multiplier,
             256: 64 * channel_multiplier, 512: 32 * channel_multiplier, 1024: 16 * channel_multiplier}
    ch = {k: max(4, v // width_div) for k, v in table.items()}
    D = style_dim
    conv1 = StyledConvSpec('conv1', ch[4], ch[4], 3, D, 4, True, True)
    rgb1 = StyledConvSpec('to_rgb1', ch[4], 3, 1, D, 4, False, False)
    convs, rgbs, cin = [], [], ch[4]
    for i in range(3, log_size + 1):
        r, co = 2 ** i, ch[2 ** i]
        convs.append(StyledConvSpec(f'convs.{len(convs)}', cin, co, 3, D, r, True, True, True))
        convs.append(StyledConvSpec(f'convs.{len(convs)}', co, co, 3, D, r, True, True))
        rgbs.append(StyledConvSpec(f'to_rgbs.{len(rgbs)}', co, 3, 1, D, r, False, False))
        cin = co
    return StyleGanSpec(size, D, log_size * 2 - 2, ch[4], conv1, rgb1, tuple(convs), tuple(rgbs))


def init_stylegan_state_dict(spec: StyleGanSpec, seed: int = 0) -> dict:
    """the synthesis network's parameters and fixed noise buffers under the reference's key names (Generator.state_dict();
    the constant blur kernels `*.blur.kernel` / `*.upsample.kernel` = [1,3,3,1] are not read)"""
    g = torch.Generator().manual_seed(seed)
    sd = {'input.input': torch.randn(1, spec.const_channels, 4, 4, generator=g)}
    for k, sp in enumerate((spec.conv1, spec.to_rgb1) + spec.convs + spec.to_rgbs):
        sd.update(init_styled_conv_state_dict(sp, seed * 1000 + k + 1))
    D = spec.style_dim
    for k in range(1, N_MLP + 1):                           # mapping network: PixelNorm + 8 EqualLinear(lr_mul=0.01, fused_lrelu)
        sd[f'style.{k}.weight'] = torch.randn(D, D, generator=g) / LR_MLP        # generator.py:74 (randn / lr_mul)
        sd[f'style.{k}.bias'] = 10.0 * torch.randn(D, generator=g)               # used as bias * lr_mul
    for i in range(1 + len(spec.convs)):                    # noise_0 at 4x4, then two per resolution (generator.py:349-352)
        r = 2 ** ((i + 5) // 2)
        sd[f'noises.noise_{i}'] = torch.randn(1, 1, r, r, generator=g)
    return sd
