"""
Structural description of the A-VAE competitor purifier (`StyledGenerator`, a StyleGAN-v1-like auto-encoder) and a seeded
parameter initialiser with the reference's key names.

Reference for the module tree:
  src/defenses/competitors/a_vae/model.py      Encoder :9-27, Generator :30-105 (forward :73-105), StyledGenerator :108-141
  src/defenses/competitors/a_vae/modules.py    EqualLR :8-35 (weight = weight_orig * sqrt(2 / fan_in) at every forward),
      FusedUpsample :38-65, Blur :142-156, EqualConv2d :159-169, EqualLinear :172-182, AdaptiveInstanceNorm :278-296,
      NoiseInjection :299-306, ConstantInput :309-320, StyledConvBlock :323-381, EncodeConvBlock :384-416
  src/defenses/competitors/a_vae/purification_model.py:16-25 (avg_pool2d(x * 2 - 1, kernel_size) -> purifier(inference=True) -> (x + 1) / 2)
  src/experiments/load_defense.py:95-106 (StyledGenerator(args.image_size), kernel_size from the yaml)
Quirks of the reference that are reproduced, not repaired:
  * `EncodeConvBlock.forward` calls `self.norm1(out)` / `self.norm2(out)` and drops the result: the encoder has NO normalisation;
  * the first generator block starts from `ConstantInput`: the sampled latent reaches the image only through the style MLP
    (AdaIN scales / shifts) and the encoder's 16 x 16 skip tensor, concatenated where the resolutions meet;
  * `inject_index = [len(progression) + 1]`: one style for every block.
`width_div` shrinks every channel count (512 / 256 / 128 and the 512-wide style) for cheap structural tests; the reference is
width_div = 1.  Nothing here runs on the hot path.
"""
from __future__ import annotations

from collections import OrderedDict
from dataclasses import dataclass
from typing import List

import torch

TEMP_INFERENCE = 0.6        # Generator.forward(inference=True), model.py:75-76


@dataclass
class AvaeBlock:
    idx: int
    kind: str               # 'initial' (ConstantInput), 'up' (nearest x2 -> EqualConv2d -> Blur), 'fused' (FusedUpsample -> Blur)
    cin: int                # input channels of conv1 INCLUDING the concatenated encoder skip
    cout: int
    res: int                # output resolution
    skip: bool              # the encoder's x1 is concatenated in front of this block's conv1


@dataclass
class AvaeSpec:
    output_size: int
    width_div: int
    c512: int
    c256: int
    c128: int
    style_dim: int
    blocks: List[AvaeBlock]
    enc_res: int = 32        # resolution the encoder expects (image_size / kernel_size)
    n_mlp: int = 3


def build_avae_spec(output_size: int, width_div: int = 1) -> AvaeSpec:
    if output_size not in (64, 128, 256):
        raise NotImplementedError(f'Output size {output_size} is not supported')          # model.py:65-66
    a, b, c = 512 // width_div, 256 // width_div, 128 // width_div
    plan = {64: [('initial', a, a), ('up', a, a), ('up', a, a), ('fused', a + b, b), ('fused', b, c)],
            128: [('initial', a, a), ('up', a, a), ('up', a, a), ('fused', a + b, b), ('fused', b, b), ('fused', b, c)],
            256: [('initial', a, a), ('up', a, a), ('up', a, a), ('fused', a + b, b), ('fused', b, b), ('fused', b, b), ('fused', b, c)]}
    blocks, res = [], 4
    for i, (kind, cin, cout) in enumerate(plan[output_size]):
        if kind != 'initial':
            res *= 2
        blocks.append(AvaeBlock(i, kind, cin, cout, res, skip=(cin == a + b)))
    assert res == output_size
    return AvaeSpec(output_size, width_div, a, b, c, a, blocks)


def init_avae_state_dict(output_size: int, seed: int = 0, width_div: int = 1) -> "OrderedDict[str, torch.Tensor]":
    """Seeded weights with the keys and shapes of `StyledGenerator(output_size).state_dict()` (equal-lr parameters are stored as
    `weight_orig` ~ N(0, 1): the sqrt(2 / fan_in) gain is applied at run time, modules.py:13-17).  AdaIN biases keep the
    reference's (1 | 0) initialisation plus a perturbation, noise weights are non-zero so that the noise path is exercised."""
    sp = build_avae_spec(output_size, width_div)
    g = torch.Generator().manual_seed(seed)
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()

    def eq_conv(prefix, cout, cin, k):
        sd[f'{prefix}.conv.bias'] = 0.1 * torch.randn(cout, generator=g)
        sd[f'{prefix}.conv.weight_orig'] = torch.randn(cout, cin, k, k, generator=g)

    def eq_lin(prefix, cout, cin, bias=None):
        sd[f'{prefix}.linear.bias'] = bias if bias is not None else 0.1 * torch.randn(cout, generator=g)
        sd[f'{prefix}.linear.weight_orig'] = torch.randn(cout, cin, generator=g)

    e = sp.c512
    for name, cin, cout in (('conv2', 3, e // 2), ('conv3', e // 2, e), ('conv4', e, 2 * e)):
        eq_conv(f'encoder.{name}.conv1', cout, cin, 3)
        eq_conv(f'encoder.{name}.conv2', cout, cout, 3)
    blur = torch.tensor([[1., 2., 1.], [2., 4., 2.], [1., 2., 1.]]) / 16.0
    for b in sp.blocks:
        p = f'generator.progression.{b.idx}'
        if b.kind == 'initial':
            sd[f'{p}.conv1.input'] = torch.randn(1, b.cin, 4, 4, generator=g)
        elif b.kind == 'up':
            eq_conv(f'{p}.conv1.1', b.cout, b.cin, 3)
            sd[f'{p}.conv1.2.weight'] = blur.view(1, 1, 3, 3).repeat(b.cout, 1, 1, 1)
            sd[f'{p}.conv1.2.weight_flip'] = blur.flip(0, 1).view(1, 1, 3, 3).repeat(b.cout, 1, 1, 1)
        else:
            sd[f'{p}.conv1.0.weight'] = torch.randn(b.cin, b.cout, 3, 3, generator=g)
            sd[f'{p}.conv1.0.bias'] = 0.1 * torch.randn(b.cout, generator=g)
            sd[f'{p}.conv1.1.weight'] = blur.view(1, 1, 3, 3).repeat(b.cout, 1, 1, 1)
            sd[f'{p}.conv1.1.weight_flip'] = blur.flip(0, 1).view(1, 1, 3, 3).repeat(b.cout, 1, 1, 1)
        for j in (1, 2):
            sd[f'{p}.noise{j}.weight_orig'] = 0.5 * torch.randn(1, b.cout, 1, 1, generator=g)
            bias = torch.cat([torch.ones(b.cout), torch.zeros(b.cout)]) + 0.1 * torch.randn(2 * b.cout, generator=g)
            eq_lin(f'{p}.adain{j}.style', 2 * b.cout, sp.style_dim, bias=bias)
            if j == 1:
                eq_conv(f'{p}.conv2', b.cout, b.cout, 3)
    eq_conv('generator.to_rgb', 3, sp.c128, 1)
    eq_lin('style.1', sp.style_dim, sp.c512 * 16)
    for i in range(sp.n_mlp):
        eq_lin(f'style.{3 + 2 * i}', sp.style_dim, sp.style_dim)
    # the reference's key order (state_dict order does not matter for loading; kept close to it for readability)
    return sd
