"""
e4e encoder (`Encoder4Editing(50, 'ir_se', opts)`) restated structurally — SURVEY.md §8 row a14.

Reference: src/mlvgms_autoencoders/StyleGan_E4E/encoding/encoder.py:33-140 (GradualStyleBlock, Encoder4Editing) and
encoding/helpers.py:24-119 (get_blocks, SEModule, bottleneck_IR_SE), EqualLinear at stylegan2/generator.py:69-100:

  input_layer  Conv2d(3, 64, 3, 1, 1, bias=False) -> BatchNorm2d -> PReLU(64)
  body         24 bottleneck_IR_SE units (64,64,s2)(64,64,1)x2 | (64,128,s2)(128,128,1)x3 | (128,256,s2)(256,256,1)x13 |
               (256,512,s2)(512,512,1)x2;  unit(x) = shortcut(x) + SE(BN(conv3x3_s(PReLU(conv3x3(BN(x))))))
               shortcut = MaxPool2d(1, s) (= x[::s, ::s]) when in == depth, else Conv2d(in, depth, 1, s, bias=False) + BN
               taps: c1 = body[6] (128 ch), c2 = body[20] (256 ch), c3 = body[23] (512 ch)
  FPN          p2 = bilinear_up(c3 -> c2's size, align_corners=True) + latlayer1(c2);  p1 = up(p2) + latlayer2(c1)
  styles       style_count = 2*log2(stylegan_size) - 2 GradualStyleBlocks: i < 3 on c3 (spatial 16), i < 7 on p2 (32), else on
               p1 (64); each = log2(spatial) x [Conv2d(512, 512, 3, 2, 1) + LeakyReLU(0.01)] -> view(-1, 512) -> EqualLinear
  output       w[:, 0] = styles[0](c3);  w[:, i] = w[:, 0] + styles[i](features_i)        -> (B, style_count, 512)

`width_div` shrinks every channel count and `units` the stage depths (tests only); the real model is the default.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from dataclasses import dataclass
from typing import List, Tuple

import numpy as np
import torch

from .nvae_spec import _Rng

IR50_UNITS = (3, 4, 14, 3)


@dataclass
class IRUnit:
    prefix: str        # 'body.7'
    cin: int
    depth: int
    stride: int


@dataclass
class E4ESpec:
    units: List[IRUnit]
    taps: Tuple[int, int, int]            # body indices of c1, c2, c3
    base: int                              # input_layer channels (64)
    style_dim: int                         # 512
    style_count: int
    style_pools: List[int]                 # stride-2 convs per style block (log2 of its nominal spatial size)
    style_src: List[int]                   # 0: c3, 1: p2, 2: p1
    reduction: int = 16


def build_e4e_spec(stylegan_size: int = 1024, width_div: int = 1, units: Tuple[int, ...] = IR50_UNITS) -> E4ESpec:
    base = 64 // width_div
    depths = [64 // width_div, 128 // width_div, 256 // width_div, 512 // width_div]
    out: List[IRUnit] = []
    cin = base
    ends = []
    for nb, depth in zip(units, depths):
        for b in range(nb):
            out.append(IRUnit(f'body.{len(out)}', cin, depth, 2 if b == 0 else 1))
            cin = depth
        ends.append(len(out) - 1)
    taps = (ends[1], ends[2], ends[3])      # 6, 20, 23 for IR-50 (encoder.py:112-117)
    count = 2 * int(math.log(stylegan_size, 2)) - 2
    pools, src = [], []
    for i in range(count):
        spatial, s = (16, 0) if i < 3 else (32, 1) if i < 7 else (64, 2)
        pools.append(int(np.log2(spatial)))
        src.append(s)
    return E4ESpec(out, taps, base, depths[3], count, pools, src)


def _bn(sd, rng, prefix, c):
    sd[f'{prefix}.weight'] = rng.uniform((c,), 0.8, 1.2)
    sd[f'{prefix}.bias'] = rng.normal((c,), std=0.1)
    sd[f'{prefix}.running_mean'] = rng.normal((c,), std=0.1)
    sd[f'{prefix}.running_var'] = rng.uniform((c,), 0.5, 1.5)
    sd[f'{prefix}.num_batches_tracked'] = torch.tensor(0, dtype=torch.long)


def init_e4e_state_dict(stylegan_size: int = 1024, width_div: int = 1, seed: int = 0, units: Tuple[int, ...] = IR50_UNITS):
    """seeded random weights with the reference module's key names and shapes (load_state_dict(strict=True) compatible)"""
    spec = build_e4e_spec(stylegan_size, width_div, units)
    rng = _Rng(seed)
    sd = OrderedDict()
    sd['input_layer.0.weight'] = rng.normal((spec.base, 3, 3, 3), std=np.sqrt(2.0 / 27))
    _bn(sd, rng, 'input_layer.1', spec.base)
    sd['input_layer.2.weight'] = rng.uniform((spec.base,), -0.1, 0.4)
    for u in spec.units:
        p = u.prefix
        if u.cin != u.depth:
            sd[f'{p}.shortcut_layer.0.weight'] = rng.normal((u.depth, u.cin, 1, 1), std=np.sqrt(1.0 / u.cin))
            _bn(sd, rng, f'{p}.shortcut_layer.1', u.depth)
        _bn(sd, rng, f'{p}.res_layer.0', u.cin)
        sd[f'{p}.res_layer.1.weight'] = rng.normal((u.depth, u.cin, 3, 3), std=np.sqrt(2.0 / (u.cin * 9)))
        sd[f'{p}.res_layer.2.weight'] = rng.uniform((u.depth,), -0.1, 0.4)
        sd[f'{p}.res_layer.3.weight'] = rng.normal((u.depth, u.depth, 3, 3), std=0.5 * np.sqrt(2.0 / (u.depth * 9)))
        _bn(sd, rng, f'{p}.res_layer.4', u.depth)
        hid = max(1, u.depth // spec.reduction)
        sd[f'{p}.res_layer.5.fc1.weight'] = rng.normal((hid, u.depth, 1, 1), std=np.sqrt(2.0 / u.depth))
        sd[f'{p}.res_layer.5.fc2.weight'] = rng.normal((u.depth, hid, 1, 1), std=np.sqrt(1.0 / hid))
    d = spec.style_dim
    for j, pools in enumerate(spec.style_pools):
        for k in range(pools):
            sd[f'styles.{j}.convs.{2 * k}.weight'] = rng.normal((d, d, 3, 3), std=np.sqrt(2.0 / (d * 9)))
            sd[f'styles.{j}.convs.{2 * k}.bias'] = rng.normal((d,), std=0.05)
        sd[f'styles.{j}.linear.weight'] = rng.normal((d, d), std=1.0)
        sd[f'styles.{j}.linear.bias'] = rng.normal((d,), std=0.05)
    c2, c1 = spec.units[spec.taps[1]].depth, spec.units[spec.taps[0]].depth
    sd['latlayer1.weight'] = rng.normal((d, c2, 1, 1), std=np.sqrt(1.0 / c2))
    sd['latlayer1.bias'] = rng.normal((d,), std=0.05)
    sd['latlayer2.weight'] = rng.normal((d, c1, 1, 1), std=np.sqrt(1.0 / c1))
    sd['latlayer2.bias'] = rng.normal((d,), std=0.05)
    return sd
