"""
ResNet-50 + projector head as the reference's `ResNet` wrapper builds it, restated structurally.

Reference: src/classifier/model.py:10-28 (torchvision `resnet50` whose `fc` is replaced by
Linear(2048, 2048, bias=False) -> BatchNorm1d(2048) -> ReLU -> Linear(2048, n_classes)) and
src/defenses/loading_utils.py:10-16 (state dict under ckpt['state_dict'], keys `model.conv1.weight`, `model.bn1.*`,
`model.layerL.B.{conv1,bn1,conv2,bn2,conv3,bn3}.*`, `model.layerL.0.downsample.{0,1}.*`, `model.fc.{0,1,3}.*`).
torchvision itself is a third-party dependency that is absent from the reference tree and from this image
(environment.yml:10, unpinned): the topology below restates its published ResNet-50 ("v1.5": the stride of a
down-sampling Bottleneck sits on its 3x3 convolution): 7x7/2 conv 64 + BN + ReLU, 3x3/2 max pool (pad 1),
stages of [3, 4, 6, 3] Bottlenecks with widths 64/128/256/512 and expansion 4 (1x1 -> 3x3 -> 1x1, BatchNorm2d(eps=1e-5)
after each, ReLU after the first two and after the residual sum; a 1x1 strided conv + BN on the shortcut where shapes
change), AdaptiveAvgPool2d((1,1)), all convolutions without bias.

ResNeXt-50 32x4d (src/classifier/model.py:52-70, `CarsTypeClassifier`) is the same network with `groups=32,
width_per_group=4`: the Bottleneck width becomes planes * 4/64 * 32 (128/256/512/1024) and its 3x3 convolution is
grouped (weight [width, width/32, 3, 3]).

`width_div` shrinks every channel count and `blocks` the stage depths (tests only); the real model is the default.
"""
from __future__ import annotations

from collections import OrderedDict
from dataclasses import dataclass
from typing import List, Tuple

import numpy as np
import torch

from .nvae_spec import _Rng

RESNET50_BLOCKS = (3, 4, 6, 3)


@dataclass
class BottleneckSpec:
    prefix: str            # 'model.layer1.0'
    cin: int
    width: int
    cout: int
    stride: int
    downsample: bool
    groups: int = 1


@dataclass
class ResNetSpec:
    stem_channels: int
    blocks: List[BottleneckSpec]
    feat_channels: int     # 2048 / width_div
    n_classes: int


def build_resnet_spec(n_classes: int = 2, width_div: int = 1, blocks: Tuple[int, ...] = RESNET50_BLOCKS, groups: int = 1,
                      width_per_group: int = 64) -> ResNetSpec:
    stem = 64 // width_div
    out: List[BottleneckSpec] = []
    cin = stem
    for li, (nb, planes) in enumerate(zip(blocks, (64, 128, 256, 512))):
        width = int(planes * (width_per_group / 64.0)) * groups // width_div      # torchvision Bottleneck.__init__
        cout = planes * 4 // width_div
        for b in range(nb):
            stride = 2 if (b == 0 and li > 0) else 1
            out.append(BottleneckSpec(f'model.layer{li + 1}.{b}', cin, width, cout, stride, b == 0 and (stride != 1 or cin != cout), groups))
            cin = cout
    return ResNetSpec(stem, out, cin, n_classes)


def _bn(sd, rng, prefix, c):
    sd[f'{prefix}.weight'] = rng.uniform((c,), 0.8, 1.2)
    sd[f'{prefix}.bias'] = rng.normal((c,), std=0.1)
    sd[f'{prefix}.running_mean'] = rng.normal((c,), std=0.1)
    sd[f'{prefix}.running_var'] = rng.uniform((c,), 0.5, 1.5)
    sd[f'{prefix}.num_batches_tracked'] = torch.tensor(0, dtype=torch.long)


def init_resnet_state_dict(n_classes: int = 2, width_div: int = 1, seed: int = 0, blocks: Tuple[int, ...] = RESNET50_BLOCKS,
                           groups: int = 1, width_per_group: int = 64):
    """seeded random weights with torchvision's key names and shapes (He-style scales so that activations stay O(1))"""
    spec = build_resnet_spec(n_classes, width_div, blocks, groups, width_per_group)
    rng = _Rng(seed)
    sd = OrderedDict()
    sd['model.conv1.weight'] = rng.normal((spec.stem_channels, 3, 7, 7), std=np.sqrt(2.0 / (3 * 49)))
    _bn(sd, rng, 'model.bn1', spec.stem_channels)
    for b in spec.blocks:
        sd[f'{b.prefix}.conv1.weight'] = rng.normal((b.width, b.cin, 1, 1), std=np.sqrt(2.0 / b.cin))
        _bn(sd, rng, f'{b.prefix}.bn1', b.width)
        cg = b.width // b.groups
        sd[f'{b.prefix}.conv2.weight'] = rng.normal((b.width, cg, 3, 3), std=np.sqrt(2.0 / (cg * 9)))
        _bn(sd, rng, f'{b.prefix}.bn2', b.width)
        sd[f'{b.prefix}.conv3.weight'] = rng.normal((b.cout, b.width, 1, 1), std=0.5 * np.sqrt(2.0 / b.width))
        _bn(sd, rng, f'{b.prefix}.bn3', b.cout)
        if b.downsample:
            sd[f'{b.prefix}.downsample.0.weight'] = rng.normal((b.cout, b.cin, 1, 1), std=np.sqrt(1.0 / b.cin))
            _bn(sd, rng, f'{b.prefix}.downsample.1', b.cout)
    d = spec.feat_channels
    sd['model.fc.0.weight'] = rng.normal((d, d), std=np.sqrt(2.0 / d))
    _bn(sd, rng, 'model.fc.1', d)
    sd['model.fc.3.weight'] = rng.normal((n_classes, d), std=np.sqrt(1.0 / d))
    sd['model.fc.3.bias'] = rng.normal((n_classes,), std=0.05)
    return sd
