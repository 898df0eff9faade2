"""
VGG-11-BN + projector head as the reference's `Vgg` wrapper builds it, restated structurally.

Reference: src/classifier/model.py:31-49 (torchvision `vgg11_bn` whose `classifier` is replaced by
Linear(d, d, bias=False) -> BatchNorm1d(d) -> ReLU -> Linear(d, n_classes), d = 512*7*7 = 25088) and
src/defenses/loading_utils.py:19-25 (state dict under ckpt['state_dict'], keys `model.features.N.*`,
`model.classifier.{0,1,3}.*`).  torchvision itself is a third-party dependency that is absent from the
reference tree and from this image (environment.yml:10, unpinned): the topology below restates its
published configuration 'A' with batch norm: 64 M 128 M 256 256 M 512 512 M 512 512 M, 3x3 pad 1
convolutions with bias, BatchNorm2d(eps=1e-5), ReLU, MaxPool(2,2), AdaptiveAvgPool2d((7,7)).

`width_div` shrinks every channel count (tests only); the real model is width_div=1.
"""
from __future__ import annotations

from collections import OrderedDict
from dataclasses import dataclass
from typing import List, Tuple, Union

import numpy as np
import torch

from .nvae_spec import _Rng

VGG11_CFG = [64, 'M', 128, 'M', 256, 256, 'M', 512, 512, 'M', 512, 512, 'M']


@dataclass
class VggSpec:
    # feature program: ('conv', features_idx_of_conv, cin, cout) | ('pool',)
    program: List[Tuple]
    feat_channels: int
    head_dim: int          # feat_channels * 49
    n_classes: int


def build_vgg_spec(n_classes: int = 100, width_div: int = 1) -> VggSpec:
    program, idx, cin = [], 0, 3
    for v in VGG11_CFG:
        if v == 'M':
            program.append(('pool',))
            idx += 1
        else:
            cout = v // width_div
            program.append(('conv', idx, cin, cout))
            cin = cout
            idx += 3                      # conv, bn, relu
    return VggSpec(program, cin, cin * 49, n_classes)


def init_vgg_state_dict(n_classes: int = 100, width_div: int = 1, seed: int = 0):
    spec = build_vgg_spec(n_classes, width_div)
    rng = _Rng(seed)
    sd = OrderedDict()
    for op in spec.program:
        if op[0] != 'conv':
            continue
        _, i, cin, cout = op
        sd[f'model.features.{i}.weight'] = rng.normal((cout, cin, 3, 3), std=np.sqrt(2.0 / (cin * 9)))
        sd[f'model.features.{i}.bias'] = rng.normal((cout,), std=0.05)
        b = f'model.features.{i + 1}'
        sd[f'{b}.weight'] = rng.uniform((cout,), 0.8, 1.2)
        sd[f'{b}.bias'] = rng.normal((cout,), std=0.1)
        sd[f'{b}.running_mean'] = rng.normal((cout,), std=0.1)
        sd[f'{b}.running_var'] = rng.uniform((cout,), 0.5, 1.5)
        sd[f'{b}.num_batches_tracked'] = torch.tensor(0, dtype=torch.long)
    d = spec.head_dim
    sd['model.classifier.0.weight'] = rng.normal((d, d), std=np.sqrt(2.0 / d))
    sd['model.classifier.1.weight'] = rng.uniform((d,), 0.8, 1.2)
    sd['model.classifier.1.bias'] = rng.normal((d,), std=0.1)
    sd['model.classifier.1.running_mean'] = rng.normal((d,), std=0.1)
    sd['model.classifier.1.running_var'] = rng.uniform((d,), 0.5, 1.5)
    sd['model.classifier.1.num_batches_tracked'] = torch.tensor(0, dtype=torch.long)
    sd['model.classifier.3.weight'] = rng.normal((n_classes, d), std=np.sqrt(1.0 / d))
    sd['model.classifier.3.bias'] = rng.normal((n_classes,), std=0.05)
    return sd


def adaptive_avgpool_matrix(n_in: int, n_out: int = 7) -> torch.Tensor:
    """
    1-D AdaptiveAvgPool as an (n_out, n_in) matrix: output o averages inputs
    floor(o*n_in/n_out) .. ceil((o+1)*n_in/n_out)-1 (PyTorch's published adaptive-pool window rule).
    """
    m = torch.zeros(n_out, n_in, dtype=torch.float64)
    for o in range(n_out):
        lo = (o * n_in) // n_out
        hi = -((-(o + 1) * n_in) // n_out)
        m[o, lo:hi] = 1.0 / (hi - lo)
    return m
