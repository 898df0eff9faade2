"""
ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/nvae_oracle.py).  CPU restatement of the e4e encoder forward,
functional PyTorch fp32 on a state dict in the reference's layout.  Pinned by tests/golden/e4e_*.npz, produced by the
reference's own `Encoder4Editing` (tests/golden/make_e4e_golden.py).
"""
from __future__ import annotations

import math
from typing import Dict

import torch
import torch.nn.functional as F

from oracle import kinks as K     # relu / leaky_relu / prelu / clamp / max_pool2d: the torch functions unless a test flips near-ties

from gen_adversarial_amd.e4e_spec import E4ESpec

SD = Dict[str, torch.Tensor]


def _bn(sd: SD, p: str, x):
    return F.batch_norm(x, sd[f'{p}.running_mean'], sd[f'{p}.running_var'], sd[f'{p}.weight'], sd[f'{p}.bias'], False, 0.0, 1e-5)


def ir_se_unit(sd: SD, u, x):
    """bottleneck_IR_SE.forward — encoding/helpers.py:97-119; SEModule :57-73."""
    p = u.prefix
    if u.cin == u.depth:
        shortcut = x[:, :, ::u.stride, ::u.stride]                            # MaxPool2d(1, stride)
    else:
        shortcut = _bn(sd, f'{p}.shortcut_layer.1', F.conv2d(x, sd[f'{p}.shortcut_layer.0.weight'], stride=u.stride))
    r = _bn(sd, f'{p}.res_layer.0', x)
    r = F.conv2d(r, sd[f'{p}.res_layer.1.weight'], padding=1)
    r = K.prelu(r, sd[f'{p}.res_layer.2.weight'])
    r = F.conv2d(r, sd[f'{p}.res_layer.3.weight'], stride=u.stride, padding=1)
    r = _bn(sd, f'{p}.res_layer.4', r)
    s = F.adaptive_avg_pool2d(r, 1)
    s = K.relu(F.conv2d(s, sd[f'{p}.res_layer.5.fc1.weight']))
    s = torch.sigmoid(F.conv2d(s, sd[f'{p}.res_layer.5.fc2.weight']))
    return r * s + shortcut


def style_block(sd: SD, spec: E4ESpec, j: int, x):
    """GradualStyleBlock.forward — encoder.py:33-54; EqualLinear (lr_mul=1) generator.py:69-100."""
    for k in range(spec.style_pools[j]):
        x = K.leaky_relu(F.conv2d(x, sd[f'styles.{j}.convs.{2 * k}.weight'], sd[f'styles.{j}.convs.{2 * k}.bias'], stride=2, padding=1))
    x = x.view(-1, spec.style_dim)
    w = sd[f'styles.{j}.linear.weight']
    return F.linear(x, w * (1.0 / math.sqrt(w.shape[1])), sd[f'styles.{j}.linear.bias'])


def upsample_add(x, y):
    """_upsample_add — helpers.py:122-139: bilinear to y's size, align_corners=True, plus y."""
    return F.interpolate(x, size=y.shape[2:], mode='bilinear', align_corners=True) + y


def e4e_encode(sd: SD, spec: E4ESpec, x: torch.Tensor) -> torch.Tensor:
    """Encoder4Editing.forward at ProgressiveStage.Inference — encoder.py:108-140.  x: (B,3,H,W) -> (B, style_count, 512)."""
    x = K.prelu(_bn(sd, 'input_layer.1', F.conv2d(x, sd['input_layer.0.weight'], padding=1)), sd['input_layer.2.weight'])
    feats = {}
    for i, u in enumerate(spec.units):
        x = ir_se_unit(sd, u, x)
        if i in spec.taps:
            feats[spec.taps.index(i)] = x
    c1, c2, c3 = feats[0], feats[1], feats[2]
    p2 = upsample_add(c3, F.conv2d(c2, sd['latlayer1.weight'], sd['latlayer1.bias']))
    p1 = upsample_add(p2, F.conv2d(c1, sd['latlayer2.weight'], sd['latlayer2.bias']))
    src = (c3, p2, p1)
    w0 = style_block(sd, spec, 0, c3)
    cols = [w0] + [w0 + style_block(sd, spec, j, src[spec.style_src[j]]) for j in range(1, spec.style_count)]
    return torch.stack(cols, dim=1)
