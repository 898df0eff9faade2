"""
Style-Transformer encoder (`GradualStyleEncoder(50, 'ir_se', opts)`) restated structurally — SURVEY.md §8 row a18.

Reference: src/mlvgms_autoencoders/StyleGan_Trans/models/encoders/style_transformer_encoders.py:10-85 and
models/transformer.py:17-100:
  trunk   the IR-SE50 input layer + body + FPN of the e4e encoder (same modules, encoders/helpers.py; taps 6 / 20 / 23;
          latlayer1 256 -> 512 on c2, latlayer2 128 -> 512 on c1, bilinear align_corners=True upsample-add)        -> c3, p2, p1
  queries `style(z)` of the decoder's mapping network on the LEARNED z (1 x 16 x 512)  (style_transformer.py:61-66; models.py:311-316)
  3 DETR post-norm decoder layers (d_model 512, 4 heads, FFN 1024, ReLU, dropout inactive, no positional encodings):
          coarse: memory = c3 tokens, medium: p2 tokens, fine: p1 tokens (HW x B x C, :73-80)
  output  codes (B, 16, 512); the defender adds latent_avg and mixes every index with a freshly mapped style.
State-dict keys: the e4e trunk's (input_layer.*, body.N.*, latlayer1/2.*) + transformerlayer_{coarse,medium,fine}.
{self_attn,multihead_attn}.{in_proj_weight,in_proj_bias,out_proj.weight,out_proj.bias}, .linear1/2.{weight,bias}, .norm1/2/3.{weight,bias}, + z.
`width_div` / `units` shrink the network for tests (d_model = 512 / width_div, still 4 heads).
"""
from __future__ import annotations

from collections import OrderedDict
from dataclasses import dataclass
from typing import Tuple

import numpy as np
import torch

from .e4e_spec import E4ESpec, IR50_UNITS, build_e4e_spec, init_e4e_state_dict
from .nvae_spec import _Rng

LAYERS = ('transformerlayer_coarse', 'transformerlayer_medium', 'transformerlayer_fine')      # memory: c3, p2, p1


@dataclass
class TransSpec:
    trunk: E4ESpec
    d_model: int
    nhead: int
    dff: int
    n_query: int          # 16 = the style count of a 512-px StyleGAN2 (style_transformer.py:22)
    eps: float = 1e-5


def build_trans_spec(width_div: int = 1, units: Tuple[int, ...] = IR50_UNITS) -> TransSpec:
    trunk = build_e4e_spec(512, width_div, units)
    return TransSpec(trunk, trunk.style_dim, 4, 1024 // width_div, 16)


def init_trans_state_dict(width_div: int = 1, seed: int = 0, units: Tuple[int, ...] = IR50_UNITS):
    """seeded random weights with the reference module's key names and shapes (load_state_dict(strict=True) compatible)"""
    spec = build_trans_spec(width_div, units)
    e4e = init_e4e_state_dict(512, width_div, seed, units)
    sd = OrderedDict((k, v) for k, v in e4e.items() if not k.startswith('styles.'))
    rng = _Rng(seed + 77)
    d, f = spec.d_model, spec.dff
    for name in LAYERS:
        for att in ('self_attn', 'multihead_attn'):
            sd[f'{name}.{att}.in_proj_weight'] = rng.normal((3 * d, d), std=np.sqrt(1.0 / d))
            sd[f'{name}.{att}.in_proj_bias'] = rng.normal((3 * d,), std=0.05)
            sd[f'{name}.{att}.out_proj.weight'] = rng.normal((d, d), std=np.sqrt(1.0 / d))
            sd[f'{name}.{att}.out_proj.bias'] = rng.normal((d,), std=0.05)
        sd[f'{name}.linear1.weight'] = rng.normal((f, d), std=np.sqrt(2.0 / d))
        sd[f'{name}.linear1.bias'] = rng.normal((f,), std=0.05)
        sd[f'{name}.linear2.weight'] = rng.normal((d, f), std=np.sqrt(1.0 / f))
        sd[f'{name}.linear2.bias'] = rng.normal((d,), std=0.05)
        for k in (1, 2, 3):
            sd[f'{name}.norm{k}.weight'] = rng.uniform((d,), 0.8, 1.2)
            sd[f'{name}.norm{k}.bias'] = rng.normal((d,), std=0.1)
    sd['z'] = rng.normal((1, spec.n_query, d), std=1.0)
    return sd
