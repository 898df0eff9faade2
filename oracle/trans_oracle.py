"""
ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/nvae_oracle.py for the rules).  CPU restatement of the Style-Transformer
encoder and defender, functional PyTorch fp32 on state dicts in the reference's layout:
  src/mlvgms_autoencoders/StyleGan_Trans/models/transformer.py:17-100        TransformerDecoderLayer.forward_post
  .../models/encoders/style_transformer_encoders.py:58-85                    GradualStyleEncoder.forward
  src/defenses/ours/models.py:299-353                                        TransStyleGanDefenseModel.purify
torch.nn.MultiheadAttention (third party: torch) is restated from its documented arithmetic: q/k/v = in_proj thirds, heads
split along the channels, softmax(q k^T / sqrt(d_head)) v, out_proj.  kornia.geometry.resize (third party, absent from the
image, unpinned in environment.yml:17) is restated as F.interpolate(bilinear, align_corners=False, no antialias) — its default.
Pinned by tests/golden/trans_*.npz (tests/golden/make_trans_golden.py: the reference's own TransformerDecoderLayer,
GradualStyleEncoder and TransStyleGanDefenseModel.purify, imported with the stale package name aliased).
"""
from __future__ import annotations

import math
from typing import Dict

import torch
import torch.nn.functional as F

from gen_adversarial_amd.trans_spec import LAYERS, TransSpec
from oracle import e4e_oracle as E
from oracle import kinks as K     # relu: the torch function unless a test replays / flips decisions

SD = Dict[str, torch.Tensor]


def mha(sd: SD, p: str, query, key, value, nhead: int):
    """nn.MultiheadAttention.forward (batch_first layout here: [B, T, C]); returns the attention output only"""
    d = query.shape[-1]
    w, b = sd[f'{p}.in_proj_weight'], sd[f'{p}.in_proj_bias']
    q = F.linear(query, w[:d], b[:d])
    k = F.linear(key, w[d:2 * d], b[d:2 * d])
    v = F.linear(value, w[2 * d:], b[2 * d:])
    B, Tq, _ = q.shape
    dh = d // nhead

    def heads(t):
        return t.view(B, t.shape[1], nhead, dh).transpose(1, 2)              # [B, h, T, dh]
    att = torch.softmax(heads(q) @ heads(k).transpose(-1, -2) / math.sqrt(dh), dim=-1)
    out = (att @ heads(v)).transpose(1, 2).reshape(B, Tq, d)
    return F.linear(out, sd[f'{p}.out_proj.weight'], sd[f'{p}.out_proj.bias'])


def decoder_layer(sd: SD, p: str, tgt, memory, nhead: int, eps: float = 1e-5):
    """TransformerDecoderLayer.forward_post with pos = query_pos = None, eval mode (transformer.py:42-64)"""
    def ln(x, k):
        return F.layer_norm(x, (x.shape[-1],), sd[f'{p}.norm{k}.weight'], sd[f'{p}.norm{k}.bias'], eps)
    tgt = ln(tgt + mha(sd, f'{p}.self_attn', tgt, tgt, tgt, nhead), 1)
    tgt = ln(tgt + mha(sd, f'{p}.multihead_attn', tgt, memory, memory, nhead), 2)
    ff = F.linear(K.relu(F.linear(tgt, sd[f'{p}.linear1.weight'], sd[f'{p}.linear1.bias'])), sd[f'{p}.linear2.weight'], sd[f'{p}.linear2.bias'])
    return ln(tgt + ff, 3)


def trunk(sd: SD, spec: TransSpec, x):
    """input layer, body and FPN (style_transformer_encoders.py:59-73): returns c3, p2, p1"""
    t = spec.trunk
    x = K.prelu(E._bn(sd, 'input_layer.1', F.conv2d(x, sd['input_layer.0.weight'], padding=1)), sd['input_layer.2.weight'])
    feats = {}
    for i, u in enumerate(t.units):
        x = E.ir_se_unit(sd, u, x)
        if i in t.taps:
            feats[t.taps.index(i)] = x
    c1, c2, c3 = feats[0], feats[1], feats[2]
    p2 = E.upsample_add(c3, F.conv2d(c2, sd['latlayer1.weight'], sd['latlayer1.bias']))
    p1 = E.upsample_add(p2, F.conv2d(c1, sd['latlayer2.weight'], sd['latlayer2.bias']))
    return c3, p2, p1


def encode(sd: SD, spec: TransSpec, x, query):
    """GradualStyleEncoder.forward(x, query): x (B,3,H,W) normalised, query (B, 16, C) -> codes (B, 16, C)"""
    c3, p2, p1 = trunk(sd, spec, x)
    q = query
    for name, mem in zip(LAYERS, (c3, p2, p1)):
        q = decoder_layer(sd, name, q, mem.flatten(2).transpose(1, 2), spec.nhead, spec.eps)
    return q


def queries(sd: SD, gsd: SD, rows: int):
    """style(z) of the decoder's mapping network on the learned z (models.py:311-316)"""
    from oracle.stylegan_oracle import mapping_network
    z = sd['z']
    return mapping_network(gsd, z[0]).unsqueeze(0).expand(rows, -1, -1)


def resize_bilinear(x, size: int):
    """kornia.geometry.resize(x, size) for square images: bilinear, align_corners=False, antialias=False"""
    return F.interpolate(x, size=(size, size), mode='bilinear', align_corners=False)


def trans_purify(sd: SD, spec: TransSpec, gsd: SD, gspec, latent_avg, x01, alphas, z, out_size: int = 128, mid: int = 256,
                 crop: int = 32, pool_to: int = 256):
    """MLVGMDefenseModel.__call__ around TransStyleGanDefenseModel.purify up to the de-normalised purified image
    (abstract_models.py:176-185; models.py:299-353).  x01 (B,3,out_size,out_size) in [0,1]; z (B,16,C): the N(0, 0.8) draw of
    models.py:331 (already scaled).  mid / crop / pool_to are 256 / 32 / 256 in the reference."""
    from oracle import stylegan_oracle as S
    x = (x01 - 0.5) / 0.5
    x = resize_bilinear(x, mid)[:, :, crop:-crop]
    codes = encode(sd, spec, x, queries(sd, gsd, x.shape[0]))
    if latent_avg is not None:
        codes = codes + latent_avg.unsqueeze(0)
    styles = S.mapping_network(gsd, z)
    a = torch.tensor(list(alphas), dtype=codes.dtype).view(1, -1, 1)
    codes = (1 - a) * codes + a * styles
    img = S.generator_forward(gsd, gspec, codes)
    img = F.adaptive_avg_pool2d(img, (pool_to, pool_to))
    band = torch.ones(1, 1, pool_to, 1)
    band[:, :, :crop] = 0
    band[:, :, -crop:] = 0
    img = img * band + (band - 1.0)                       # images[:, :, :32] = -1; images[:, :, -32:] = -1
    img = resize_bilinear(img, out_size)
    return img * 0.5 + 0.5
