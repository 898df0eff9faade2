"""
ORACLE — TEST INFRASTRUCTURE ONLY (nothing under gen_adversarial_amd/ imports this module).  CPU restatement, in plain
functional PyTorch fp32, of the A-VAE competitor defender:

  AVaeDefenseModel.purify / forward   src/defenses/competitors/a_vae/purification_model.py:16-25
  StyledGenerator.forward             src/defenses/competitors/a_vae/model.py:127-141 (inference=True)
  Encoder / Generator                 model.py:9-27, :73-105
  EncodeConvBlock, StyledConvBlock, AdaptiveInstanceNorm, NoiseInjection, FusedUpsample, Blur, EqualLR, PixelNorm
                                      src/defenses/competitors/a_vae/modules.py:384-416, :367-381, :278-296, :299-306, :38-65,
                                      :142-156, :8-35, :98-104
with the random draws passed in: `eps` (the latent sample, model.py:82) and the per-block noise images `noise[i]` [B,1,s,s]
(model.py:131-135 draws them with .cuda(); the reference cannot run its own default on a CPU).  Pinned by tests/golden/avae.npz
(tests/golden/make_avae_golden.py: the reference's own StyledGenerator + AVaeDefenseModel).
"""
from __future__ import annotations

from math import sqrt
from typing import Dict, Sequence

import torch
import torch.nn.functional as F

from gen_adversarial_amd.avae_spec import TEMP_INFERENCE, AvaeSpec

from . import kinks as K

SD = Dict[str, torch.Tensor]


def _eq_conv(sd: SD, p: str, x, stride=1, padding=1):
    w = sd[f'{p}.conv.weight_orig']
    return F.conv2d(x, w * sqrt(2.0 / (w.shape[1] * w.shape[2] * w.shape[3])), sd[f'{p}.conv.bias'], stride=stride, padding=padding)


def _eq_linear(sd: SD, p: str, x):
    w = sd[f'{p}.linear.weight_orig']
    return F.linear(x, w * sqrt(2.0 / w.shape[1]), sd[f'{p}.linear.bias'])


def _lrelu(x):
    return K.leaky_relu(x, 0.2)


def encoder(sd: SD, x):
    """Encoder.forward (model.py:20-27); EncodeConvBlock.forward (modules.py:404-416: the InstanceNorms are computed and dropped)"""
    outs = []
    for name in ('conv2', 'conv3', 'conv4'):
        x = _lrelu(_eq_conv(sd, f'encoder.{name}.conv1', x, 1, 1))
        x = _lrelu(_eq_conv(sd, f'encoder.{name}.conv2', x, 2, 1))
        outs.append(x)
    c = outs[2].shape[1] // 2
    return outs[0], outs[2][:, :c], outs[2][:, c:]


def style_mlp(sd: SD, spec: AvaeSpec, z):
    z = z / torch.sqrt(torch.mean(z ** 2, dim=1, keepdim=True) + 1e-8)
    z = _lrelu(_eq_linear(sd, 'style.1', z))
    for i in range(spec.n_mlp):
        z = _lrelu(_eq_linear(sd, f'style.{3 + 2 * i}', z))
    return z


def _blur(x):
    k = torch.tensor([[1., 2., 1.], [2., 4., 2.], [1., 2., 1.]]) / 16.0
    return F.conv2d(x, k.view(1, 1, 3, 3).repeat(x.shape[1], 1, 1, 1), padding=1, groups=x.shape[1])


def _adain(sd: SD, p: str, x, style):
    gb = _eq_linear(sd, f'{p}.style', style)[:, :, None, None]
    gamma, beta = gb.chunk(2, 1)
    return gamma * F.instance_norm(x, eps=1e-5) + beta


def _noise(sd: SD, p: str, x, noise):
    w = sd[f'{p}.weight_orig']
    return x + w * sqrt(2.0 / w.shape[1]) * noise


def generator(sd: SD, spec: AvaeSpec, x_skip, style, noise: Sequence[torch.Tensor]):
    out = None
    for b in spec.blocks:
        p = f'generator.progression.{b.idx}'
        if b.kind == 'initial':
            out = sd[f'{p}.conv1.input'].repeat(style.shape[0], 1, 1, 1)
        else:
            if b.skip:
                out = torch.cat((out, x_skip), dim=1)
            if b.kind == 'up':
                out = _blur(_eq_conv(sd, f'{p}.conv1.1', F.interpolate(out, scale_factor=2, mode='nearest'), 1, 1))
            else:
                w = sd[f'{p}.conv1.0.weight']
                w = F.pad(w * sqrt(2.0 / (w.shape[0] * w.shape[2] * w.shape[3])), [1, 1, 1, 1])
                w = (w[:, :, 1:, 1:] + w[:, :, :-1, 1:] + w[:, :, 1:, :-1] + w[:, :, :-1, :-1]) / 4
                out = _blur(F.conv_transpose2d(out, w, sd[f'{p}.conv1.0.bias'], stride=2, padding=1))
        out = _adain(sd, f'{p}.adain1', _lrelu(_noise(sd, f'{p}.noise1', out, noise[b.idx])), style)
        out = _eq_conv(sd, f'{p}.conv2', out, 1, 1)
        out = _adain(sd, f'{p}.adain2', _lrelu(_noise(sd, f'{p}.noise2', out, noise[b.idx])), style)
    return _eq_conv(sd, 'generator.to_rgb', out, 1, 0)


def avae_purify(sd: SD, spec: AvaeSpec, x, kernel_size: int, eps, noise: Sequence[torch.Tensor]):
    """AVaeDefenseModel.purify: (B,3,D,D) in [0,1] -> (B,3,D,D) (not clamped, like the reference)"""
    xin = F.avg_pool2d(x * 2 - 1, kernel_size)
    x1, m, v = encoder(sd, xin)
    out = m + eps * (torch.exp(v * 0.5) * TEMP_INFERENCE)
    style = style_mlp(sd, spec, out.reshape(out.shape[0], -1))
    return (generator(sd, spec, x1, style, noise) + 1) / 2
