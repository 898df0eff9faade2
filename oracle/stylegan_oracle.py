"""
TEST INFRASTRUCTURE ONLY — CPU restatement (PyTorch, float32/float64) of the StyleGAN2 synthesis layers of the e4e defender.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

Restated from src/mlvgms_autoencoders/StyleGan_E4E/stylegan2/generator.py:
  modulated_conv      ModulatedConv2d.forward without resampling (generator.py:160-207): per-sample weights
                      scale * W * style, optional demodulation, grouped convolution — written as the reference writes it
                      (the HIP engine uses the algebraically equal shared-weight form, see engine_stylegan.py)
  equal_linear        EqualLinear.forward, activation=None (generator.py:85-98)
  styled_conv         StyledConv.forward (generator.py:258-265) with a given noise map
  to_rgb_conv         ToRGB.forward without skip (generator.py:282-283)
  fused_leaky_relu    op/fused_act.py:80-85 + op/fused_bias_act_kernel.cu:18-49 (act 3, grad 0): (x + b > 0 ? x + b :
                      0.2 (x + b)) * sqrt(2)

Pinning: modulated_conv / equal_linear are checked against tests/golden/stylegan_modconv.npz, produced by importing the
reference's ModulatedConv2d (tests/golden/make_stylegan_golden.py).  fused_leaky_relu exists in the reference only as a
CUDA extension (unbuildable here): its restatement follows the .cu arithmetic cited above and is otherwise UNPINNED.
"""
import math

import torch
import torch.nn.functional as F


def equal_linear(x, weight, bias, lr_mul: float = 1.0):
    return F.linear(x, weight * ((1.0 / math.sqrt(weight.shape[1])) * lr_mul), bias * lr_mul)


def modulated_conv(x, style_latent, weight, mod_weight, mod_bias, demodulate: bool = True):
    """x [N,Cin,H,W]; style_latent [N,D]; weight [1,Cout,Cin,k,k]"""
    n, cin, h, w = x.shape
    _, cout, _, k, _ = weight.shape
    style = equal_linear(style_latent, mod_weight, mod_bias).view(n, 1, cin, 1, 1)
    wgt = (1.0 / math.sqrt(cin * k * k)) * weight * style
    if demodulate:
        demod = torch.rsqrt(wgt.pow(2).sum([2, 3, 4]) + 1e-8)
        wgt = wgt * demod.view(n, cout, 1, 1, 1)
    out = F.conv2d(x.reshape(1, n * cin, h, w), wgt.view(n * cout, cin, k, k), padding=k // 2, groups=n)
    return out.view(n, cout, h, w)


def fused_leaky_relu(x, bias, negative_slope: float = 0.2, scale: float = 2 ** 0.5):
    return F.leaky_relu(x + bias.view(1, -1, 1, 1), negative_slope) * scale


def styled_conv(sd, p, x, style_latent, noise):
    """noise: [H,W] fixed buffer (Generator.noises, randomize_noise=False) or None"""
    out = modulated_conv(x, style_latent, sd[f'{p}.conv.weight'], sd[f'{p}.conv.modulation.weight'],
                         sd[f'{p}.conv.modulation.bias'], True)
    if noise is not None:
        out = out + sd[f'{p}.noise.weight'] * noise.view(1, 1, *noise.shape[-2:])
    return fused_leaky_relu(out, sd[f'{p}.activate.bias'])


def to_rgb_conv(sd, p, x, style_latent):
    out = modulated_conv(x, style_latent, sd[f'{p}.conv.weight'], sd[f'{p}.conv.modulation.weight'],
                         sd[f'{p}.conv.modulation.bias'], False)
    return out + sd[f'{p}.bias']
