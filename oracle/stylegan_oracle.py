"""
TEST INFRASTRUCTURE ONLY — CPU restatement (PyTorch, float32/float64) of the StyleGAN2 synthesis layers of the e4e defender.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

Restated from src/mlvgms_autoencoders/StyleGan_E4E/stylegan2/generator.py:
  modulated_conv      ModulatedConv2d.forward without resampling (generator.py:160-207): per-sample weights
                      scale * W * style, optional demodulation, grouped convolution — written as the reference writes it
                      (the HIP engine uses the algebraically equal shared-weight form, see engine_stylegan.py)
  equal_linear        EqualLinear.forward, activation=None (generator.py:85-98)
  styled_conv         StyledConv.forward (generator.py:258-265) with a given noise map
  to_rgb_conv / to_rgb ToRGB.forward without / with the up-sampled skip (generator.py:279-290)
  upfirdn2d           op/upfirdn2d.py:150-187 (the reference's pure-PyTorch statement `upfirdn2d_native`)
  generator_forward   Generator.forward, input_is_latent=True, randomize_noise=False (generator.py:399-470)
  fused_leaky_relu    op/fused_act.py:80-85 + op/fused_bias_act_kernel.cu:18-49 (act 3, grad 0): (x + b > 0 ? x + b :
                      0.2 (x + b)) * sqrt(2)

Pinning: modulated_conv / equal_linear are checked against tests/golden/stylegan_modconv.npz, produced by importing the
reference's ModulatedConv2d (tests/golden/make_stylegan_golden.py).  fused_leaky_relu exists in the reference only as a
CUDA extension (unbuildable here): its restatement follows the .cu arithmetic cited above and is otherwise UNPINNED.
The up-sampling path is pinned in two parts: the transposed convolution against the reference's ModulatedConv2d(upsample=
True) run with its Blur module replaced by the identity (same golden file), the blur / Upsample through upfirdn2d, which is
a restatement of reference PYTHON code that cannot be imported (its module JIT-compiles the CUDA extension on import).
"""
import math

import torch
import torch.nn.functional as F

from oracle import kinks as K     # relu / leaky_relu / prelu / clamp / max_pool2d: the torch functions unless a test flips near-ties


def equal_linear(x, weight, bias, lr_mul: float = 1.0):
    return F.linear(x, weight * ((1.0 / math.sqrt(weight.shape[1])) * lr_mul), bias * lr_mul)


def make_kernel(k=(1, 3, 3, 1)):
    """generator.py:18-27"""
    k = torch.tensor(k, dtype=torch.float32)
    k = k[None, :] * k[:, None]
    return k / k.sum()


def upfirdn2d(x, kernel, up: int = 1, pad=(0, 0)):
    """op/upfirdn2d.py:150-187 (upfirdn2d_native, the reference's own pure-PyTorch statement of the CUDA op) for NCHW input,
    down = 1, non-negative pads: zero insertion, zero padding, correlation with the flipped kernel"""
    n, c, h, w = x.shape
    out = x.reshape(n * c, 1, h, 1, w, 1)
    out = F.pad(out, [0, up - 1, 0, 0, 0, up - 1])
    out = out.reshape(n * c, 1, h * up, w * up)
    out = F.pad(out, [pad[0], pad[1], pad[0], pad[1]])
    out = F.conv2d(out, torch.flip(kernel, [0, 1]).view(1, 1, *kernel.shape).to(x.dtype))
    return out.reshape(n, c, out.shape[-2], out.shape[-1])


def modulated_conv(x, style_latent, weight, mod_weight, mod_bias, demodulate: bool = True, upsample: bool = False,
                   blur: bool = True):
    """x [N,Cin,H,W]; style_latent [N,D]; weight [1,Cout,Cin,k,k]"""
    n, cin, h, w = x.shape
    _, cout, _, k, _ = weight.shape
    style = equal_linear(style_latent, mod_weight, mod_bias).view(n, 1, cin, 1, 1)
    wgt = (1.0 / math.sqrt(cin * k * k)) * weight * style
    if demodulate:
        demod = torch.rsqrt(wgt.pow(2).sum([2, 3, 4]) + 1e-8)
        wgt = wgt * demod.view(n, cout, 1, 1, 1)
    if upsample:                                   # generator.py:178-189 + Blur (generator.py:45-65), pads from :133-139
        wt = wgt.transpose(1, 2).reshape(n * cin, cout, k, k)
        out = F.conv_transpose2d(x.reshape(1, n * cin, h, w), wt, padding=0, stride=2, groups=n)
        out = out.view(n, cout, out.shape[-2], out.shape[-1])
        if not blur:                               # golden check of the transposed conv alone
            return out
        p = (4 - 2) - (k - 1)
        return upfirdn2d(out, make_kernel() * 4, pad=((p + 1) // 2 + 1, p // 2 + 1))
    out = F.conv2d(x.reshape(1, n * cin, h, w), wgt.view(n * cout, cin, k, k), padding=k // 2, groups=n)
    return out.view(n, cout, h, w)


def fused_leaky_relu(x, bias, negative_slope: float = 0.2, scale: float = 2 ** 0.5):
    return K.leaky_relu(x + bias.view(1, -1, 1, 1), negative_slope) * scale


def styled_conv(sd, p, x, style_latent, noise, upsample: bool = False):
    """noise: [H,W] fixed buffer (Generator.noises, randomize_noise=False) or None"""
    out = modulated_conv(x, style_latent, sd[f'{p}.conv.weight'], sd[f'{p}.conv.modulation.weight'],
                         sd[f'{p}.conv.modulation.bias'], True, upsample)
    if noise is not None:
        out = out + sd[f'{p}.noise.weight'] * noise.view(1, 1, *noise.shape[-2:])
    return fused_leaky_relu(out, sd[f'{p}.activate.bias'])


def to_rgb_conv(sd, p, x, style_latent):
    out = modulated_conv(x, style_latent, sd[f'{p}.conv.weight'], sd[f'{p}.conv.modulation.weight'],
                         sd[f'{p}.conv.modulation.bias'], False)
    return out + sd[f'{p}.bias']


def to_rgb(sd, p, x, style_latent, skip=None):
    """ToRGB.forward (generator.py:279-290); Upsample (generator.py:29-46): kernel * 4, up 2, pad (2, 1)"""
    out = to_rgb_conv(sd, p, x, style_latent)
    if skip is not None:
        out = out + upfirdn2d(skip, make_kernel() * 4, up=2, pad=(2, 1))
    return out


def generator_forward(sd, spec, latent):
    """Generator.forward for styles=[latent] with input_is_latent=True, randomize_noise=False (generator.py:399-470):
    latent [N, n_latent, D] -> image [N,3,size,size]"""
    n = latent.shape[0]
    noise = [sd[f'noises.noise_{i}'][0, 0] for i in range(1 + len(spec.convs))]
    out = sd['input.input'].repeat(n, 1, 1, 1)
    out = styled_conv(sd, 'conv1', out, latent[:, 0], noise[0])
    skip = to_rgb(sd, 'to_rgb1', out, latent[:, 1])
    i = 1
    for k in range(len(spec.to_rgbs)):
        out = styled_conv(sd, f'convs.{2 * k}', out, latent[:, i], noise[1 + 2 * k], upsample=True)
        out = styled_conv(sd, f'convs.{2 * k + 1}', out, latent[:, i + 1], noise[2 + 2 * k])
        skip = to_rgb(sd, f'to_rgbs.{k}', out, latent[:, i + 2], skip)
        i += 2
    return skip


def mapping_network(sd, z, n_mlp: int = 8, lr_mul: float = 0.01):
    """Generator.style (generator.py:306-317): PixelNorm (generator.py:10-15) + n_mlp EqualLinear(lr_mul, 'fused_lrelu')
    (generator.py:85-92: linear with weight * (lr_mul / sqrt(in)), then fused_leaky_relu with bias * lr_mul); z [..., D]"""
    h = z * torch.rsqrt(torch.mean(z ** 2, dim=-1, keepdim=True) + 1e-8)
    for k in range(1, n_mlp + 1):
        w = sd[f'style.{k}.weight']
        h = F.linear(h, w * ((1.0 / math.sqrt(w.shape[1])) * lr_mul))
        h = K.leaky_relu(h + sd[f'style.{k}.bias'] * lr_mul, 0.2) * 2 ** 0.5
    return h
