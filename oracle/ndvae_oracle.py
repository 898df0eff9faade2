"""
ORACLE — TEST INFRASTRUCTURE ONLY.  Nothing under gen_adversarial_amd/ imports this module; only tests/, bench.py's cpu_baseline
leg and __graft_entry__.smoke() may.  CPU restatement, in plain functional PyTorch fp32, of the ND-VAE competitor defender:

  NDVaeDefenseModel.purify / forward     src/defenses/competitors/nd_vae/purification_model.py:18-31
  Defence_NVAE.forward                   src/defenses/competitors/nd_vae/modules/models/NVAE.py:688-720
  Residual_Cell_NVAE, FactorizedReduce   NVAE.py:255-297, :117-135
  Generative_Cell_NVAE                   NVAE.py:156-228
  Encoder_tower / Decoder_tower / Decoder_group / Sampler / Normal    NVAE.py:380-444, :472-575, :449-469, :583-634, :88-101
  DiscMixLogistic.mean                   src/defenses/competitors/nd_vae/modules/models/NVAE_utils.py:224-248

with the random draws passed in explicitly: the input noise, one eps per sampler (the reference draws `mu.mul(0).normal_()` per
sampler, NVAE.py:82-86) and the unregistered `Decoder_tower.h` (ndvae_spec.py header).  BatchNorm layers are evaluated with their
running statistics (the reference calls `.eval()`, load_defense.py:120).  Pinned by tests/golden/ndvae.npz, which
tests/golden/make_ndvae_golden.py produces by running the reference's own `Defence_NVAE` / `NDVaeDefenseModel`.
"""
from __future__ import annotations

from typing import Dict, List, Sequence

import torch
import torch.nn.functional as F

from gen_adversarial_amd.ndvae_spec import NdGenCell, NdResCell, NdvaeSpec

from . import kinks as K

SD = Dict[str, torch.Tensor]


def _bn(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    return F.batch_norm(x, sd[f'{p}.running_mean'], sd[f'{p}.running_var'], sd[f'{p}.weight'], sd[f'{p}.bias'], False, 0.0, 1e-5)


def _conv(sd: SD, p: str, x: torch.Tensor, stride: int = 1, padding: int = 0, groups: int = 1) -> torch.Tensor:
    return F.conv2d(x, sd[f'{p}.weight'], sd[f'{p}.bias'], stride=stride, padding=padding, groups=groups)


def _swish(x):
    return x * torch.sigmoid(x)


def se_block(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    """SE_Block (NVAE.py:57-69): x * sigmoid(W2 relu(W1 mean(x)))"""
    s = x.mean(dim=(2, 3))
    s = K.relu(F.linear(s, sd[f'{p}.se.0.weight'], sd[f'{p}.se.0.bias']))
    s = torch.sigmoid(F.linear(s, sd[f'{p}.se.2.weight'], sd[f'{p}.se.2.bias']))
    return x * s[:, :, None, None]


def factorized_reduce(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    out = _swish(x)
    return torch.cat([_conv(sd, f'{p}.conv_1', out, 2), _conv(sd, f'{p}.conv_2', out[:, :, 1:, 1:], 2),
                      _conv(sd, f'{p}.conv_3', out[:, :, :, 1:], 2), _conv(sd, f'{p}.conv_4', out[:, :, 1:, :], 2)], dim=1)


def res_cell(sd: SD, c: NdResCell, x: torch.Tensor) -> torch.Tensor:
    p = c.prefix
    skip = factorized_reduce(sd, f'{p}.skip', x) if c.down else x
    t = _conv(sd, f'{p}.conv1', _swish(_bn(sd, f'{p}.bn1', x)), 2 if c.down else 1, 1)
    t = _conv(sd, f'{p}.conv2', _swish(_bn(sd, f'{p}.bn2', t)), 1, 1)
    return skip + se_block(sd, f'{p}.squeeze_excitation', t)


def gen_cell(sd: SD, c: NdGenCell, x: torch.Tensor) -> torch.Tensor:
    p = c.prefix
    if c.up:
        skip = _conv(sd, f'{p}.skip.1', F.interpolate(x, scale_factor=2, mode='bilinear', align_corners=True))
        t = F.interpolate(x, scale_factor=2, mode='nearest')
    else:
        skip, t = x, x
    t = _bn(sd, f'{p}.bn_expanded1', _conv(sd, f'{p}.expand', _bn(sd, f'{p}.bn1', t)))
    t = _conv(sd, f'{p}.dep_sep_conv.depthwise', _swish(t), padding=2, groups=c.hidden)
    t = _bn(sd, f'{p}.bn_expanded2', _conv(sd, f'{p}.dep_sep_conv.pointwise', t))
    t = _bn(sd, f'{p}.bn2', _conv(sd, f'{p}.expand2', _swish(t)))
    return skip + se_block(sd, f'{p}.squeeze_excitation', t)


def soft_clamp5(x):
    return 5.0 * torch.tanh(x / 5.0)


def sampler(sd: SD, p: str, x: torch.Tensor, eps: torch.Tensor) -> torch.Tensor:
    """Sampler.forward (NVAE.py:608-634): q = Normal(mu_q + mu_p, log_sig_q + log_sig_p), z = mu + sigma * eps with
    mu = soft_clamp5(.), sigma = exp(soft_clamp5(.)) + 1e-2 (Normal.__init__, :89-95)"""
    mu_p, ls_p = torch.chunk(_conv(sd, f'{p}.prior_cell.1', F.elu(x)), 2, dim=1)
    mu_q, ls_q = torch.chunk(_conv(sd, f'{p}.cell', x, padding=1), 2, dim=1)
    mu = soft_clamp5(mu_q + mu_p)
    sigma = torch.exp(soft_clamp5(ls_q + ls_p)) + 1e-2
    return eps * sigma + mu


def ndvae_logits(sd: SD, spec: NdvaeSpec, x: torch.Tensor, eps: Sequence[torch.Tensor], h: torch.Tensor) -> torch.Tensor:
    """Defence_NVAE.forward(x)[0]: the mixture logits [B, 100, D, D]"""
    x = torch.clamp(x, 0, 1) * 2.0 - 1.0
    x = _conv(sd, 'stem', x, padding=1)
    for c in spec.pre_cells:
        x = res_cell(sd, c, x)
    outs = [x]
    for cells in spec.enc_scales:
        for c in cells:
            x = res_cell(sd, c, x)
        outs.append(x)
    latent = outs[::-1]
    S = len(spec.enc_scales)
    z = sampler(sd, 'decoder.samplers.0', latent[0], eps[0])
    hb = h.unsqueeze(0).expand(z.shape[0], -1, -1, -1)
    out = _conv(sd, 'decoder.combiner_cells.0.conv', torch.cat([z, hb], dim=1))
    for s, sc in enumerate(spec.dec_scales):
        y = out
        for grp in sc.groups:
            t = y
            for c in grp.cells:
                t = gen_cell(sd, c, t)
            y = _conv(sd, f'{grp.prefix}.combiner.conv', torch.cat([y, t], dim=1))
        if sc.up is not None:
            y = gen_cell(sd, sc.up, y)
        comb = latent[s + 1] + _conv(sd, f'encoder.combiner_cells.{s}.conv', y)
        z = sampler(sd, f'decoder.samplers.{s + 1}', comb, eps[s + 1])
        out = _conv(sd, f'decoder.combiner_cells.{s + 1}.conv', torch.cat([z, y], dim=1))
    for c in spec.post_cells:
        out = gen_cell(sd, c, out)
    return _conv(sd, 'image_conditional.1', F.elu(out), padding=1)


def dml_mean(logits: torch.Tensor, nmix: int = 10) -> torch.Tensor:
    """DiscMixLogistic(logits).mean() (NVAE_utils.py:91-117, 224-248) -> [B, 3, H, W] in [0, 1]"""
    b, _, hh, ww = logits.shape
    probs = torch.softmax(logits[:, :nmix], dim=1).unsqueeze(2)                       # B, M, 1, H, W
    par = logits[:, nmix:].reshape(b, nmix, 9, hh, ww)
    m, k = par[:, :, 0:3], torch.tanh(par[:, :, 6:9])
    mu, kk = (m * probs).sum(dim=1), (k * probs).sum(dim=1)
    r = K.clamp(mu[:, 0], -1.0, 1.0)
    g = K.clamp(mu[:, 1] + kk[:, 0] * r, -1.0, 1.0)
    bl = K.clamp(mu[:, 2] + kk[:, 1] * r + kk[:, 2] * g, -1.0, 1.0)
    return (torch.stack([r, g, bl], dim=1) + 1.0) / 2.0


def ndvae_purify(sd: SD, spec: NdvaeSpec, x: torch.Tensor, noise: torch.Tensor, noise_std: float, eps: Sequence[torch.Tensor],
                 h: torch.Tensor) -> torch.Tensor:
    """NDVaeDefenseModel.purify (purification_model.py:18-26) with the N(0,1) draws passed in"""
    x = K.clamp(x + noise * noise_std, 0.0, 1.0)
    return dml_mean(ndvae_logits(sd, spec, x, eps, h), spec.num_mixtures)
