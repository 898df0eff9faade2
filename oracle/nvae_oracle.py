"""
ORACLE — TEST INFRASTRUCTURE ONLY.  Never imported by the product path (gen_adversarial_amd/*):
only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it, and only as the checker.

CPU restatement (plain PyTorch fp32, functional, no nn.Module reuse from the reference) of the NVAE
purification path of SerezD/gen_adversarial, operating directly on a state dict in the reference's
checkpoint layout, with every random draw passed in explicitly.

Parity pin: tests/golden/nvae_*.npz were produced by importing the reference itself in the build
container (tests/golden/make_golden.py); tests/test_oracle_golden.py checks this file against them at 1e-5.

Each function cites the reference code it follows (paths relative to the reference root).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import torch
import torch.nn.functional as F

from oracle import kinks as K     # relu / leaky_relu / prelu / clamp / max_pool2d: the torch functions unless a test flips near-ties

from gen_adversarial_amd.nvae_spec import DecCellSpec, EncCellSpec, NVAESpec, build_spec

SD = Dict[str, torch.Tensor]


# ---------------------------------------------------------------------------------------------------------------
# building blocks
# ---------------------------------------------------------------------------------------------------------------

def wn_weight(sd: SD, prefix: str) -> torch.Tensor:
    """torch.nn.utils.parametrizations.weight_norm (dim=0): w = g * v / ||v||, norm over dims 1..3.
    Used by every `weight_norm(Conv2d(...))` in NVAE/model.py and NVAE/modules/architecture.py."""
    g = sd[f'{prefix}.parametrizations.weight.original0']
    v = sd[f'{prefix}.parametrizations.weight.original1']
    return torch._weight_norm(v, g, 0)


def wn_conv(sd: SD, prefix: str, x, stride=1, padding=0):
    return F.conv2d(x, wn_weight(sd, prefix), sd.get(f'{prefix}.bias'), stride=stride, padding=padding)


def bn_eval(sd: SD, prefix: str, x):
    """SyncBatchNorm(eps=1e-5) in eval mode == affine with running stats (architecture.py:120,123,165-173;
    model set to .eval() in loading_utils.py:65)."""
    return F.batch_norm(x, sd[f'{prefix}.running_mean'], sd[f'{prefix}.running_var'],
                        sd[f'{prefix}.weight'], sd[f'{prefix}.bias'], False, 0.0, 1e-5)


def se(sd: SD, prefix: str, x):
    """SE.forward — architecture.py:52-61."""
    b, c, _, _ = x.shape
    s = torch.mean(x, dim=[2, 3])
    s = K.relu(F.linear(s, sd[f'{prefix}.linear_1.weight'], sd[f'{prefix}.linear_1.bias']))
    s = torch.sigmoid(F.linear(s, sd[f'{prefix}.linear_2.weight'], sd[f'{prefix}.linear_2.bias']))
    return x * s.view(b, c, 1, 1)


def enc_cell(sd: SD, cell: EncCellSpec, x):
    """ResidualCellEncoder.forward — architecture.py:131-136; SkipDown.forward :77-82."""
    p = cell.prefix
    stride = 2 if cell.down else 1
    r = F.silu(bn_eval(sd, f'{p}.residual.0', x))
    r = wn_conv(sd, f'{p}.residual.2', r, stride=stride, padding=1)
    r = F.silu(bn_eval(sd, f'{p}.residual.3', r))
    r = wn_conv(sd, f'{p}.residual.5', r, stride=1, padding=1)
    r = se(sd, f'{p}.residual.6', r)
    if cell.down:
        skip = wn_conv(sd, f'{p}.skip_connection.conv', F.silu(x), stride=2)
    else:
        skip = x
    return skip + 0.1 * r


def dec_cell(sd: SD, cell: DecCellSpec, x):
    """ResidualCellDecoder.forward — architecture.py:181-186; SkipUp.forward :91-93."""
    p, o = cell.prefix, cell.ridx
    r = x
    if cell.up:
        r = F.interpolate(r, scale_factor=2, mode='nearest')               # nn.UpsamplingNearest2d(2)
    r = bn_eval(sd, f'{p}.residual.{o + 0}', r)
    r = F.conv2d(r, sd[f'{p}.residual.{o + 1}.weight'])
    r = F.silu(bn_eval(sd, f'{p}.residual.{o + 2}', r))
    r = F.conv2d(r, sd[f'{p}.residual.{o + 4}.weight'], padding=2, groups=cell.hidden)
    r = F.silu(bn_eval(sd, f'{p}.residual.{o + 5}', r))
    r = F.conv2d(r, sd[f'{p}.residual.{o + 7}.weight'])
    r = bn_eval(sd, f'{p}.residual.{o + 8}', r)
    r = se(sd, f'{p}.residual.{o + 9}', r)
    if cell.up:
        skip = F.interpolate(x, scale_factor=2, mode='bilinear', align_corners=True)
        skip = wn_conv(sd, f'{p}.skip_connection.conv', skip)
    else:
        skip = x
    return skip + 0.1 * r


def nf_cell(sd: SD, prefix: str, z, hidden: int):
    """NFCell.forward — architecture.py:221-239 with MaskedConv2d (:9-34): weights multiplied by their masks."""
    def w(i):
        return sd[f'{prefix}.layers.{i}.weight'] * sd[f'{prefix}.layers.{i}.mask']
    h = F.elu(F.conv2d(z, w(0), sd[f'{prefix}.layers.0.bias'], padding=1))
    h = F.elu(F.conv2d(h, w(2), sd[f'{prefix}.layers.2.bias'], padding=2, groups=hidden))
    h = F.conv2d(h, w(4), sd[f'{prefix}.layers.4.bias'])
    return z - h


def nf_blocks(sd: SD, spec: NVAESpec, key: str, z):
    """nf_cells['nf_s:g'] = Sequential of NFBlocks (cell1 then mirrored cell2) — model.py:216-221, architecture.py:242-253."""
    for n in range(spec.num_nf_cells):
        z = nf_cell(sd, f'nf_cells.nf_{key}.{n}.cell1', z, spec.num_latent * 6)
        z = nf_cell(sd, f'nf_cells.nf_{key}.{n}.cell2', z, spec.num_latent * 6)
    return z


def soft_clamp(x, n: float = 5.0):
    """distributions.py:20-29."""
    return torch.tanh(x / n) * n


def normal_mu_sigma(mu, log_sigma, temp: float = 1.0):
    """Normal.__init__ — distributions.py:33-35."""
    return soft_clamp(mu), temp * torch.exp(soft_clamp(log_sigma))


def disc_mix_logistic_mean(logits: torch.Tensor, num_mixtures: int) -> torch.Tensor:
    """DiscMixLogistic.__init__ + mean — distributions.py:103-129, 231-254 (3-channel images)."""
    b, _, h, w = logits.shape
    n = num_mixtures
    mix = logits[:, :n].reshape(b, n, h * w)
    rest = logits[:, n:].reshape(b, n, 9, h * w)            # 'b (n c) h w -> b n c (h w)', c = 9
    means, _log_scales, coeffs = rest[:, :, 0:3], rest[:, :, 3:6], torch.tanh(rest[:, :, 6:9])
    probs = torch.softmax(mix, dim=1).unsqueeze(2)
    mu = torch.sum(means * probs, dim=1)                    # B, 3, HW
    k = torch.sum(coeffs * probs, dim=1)
    r = K.clamp(mu[:, 0], -1.0, 1.0)
    g = K.clamp(mu[:, 1] + k[:, 0] * r, -1.0, 1.0)
    bl = K.clamp(mu[:, 2] + k[:, 1] * r + k[:, 2] * g, -1.0, 1.0)
    return torch.stack([r, g, bl], dim=1).reshape(b, 3, h, w)


# ---------------------------------------------------------------------------------------------------------------
# NVAEDefenseModel.purify
# ---------------------------------------------------------------------------------------------------------------

def nvae_purify(sd: SD, spec: NVAESpec, batch: torch.Tensor, alphas: Sequence[float], eps: List[torch.Tensor],
                temperature: float = 0.6, return_latents: bool = False):
    """
    NVAEDefenseModel.purify — src/defenses/ours/models.py:160-274.

    :param batch: (B,3,H,W) in [0,1]
    :param alphas: interpolation_alphas already multiplied by alpha_attenuation (abstract_models.py:107)
    :param eps: one N(0,1) tensor per latent group, shape (B, NL, h_s, w_s), in the order the reference
                draws them (models.py:206 then :250 per group); drawn even when alpha == 0.
    """
    b = batch.shape[0]
    x = (batch - 0.5) / 0.5                                                  # models.py:170 (kornia Normalize)
    x = wn_conv(sd, 'preprocessing_block.init_conv', x, padding=1)          # model.py:106-107
    for cell in spec.pre_cells:
        x = enc_cell(sd, cell, x)

    stash = {}
    for kind, payload in spec.enc_program:                                   # models.py:176-192
        if kind == 'stash':
            stash[payload] = x
        else:
            x = enc_cell(sd, payload, x)

    x = F.elu(wn_conv(sd, 'encoder_0.1', F.elu(x)))                          # models.py:195; model.py:184-187

    latents = []
    g0 = spec.groups[0]
    mu_q, _ = torch.chunk(wn_conv(sd, 'enc_sampler.sampler_0:0', x, padding=1), 2, dim=1)   # models.py:198
    enc_mu = soft_clamp(mu_q)
    dec_mu, dec_sigma = normal_mu_sigma(torch.zeros_like(mu_q), torch.zeros_like(mu_q), temperature)
    a = float(alphas[g0.latent_idx])
    z = (1 - a) * enc_mu + a * (eps[0] * dec_sigma + dec_mu)                 # models.py:206 (+ Normal.sample :43-45)
    z = nf_blocks(sd, spec, '0:0', z)                                        # models.py:209-210
    latents.append(z)

    x = sd['const_prior'].expand(b, -1, -1, -1)                              # models.py:215
    x = wn_conv(sd, 'decoder_combiners.combiner_0:0.conv', torch.cat([x, z], dim=1))        # models.py:218

    for gs in spec.groups:
        if gs.dec_cells:
            for cell in gs.dec_cells:
                x = dec_cell(sd, cell, x)                                    # models.py:233-234
            key = f'{gs.s}:{gs.g}'
            comb = stash[key] + wn_conv(sd, f'encoder_combiners.combiner_{key}.conv', x)    # architecture.py:195-202
            mu_q, _ = torch.chunk(wn_conv(sd, f'enc_sampler.sampler_{key}', comb, padding=1), 2, dim=1)
            mu_p, log_sig_p = torch.chunk(wn_conv(sd, f'dec_sampler.sampler_{key}.1', F.elu(x)), 2, dim=1)
            enc_mu = soft_clamp(mu_p + mu_q)                                 # models.py:246
            dec_mu, dec_sigma = normal_mu_sigma(mu_p, log_sig_p, temperature)   # models.py:247
            a = float(alphas[gs.latent_idx])
            z = (1 - a) * enc_mu + a * (eps[gs.latent_idx] * dec_sigma + dec_mu)   # models.py:249-250
            z = nf_blocks(sd, spec, key, z)                                  # models.py:253-254
            latents.append(z)
            x = wn_conv(sd, f'decoder_combiners.combiner_{key}.conv', torch.cat([x, z], dim=1))   # models.py:257
        if gs.g == spec.groups_per_scale[gs.s] - 1 and gs.s in spec.dec_up_cells:
            x = dec_cell(sd, spec.dec_up_cells[gs.s], x)                     # models.py:262-263

    for cell in spec.post_cells:
        x = dec_cell(sd, cell, x)                                            # models.py:266
    logits = wn_conv(sd, 'to_logits.1', F.elu(x), padding=1)                 # models.py:269
    rec = disc_mix_logistic_mean(logits, spec.num_mixtures)                  # models.py:271-272
    out = rec * 0.5 + 0.5                                                    # models.py:274 (kornia Denormalize)
    if return_latents:
        return out, latents, logits
    return out


def latent_shapes(spec: NVAESpec, rows: int):
    """Shapes of the eps tensors, in draw order."""
    return [(rows, spec.num_latent, gs.res, gs.res) for gs in spec.groups]


def draw_eps(spec: NVAESpec, rows: int, seed: int) -> List[torch.Tensor]:
    g = torch.Generator().manual_seed(seed)
    return [torch.randn(s, generator=g) for s in latent_shapes(spec, rows)]
