"""
Structural description of the ND-VAE competitor purifier (`Defence_NVAE`, a third-party NVAE variant vendored by the
reference) and a seeded parameter initialiser with the reference's key names.

Reference for the module tree and the channel bookkeeping:
  src/defenses/competitors/nd_vae/modules/models/NVAE.py
      Residual_Cell_NVAE :255-297      BN -> Swish -> conv3x3(stride) -> BN -> Swish -> conv3x3 -> SE ;  out = skip(x) + cell(x)
      FactorizedReduce   :117-135      stride-2 skip: four 1x1/2 convs on the four pixel parities of swish(x), concatenated
      Generative_Cell_NVAE :156-228    [nearest x2] -> BN -> 1x1 (C -> E C) -> BN -> Swish -> depthwise 5x5 -> 1x1 (E C -> E C) -> BN ->
                                       Swish -> 1x1 (E C -> C') -> BN -> SE ;  out = skip(x) + cell(x)   (no 0.1 residual scale)
      Preproc_tower :311-343, Encoder_tower :380-444, Decoder_group :449-469, Decoder_tower :472-575, Sampler :583-634,
      Postproc_tower :347-377, Defence_NVAE :639-720 (forward :688-720)
  src/experiments/load_defense.py:108-124 (constructor arguments from the yaml keys x_channels, encoding_channels,
      pre_proc_groups, scales, groups, cells; input_dim = args.image_size)
Quirks of the reference that are reproduced, not repaired:
  * `Postproc_tower` builds its non-upsampling cells as `Generative_Cell_NVAE(channels, channels)`: the expansion factor E is the
    channel count itself (hidden width = channels^2);
  * `Decoder_tower.h` is `nn.Parameter(torch.rand(...)).unsqueeze(0).to(device)`: a plain tensor, NOT registered, so it is neither
    trained nor in the state dict — every instantiation draws a new one.  Here it is an explicit input (`h`), drawn by the loader.
  * `decoder.post_encoder` is constructed and never called.
Nothing here runs on the hot path; it builds names, shapes and synthetic weights.
"""
from __future__ import annotations

from collections import OrderedDict
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import torch


@dataclass
class NdResCell:
    """Residual_Cell_NVAE; `down`: stride-2 first conv + FactorizedReduce skip"""
    prefix: str
    cin: int
    cout: int
    down: bool
    res_scale: float = 1.0          # out = skip + 1.0 * SE(...)  (NVAE.py:293-297)


@dataclass
class NdGenCell:
    """Generative_Cell_NVAE; `up`: nearest x2 in the residual branch, bilinear x2 + 1x1 (C -> C/2) skip"""
    prefix: str
    cin: int
    cout: int
    up: bool
    hidden: int                     # E * cin
    res_scale: float = 1.0

    @property
    def ridx(self) -> int:
        return 1 if self.up else 0


@dataclass
class NdDecGroup:
    """Decoder_group: x = cells(dv); out = conv1x1(cat[dv, x])"""
    prefix: str
    channels: int
    cells: List[NdGenCell]


@dataclass
class NdScale:
    groups: List[NdDecGroup]
    up: Optional[NdGenCell]          # the channel-halving cell that closes every scale but the first
    out_channels: int
    res: int                         # spatial size of the scale's output


@dataclass
class NdvaeSpec:
    cfg: dict
    x_channels: int
    base: int                        # encoding_channels
    input_dim: int
    pre_cells: List[NdResCell]
    enc_scales: List[List[NdResCell]]     # encoder tower, bottom-up; the stride-2 cell is the last of every scale but the top one
    top_channels: int
    top_res: int
    h_shape: Tuple[int, int, int]
    sampler_channels: List[int]           # samplers[0..S]
    dec_scales: List[NdScale]
    latent_shapes: List[Tuple[int, int]]  # (channels, res) of z_0 .. z_S: the eps tensors the caller provides
    post_cells: List[NdGenCell]
    logits_in: int
    logits_out: int
    num_mixtures: int = 10


def build_ndvae_spec(cfg: dict) -> NdvaeSpec:
    """cfg: x_channels, encoding_channels, pre_proc_groups, scales, groups, cells, input_dim (load_defense.py:110-116)"""
    xc, C, P, S, G, Cc, D = (cfg[k] for k in ('x_channels', 'encoding_channels', 'pre_proc_groups', 'scales', 'groups', 'cells', 'input_dim'))
    if xc != 3:
        raise NotImplementedError('the DiscMixLogistic head of the reference works on 3-channel images only (NVAE_utils.py:100-101)')
    cur, res = C, D
    pre = []
    for g in range(P):                                              # Preproc_tower.__init__ (:320-334)
        for c in range(Cc):
            last = c == Cc - 1
            pre.append(NdResCell(f'pre_proc.groups_list.{g}.{c}', cur, cur * 2 if last else cur, last))
            if last:
                cur, res = cur * 2, res // 2
    enc_in = cur
    enc_scales, lat = [], [(cur, res)]                              # Encoder_tower (:396-444): outputs[0] = its input
    for s in range(S):
        cells = [NdResCell(f'encoder.enc_tower.{s}.{g}.{c}', cur, cur, False) for g in range(G) for c in range(Cc)]
        if s < S - 1:
            cells.append(NdResCell(f'encoder.enc_tower.{s}.{G}', cur, cur * 2, True))
            cur, res = cur * 2, res // 2
        enc_scales.append(cells)
        lat.append((cur, res))
    top, top_res = cur, res
    assert top == enc_in * 2 ** (S - 1)
    hs = max(D // 2 ** (S + 1), 4)                                  # Decoder_tower (:488-490)
    if hs != top_res:
        raise ValueError(f'Decoder_tower.h is {hs} x {hs} but the encoder ends at {top_res} x {top_res}: the reference\'s '
                         f'DecCombinerCell would fail on torch.cat for this configuration')
    dec_scales, cur, res = [], top, top_res
    for s in range(S):                                              # Decoder_tower (:497-522)
        groups = [NdDecGroup(f'decoder.dec_tower.{s}.{g}', cur,
                             [NdGenCell(f'decoder.dec_tower.{s}.{g}.group.{c}', cur, cur, False, 2 * cur) for c in range(Cc)])
                  for g in range(G)]
        up = None
        if s != 0:
            up = NdGenCell(f'decoder.dec_tower.{s}.{G}', cur, cur // 2, True, 2 * cur)
            cur, res = cur // 2, res * 2
        dec_scales.append(NdScale(groups, up, cur, res))
    base_out = cur
    mult, samplers = 2 ** (S - 1), []                               # samplers (:536-545)
    for i in range(S + 1):
        samplers.append(base_out * mult)
        if i != 0:
            mult //= 2
    # z_0 is drawn at the top; z_{s+1} at the output of decoder scale s
    latent_shapes = [(samplers[0], top_res)] + [(samplers[s + 1], dec_scales[s].res) for s in range(S)]
    post, mult = [], 2 ** P                                         # Postproc_tower (:360-373), in_channels = encoding_channels
    pres = res
    for b in range(P):
        for c in range(Cc):
            ch = C * mult
            if c == 0:
                post.append(NdGenCell(f'post_proc.tower.{b * Cc + c}', ch, ch // 2, True, 2 * ch))
                mult //= 2
                pres *= 2
            else:
                post.append(NdGenCell(f'post_proc.tower.{b * Cc + c}', ch, ch, False, ch * ch))      # E_param = channels (sic)
    if base_out != C * 2 ** P:
        raise ValueError('decoder output and post-processing input disagree (the reference would fail in BatchNorm)')
    return NdvaeSpec(cfg=dict(cfg), x_channels=xc, base=C, input_dim=D, pre_cells=pre, enc_scales=enc_scales, top_channels=top,
                     top_res=top_res, h_shape=(top, hs, hs), sampler_channels=samplers, dec_scales=dec_scales,
                     latent_shapes=latent_shapes, post_cells=post, logits_in=C, logits_out=10 + 10 * 3 * xc)


# ---------------------------------------------------------------------------------------------------------------------
def _bn(sd, g, prefix, c):
    sd[f'{prefix}.weight'] = torch.rand(c, generator=g) + 0.5
    sd[f'{prefix}.bias'] = 0.1 * torch.randn(c, generator=g)
    sd[f'{prefix}.running_mean'] = 0.1 * torch.randn(c, generator=g)
    sd[f'{prefix}.running_var'] = torch.rand(c, generator=g) + 0.5


def _conv(sd, g, prefix, cout, cin, k, groups=1, gain=1.0):
    fan = (cin // groups) * k * k
    sd[f'{prefix}.weight'] = gain * torch.randn(cout, cin // groups, k, k, generator=g) / fan ** 0.5
    sd[f'{prefix}.bias'] = 0.05 * torch.randn(cout, generator=g)


def _se(sd, g, prefix, c):
    hid = max(c // 16, 4)
    sd[f'{prefix}.se.0.weight'] = torch.randn(hid, c, generator=g) / c ** 0.5
    sd[f'{prefix}.se.0.bias'] = 0.1 * torch.randn(hid, generator=g)
    sd[f'{prefix}.se.2.weight'] = torch.randn(c, hid, generator=g) / hid ** 0.5
    sd[f'{prefix}.se.2.bias'] = 0.1 * torch.randn(c, generator=g)


def init_ndvae_state_dict(cfg: dict, seed: int = 0) -> "OrderedDict[str, torch.Tensor]":
    """Seeded weights under the PRIMARY key of every parameter / buffer of `Defence_NVAE` (each cell also exposes its layers a
    second time through `self.cell` — `cell.0.weight` aliases `bn1.weight` and so on: the same tensors; loaders read the primary
    names).  BatchNorm running statistics are non-trivial so that folding is exercised; residual branches are kept small (gain
    0.3 on the last conv of a cell: the reference adds them unscaled)."""
    spec = build_ndvae_spec(cfg)
    g = torch.Generator().manual_seed(seed)
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    _conv(sd, g, 'stem', spec.base, spec.x_channels, 3)

    def res_cell(c: NdResCell):
        p = c.prefix
        if c.down:
            q = c.cout // 4
            for i, co in enumerate((q, q, q, c.cout - 3 * q), start=1):
                _conv(sd, g, f'{p}.skip.conv_{i}', co, c.cin, 1)
        _bn(sd, g, f'{p}.bn1', c.cin)
        _conv(sd, g, f'{p}.conv1', c.cout, c.cin, 3)
        _bn(sd, g, f'{p}.bn2', c.cout)
        _conv(sd, g, f'{p}.conv2', c.cout, c.cout, 3, gain=0.3)
        _se(sd, g, f'{p}.squeeze_excitation', c.cout)

    def gen_cell(c: NdGenCell):
        p = c.prefix
        if c.up:
            _conv(sd, g, f'{p}.skip.1', c.cout, c.cin, 1)
        _bn(sd, g, f'{p}.bn1', c.cin)
        _bn(sd, g, f'{p}.bn2', c.cout)
        _bn(sd, g, f'{p}.bn_expanded1', c.hidden)
        _bn(sd, g, f'{p}.bn_expanded2', c.hidden)
        _conv(sd, g, f'{p}.expand', c.hidden, c.cin, 1)
        _conv(sd, g, f'{p}.dep_sep_conv.depthwise', c.hidden, c.hidden, 5, groups=c.hidden)
        _conv(sd, g, f'{p}.dep_sep_conv.pointwise', c.hidden, c.hidden, 1)
        _conv(sd, g, f'{p}.expand2', c.cout, c.hidden, 1, gain=0.3)
        _se(sd, g, f'{p}.squeeze_excitation', c.cout)

    for c in spec.pre_cells:
        res_cell(c)
    S = len(spec.enc_scales)
    # encoder.combiner_cells: inserted at index 0 scale by scale (:417-418): index i belongs to the channels of scale S-1-i
    enc_ch = [cells[0].cin for cells in spec.enc_scales]
    for i in range(S):
        _conv(sd, g, f'encoder.combiner_cells.{i}.conv', enc_ch[S - 1 - i], enc_ch[S - 1 - i], 1)
    for cells in spec.enc_scales:
        for c in cells:
            res_cell(c)
    _conv(sd, g, 'decoder.post_encoder.cell.1', spec.top_channels, spec.top_channels, 1)        # constructed, never called
    cur = spec.top_channels
    for s, sc in enumerate(spec.dec_scales):
        _conv(sd, g, f'decoder.combiner_cells.{s}.conv', cur, 2 * cur, 1)
        for grp in sc.groups:
            _conv(sd, g, f'{grp.prefix}.combiner.conv', grp.channels, 2 * grp.channels, 1)
            for c in grp.cells:
                gen_cell(c)
        if sc.up is not None:
            gen_cell(sc.up)
        cur = sc.out_channels
    _conv(sd, g, f'decoder.combiner_cells.{S}.conv', cur, 2 * cur, 1)
    for i, ch in enumerate(spec.sampler_channels):
        _conv(sd, g, f'decoder.samplers.{i}.cell', 2 * ch, ch, 3, gain=0.5)
        _conv(sd, g, f'decoder.samplers.{i}.prior_cell.1', 2 * ch, ch, 1, gain=0.5)
    for c in spec.post_cells:
        gen_cell(c)
    _conv(sd, g, 'image_conditional.1', spec.logits_out, spec.logits_in, 3)
    return sd


def init_ndvae_h(cfg: dict, seed: int = 0) -> torch.Tensor:
    """the unregistered `Decoder_tower.h` (torch.rand at construction, NVAE.py:490): [C, hs, hs]"""
    spec = build_ndvae_spec(cfg)
    return torch.rand(spec.h_shape, generator=torch.Generator().manual_seed(seed))
