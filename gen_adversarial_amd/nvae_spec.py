"""
Structural description of the NVAE autoencoder on the purification path, and a seeded
parameter initialiser that produces a state dict with the reference's key names and shapes.

Reference for the module tree (names, channel bookkeeping, residual indices):
  src/mlvgms_autoencoders/NVAE/model.py:16-315          (AutoEncoder.__init__ and builders)
  src/mlvgms_autoencoders/NVAE/modules/architecture.py:37-218   (SE, cells, combiners)
Checkpoint layout consumed by the reference loader:
  src/defenses/loading_utils.py:51-66 (checkpoint['configuration'], 'state_dict_temp=<T>')

Nothing here is executed on the hot path; it only builds names, shapes and synthetic weights.
"""
from __future__ import annotations

from collections import OrderedDict
from dataclasses import dataclass, field
from typing import Dict, List, Tuple

import numpy as np
import torch

# the config assumed by SURVEY.md §6 / BASELINE.md §2 for sizing (real one lives in the checkpoint)
ASSUMED_NVAE_CONFIG = {
    'initial_channels': 32,
    'num_pre-post_process_blocks': 2,
    'num_pre-post_process_cells': 2,
    'num_scales': 3,
    'num_groups_per_scale': 8,
    'is_adaptive': False,
    'min_groups_per_scale': 1,
    'num_cells_per_group': 2,
    'num_latent_per_group': 20,
    'num_logistic_mixtures': 10,
    'num_nf_cells': None,
}
ASSUMED_NVAE_RESOLUTION = (3, 64, 64)


@dataclass
class EncCellSpec:
    """ResidualCellEncoder (architecture.py:96-136)."""
    prefix: str
    cin: int
    cout: int
    down: bool          # stride-2 first conv + SkipDown (architecture.py:64-82)


@dataclass
class DecCellSpec:
    """ResidualCellDecoder (architecture.py:139-186)."""
    prefix: str
    cin: int
    cout: int
    up: bool            # nearest x2 in the residual branch + SkipUp (architecture.py:85-93)
    hidden_mul: int

    @property
    def hidden(self) -> int:
        return self.cin * self.hidden_mul

    @property
    def ridx(self) -> int:
        """index offset of the residual Sequential (an UpsamplingNearest2d sits at 0 when up)."""
        return 1 if self.up else 0


@dataclass
class GroupSpec:
    s: int
    g: int
    channels: int
    res: int            # spatial size of this scale
    latent_idx: int     # index into interpolation_alphas / eps list (models.py:220-259)
    dec_cells: List[DecCellSpec] = field(default_factory=list)   # empty for 0:0


@dataclass
class NVAESpec:
    cfg: dict
    img_channels: int
    resolution: int
    base_channels: int
    num_scales: int
    groups_per_scale: List[int]
    cells_per_group: int
    num_latent: int
    num_mixtures: int
    num_nf_cells: int          # NFBlocks per latent group (0 = none; model.py:58-59, 216-221)
    pre_cells: List[EncCellSpec]
    # encoder tower in execution order: list of (kind, payload)
    #   ('cell', EncCellSpec) | ('stash', 's:g') | ('down', EncCellSpec)
    enc_program: List[Tuple[str, object]]
    enc0_channels: int
    const_prior_shape: Tuple[int, int, int, int]
    groups: List[GroupSpec]                  # decoder order (s ascending, g ascending)
    dec_up_cells: Dict[int, DecCellSpec]     # scale -> upsampling cell at the end of that scale
    post_cells: List[DecCellSpec]
    logits_in: int
    logits_out: int

    @property
    def num_latent_groups(self) -> int:
        return len(self.groups)


def build_spec(cfg: dict, resolution: Tuple[int, int, int]) -> NVAESpec:
    """Replays the channel bookkeeping of AutoEncoder.__init__ (model.py:27-85)."""
    img_c, res, _ = resolution
    C = cfg['initial_channels']
    n_blocks = cfg['num_pre-post_process_blocks']
    n_cells = cfg['num_pre-post_process_cells']
    n_scales = cfg['num_scales']
    gps = [max(cfg.get('min_groups_per_scale', 1), cfg['num_groups_per_scale'] // (2 ** i))
           if cfg.get('is_adaptive', False) else cfg['num_groups_per_scale'] for i in range(n_scales)]
    gps.reverse()                                                        # model.py:46-52
    cpg = cfg['num_cells_per_group']
    NL = cfg['num_latent_per_group']
    nmix = cfg['num_logistic_mixtures']

    mult = 1
    # --- preprocessing (model.py:97-130)
    pre_cells = []
    for b in range(n_blocks):
        for c in range(n_cells):
            ch = C * mult
            p = f'preprocessing_block.block_{b}.cell_{c}'
            if c != n_cells - 1:
                pre_cells.append(EncCellSpec(p, ch, ch, False))
            else:
                pre_cells.append(EncCellSpec(p, ch, ch * 2, True))
                mult *= 2

    # --- encoder tower (model.py:132-189); forward order follows models.py:176-192
    enc_program = []
    for s in range(n_scales - 1, -1, -1):
        ch = C * mult
        for g in range(gps[s]):
            for c in range(cpg):
                enc_program.append(('cell', EncCellSpec(f'encoder_tower.scale_{s}.group_{g}.cell_{c}', ch, ch, False)))
            if not (s == 0 and g == 0):
                enc_program.append(('stash', f'{s}:{g}'))
        if s > 0:
            enc_program.append(('down', EncCellSpec(f'encoder_tower.scale_{s}.downsampling', ch, ch * 2, True)))
            mult *= 2
    enc0_channels = C * mult

    scaling = 2 ** (n_blocks + n_scales - 1)                              # model.py:70-72
    prior_shape = (1, scaling * C, res // scaling, res // scaling)

    # --- decoder tower (model.py:237-272) with latent indices as in models.py:220-259
    groups, dec_up = [], {}
    latent_idx = 0
    dmult = mult
    for s in range(n_scales):
        ch = C * dmult
        r = res // (2 ** (n_blocks + n_scales - 1 - s))
        for g in range(gps[s]):
            gs = GroupSpec(s, g, ch, r, latent_idx)
            if not (s == 0 and g == 0):
                for c in range(cpg):
                    gs.dec_cells.append(DecCellSpec(f'decoder_tower.scale_{s}.group_{g}.cell_{c}', ch, ch, False, 6))
            groups.append(gs)
            latent_idx += 1
        if s < n_scales - 1:
            dec_up[s] = DecCellSpec(f'decoder_tower.scale_{s}.upsampling', ch, ch // 2, True, 6)
            dmult //= 2

    # --- postprocessing (model.py:274-300)
    post_cells = []
    for b in range(n_blocks):
        for c in range(n_cells):
            ch = C * dmult
            p = f'postprocessing_block.block_{b}.cell_{c}'
            if c != 0:
                post_cells.append(DecCellSpec(p, ch, ch, False, 3))
            else:
                post_cells.append(DecCellSpec(p, ch, ch // 2, True, 3))
                dmult //= 2

    logits_in = C * dmult
    logits_out = nmix + nmix * 3 * img_c                                   # model.py:302-315
    return NVAESpec(cfg, img_c, res, C, n_scales, gps, cpg, NL, nmix, int(cfg.get('num_nf_cells') or 0), pre_cells,
                    enc_program, enc0_channels,
                    prior_shape, groups, dec_up, post_cells, logits_in, logits_out)


# ----------------------------------------------------------------------------------------------------------------
# seeded synthetic parameters, reference key names (numpy RandomState: frozen bit stream across versions)
# ----------------------------------------------------------------------------------------------------------------

class _Rng:
    def __init__(self, seed: int):
        self.rs = np.random.RandomState(seed)

    def normal(self, shape, std=1.0):
        return torch.from_numpy((self.rs.standard_normal(size=shape) * std).astype(np.float32))

    def uniform(self, shape, lo, hi):
        return torch.from_numpy(self.rs.uniform(lo, hi, size=shape).astype(np.float32))


def _wn_conv(sd, rng, prefix, cout, cin_per_group, k, bias=True, gain=1.0):
    """weight_norm(Conv2d): parametrizations.weight.original0 = g (cout,1,1,1), original1 = v."""
    fan_in = cin_per_group * k * k
    v = rng.normal((cout, cin_per_group, k, k), std=1.0 / np.sqrt(fan_in))
    # g near the norm of v scaled so activations keep O(1) magnitude through ~100 cells
    g = v.flatten(1).norm(dim=1).view(cout, 1, 1, 1) * rng.uniform((cout, 1, 1, 1), 0.8, 1.2) * gain
    if bias:
        sd[f'{prefix}.bias'] = rng.normal((cout,), std=0.05)
    sd[f'{prefix}.parametrizations.weight.original0'] = g
    sd[f'{prefix}.parametrizations.weight.original1'] = v


def _bn(sd, rng, prefix, c):
    sd[f'{prefix}.weight'] = rng.uniform((c,), 0.8, 1.2)
    sd[f'{prefix}.bias'] = rng.normal((c,), std=0.1)
    sd[f'{prefix}.running_mean'] = rng.normal((c,), std=0.1)
    sd[f'{prefix}.running_var'] = rng.uniform((c,), 0.5, 1.5)
    sd[f'{prefix}.num_batches_tracked'] = torch.tensor(0, dtype=torch.long)


def _plain_conv(sd, rng, prefix, cout, cin_per_group, k):
    fan_in = cin_per_group * k * k
    sd[f'{prefix}.weight'] = rng.normal((cout, cin_per_group, k, k), std=1.0 / np.sqrt(fan_in))


def _se(sd, rng, prefix, c):
    h = max(c // 16, 4)
    sd[f'{prefix}.linear_1.weight'] = rng.normal((h, c), std=1.0 / np.sqrt(c))
    sd[f'{prefix}.linear_1.bias'] = rng.normal((h,), std=0.1)
    sd[f'{prefix}.linear_2.weight'] = rng.normal((c, h), std=1.0 / np.sqrt(h))
    sd[f'{prefix}.linear_2.bias'] = rng.normal((c,), std=0.1)


def _enc_cell(sd, rng, cell: EncCellSpec):
    p = cell.prefix
    if cell.down:
        _wn_conv(sd, rng, f'{p}.skip_connection.conv', cell.cout, cell.cin, 1)
    _bn(sd, rng, f'{p}.residual.0', cell.cin)
    _wn_conv(sd, rng, f'{p}.residual.2', cell.cout, cell.cin, 3)
    _bn(sd, rng, f'{p}.residual.3', cell.cout)
    _wn_conv(sd, rng, f'{p}.residual.5', cell.cout, cell.cout, 3)
    _se(sd, rng, f'{p}.residual.6', cell.cout)


def _dec_cell(sd, rng, cell: DecCellSpec):
    p, o = cell.prefix, cell.ridx
    if cell.up:
        _wn_conv(sd, rng, f'{p}.skip_connection.conv', cell.cout, cell.cin, 1)
    _bn(sd, rng, f'{p}.residual.{o + 0}', cell.cin)
    _plain_conv(sd, rng, f'{p}.residual.{o + 1}', cell.hidden, cell.cin, 1)
    _bn(sd, rng, f'{p}.residual.{o + 2}', cell.hidden)
    _plain_conv(sd, rng, f'{p}.residual.{o + 4}', cell.hidden, 1, 5)
    _bn(sd, rng, f'{p}.residual.{o + 5}', cell.hidden)
    _plain_conv(sd, rng, f'{p}.residual.{o + 7}', cell.cout, cell.hidden, 1)
    _bn(sd, rng, f'{p}.residual.{o + 8}', cell.cout)
    _se(sd, rng, f'{p}.residual.{o + 9}', cell.cout)


def nf_mask(shape, mirror: bool, zero_diag: bool) -> torch.Tensor:
    """MaskedConv2d mask (architecture.py:16-26): keep the first (kh*kw)//2 + zero_diag taps in row-major order (the
    last ones when mirrored).  NOTE: for a 1x1 kernel with zero_diag=False this keeps NOTHING — the last conv of every
    NFCell (architecture.py:233-234) therefore outputs its bias only."""
    co, ci, kh, kw = shape
    m = torch.ones(co, ci, kh * kw)
    half = (kh * kw) // 2 + int(zero_diag)
    m[:, :, half:] = 0
    if mirror:
        m = torch.flip(m, dims=(2,))
    return m.view(co, ci, kh, kw)


def _nf_cell(sd, rng, prefix, nl, mirror):
    hid = nl * 6
    for idx, (co, ci, k, zd) in ((0, (hid, nl, 3, True)), (2, (hid, 1, 5, False)), (4, (nl, hid, 1, False))):
        fan = ci * k * k
        sd[f'{prefix}.layers.{idx}.weight'] = rng.normal((co, ci, k, k), std=1.0 / np.sqrt(fan))
        sd[f'{prefix}.layers.{idx}.bias'] = rng.normal((co,), std=0.05)
        sd[f'{prefix}.layers.{idx}.mask'] = nf_mask((co, ci, k, k), mirror, zd)


def init_nvae_state_dict(cfg: dict, resolution: Tuple[int, int, int], seed: int = 0) -> "OrderedDict[str, torch.Tensor]":
    """
    Random weights in the exact state-dict layout of the reference AutoEncoder
    (checked key-for-key against the reference module by tests/golden/make_golden.py).
    BN running stats are non-trivial so that folding is exercised.
    """
    spec = build_spec(cfg, resolution)
    rng = _Rng(seed)
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    NL = spec.num_latent

    sd['const_prior'] = rng.uniform(spec.const_prior_shape, 0.0, 1.0)
    _wn_conv(sd, rng, 'preprocessing_block.init_conv', spec.base_channels, spec.img_channels, 3)
    for cell in spec.pre_cells:
        _enc_cell(sd, rng, cell)
    for kind, payload in spec.enc_program:
        if kind in ('cell', 'down'):
            _enc_cell(sd, rng, payload)
    for kind, payload in spec.enc_program:
        if kind == 'stash':
            ch = [gs.channels for gs in spec.groups if f'{gs.s}:{gs.g}' == payload][0]
            _wn_conv(sd, rng, f'encoder_combiners.combiner_{payload}.conv', ch, ch, 1)
    _wn_conv(sd, rng, 'encoder_0.1', spec.enc0_channels, spec.enc0_channels, 1)
    for gs in spec.groups:
        _wn_conv(sd, rng, f'enc_sampler.sampler_{gs.s}:{gs.g}', 2 * NL, gs.channels, 3, gain=0.5)
        if not (gs.s == 0 and gs.g == 0):
            _wn_conv(sd, rng, f'dec_sampler.sampler_{gs.s}:{gs.g}.1', 2 * NL, gs.channels, 1, gain=0.5)
    for gs in spec.groups:
        for n in range(spec.num_nf_cells):
            _nf_cell(sd, rng, f'nf_cells.nf_{gs.s}:{gs.g}.{n}.cell1', NL, False)
            _nf_cell(sd, rng, f'nf_cells.nf_{gs.s}:{gs.g}.{n}.cell2', NL, True)
    for gs in spec.groups:
        for cell in gs.dec_cells:
            _dec_cell(sd, rng, cell)
        _wn_conv(sd, rng, f'decoder_combiners.combiner_{gs.s}:{gs.g}.conv', gs.channels, gs.channels + NL, 1)
        last_g = spec.groups_per_scale[gs.s] - 1
        if gs.g == last_g and gs.s in spec.dec_up_cells:
            _dec_cell(sd, rng, spec.dec_up_cells[gs.s])
    for cell in spec.post_cells:
        _dec_cell(sd, rng, cell)
    _wn_conv(sd, rng, 'to_logits.1', spec.logits_out, spec.logits_in, 3)
    return sd


def nvae_checkpoint(cfg: dict, resolution, seed: int = 0, temperature: float = 0.6) -> dict:
    """A dict in the reference checkpoint layout (loading_utils.py:57-64)."""
    return {'configuration': {'autoencoder': dict(cfg), 'resolution': tuple(resolution)},
            f'state_dict_temp={temperature}': init_nvae_state_dict(cfg, resolution, seed)}
