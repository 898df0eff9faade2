"""
ND-VAE competitor defender with the reference's surface (src/defenses/competitors/nd_vae/purification_model.py:7-31;
built by src/experiments/load_defense.py:108-124):

    nd_vae = load_NDVAE(autoencoder_path, x_channels, encoding_channels, pre_proc_groups, scales, groups, cells, image_size)
    NDVaeDefenseModel(base_classifier, nd_vae, noise_std)      .purify(x) -> purified image     forward(x) -> logits

`nd_vae` is a weight holder (state dict with the keys of the reference's `Defence_NVAE`, its structural spec, and the
decoder's `h`) instead of an nn.Module: the arithmetic runs in the HIP engine (engine_ndvae.build_ndvae_defense), forward and
backward-to-input, under the EoT wrapper like every other defender.  `h` (`Decoder_tower.h`) is not part of the reference's
checkpoints — `nn.Parameter(torch.rand(...)).unsqueeze(0).to(device)` is a plain tensor (NVAE.py:490) — so, as there, it is a
fresh uniform draw per loaded model (pass `h=` to pin it).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Optional

import torch

from ...engine import Engine
from ...ndvae_spec import NdvaeSpec, build_ndvae_spec
from ..ours.abstract_models import BaseClassificationModel, _EngineOwner


@dataclass
class NdvaeWeights:
    state_dict: Dict[str, torch.Tensor]
    cfg: dict
    h: torch.Tensor

    @property
    def spec(self) -> NdvaeSpec:
        return build_ndvae_spec(self.cfg)

    def to(self, device):
        return self

    def eval(self):
        return self


def load_NDVAE(path, x_channels: int, encoding_channels: int, pre_proc_groups: int, scales: int, groups: int, cells: int,
               image_size: int, h: Optional[torch.Tensor] = None) -> NdvaeWeights:
    """`Defence_NVAE(x_channels, encoding_channels, pre_proc_groups, scales, groups, cells, image_size)` +
    `load_state_dict(torch.load(path))` of load_defense.py:110-120.  `path` may also be an in-memory state dict."""
    cfg = {'x_channels': x_channels, 'encoding_channels': encoding_channels, 'pre_proc_groups': pre_proc_groups, 'scales': scales,
           'groups': groups, 'cells': cells, 'input_dim': image_size}
    spec = build_ndvae_spec(cfg)
    sd = path if isinstance(path, dict) else torch.load(path, map_location='cpu')
    need = ('stem.weight', f'{spec.pre_cells[0].prefix}.conv1.weight', 'decoder.samplers.0.cell.weight', 'image_conditional.1.weight')
    missing = [k for k in need if k not in sd]
    if missing:
        raise KeyError(f'not a Defence_NVAE state dict for this configuration: {missing} missing')
    if h is None:
        h = torch.rand(spec.h_shape)
    if tuple(h.shape) != tuple(spec.h_shape):
        raise ValueError(f'h must be {spec.h_shape}, got {tuple(h.shape)}')
    return NdvaeWeights({k: v.detach().float() for k, v in sd.items() if torch.is_tensor(v) and v.is_floating_point()}, cfg, h.float())


class NDVaeDefenseModel(torch.nn.Module, _EngineOwner):

    def __init__(self, base_classifier: BaseClassificationModel, nd_vae: NdvaeWeights, noise_std: float):
        torch.nn.Module.__init__(self)
        self.base_classifier = base_classifier
        self.purifier = nd_vae
        self.noise_std = float(noise_std)
        self._init_engines(base_classifier.device)
        self.bpda = False

    def _make_engine(self, rows: int, rep: int, with_noise: bool = True) -> Engine:
        nd, clf = self.purifier, self.base_classifier.classifier
        D = nd.cfg['input_dim']
        eng = Engine.bare(rows, device=self.device, store=self._store, rep=rep, resolution=(3, D, D), alphas=[],
                          noise_eps=self.noise_std if with_noise else 0.0)
        return eng.build_ndvae_defense(nd.state_dict, nd.spec, nd.h, clf.state_dict, clf.spec)

    def forward_rows(self, batch: torch.Tensor, rep: int = 1, preds_only: bool = True):
        logits, purified = self._run(batch, rep, not preds_only)
        return logits if preds_only else (logits, purified)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.forward_rows(x, 1)

    def purify(self, x: torch.Tensor) -> torch.Tensor:
        """x + N(0, noise_std), clamp, Defence_NVAE, DiscMixLogistic mean (purification_model.py:18-26)"""
        return self._run(x, 1, True)[1]
