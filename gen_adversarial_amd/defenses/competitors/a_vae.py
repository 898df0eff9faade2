"""
A-VAE competitor defender with the reference's surface (src/defenses/competitors/a_vae/purification_model.py:4-25; built by
src/experiments/load_defense.py:95-106):

    a_vae = load_AVAE(autoencoder_path, image_size)          # StyledGenerator(image_size) + load_state_dict(torch.load(path))
    AVaeDefenseModel(base_classifier, a_vae, kernel_size)    .purify(x) -> purified image     forward(x) -> logits

`a_vae` is a weight holder (state dict with the keys of the reference's `StyledGenerator`) instead of an nn.Module: the
arithmetic runs in the HIP engine (engine_avae.build_avae_defense), forward and backward-to-input.  Every call draws the latent
sample and the per-block noise images afresh, like `StyledGenerator.forward(noise=None)` (model.py:129-135).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict

import torch

from ...avae_spec import AvaeSpec, build_avae_spec
from ...engine import Engine
from ..ours.abstract_models import BaseClassificationModel, _EngineOwner


@dataclass
class AvaeWeights:
    state_dict: Dict[str, torch.Tensor]
    output_size: int
    width_div: int = 1

    @property
    def spec(self) -> AvaeSpec:
        return build_avae_spec(self.output_size, self.width_div)

    def to(self, device):
        return self

    def eval(self):
        return self


def load_AVAE(path, image_size: int, width_div: int = 1) -> AvaeWeights:
    """`path`: a file written by torch.save(state_dict) or an in-memory state dict"""
    sd = path if isinstance(path, dict) else torch.load(path, map_location='cpu')
    spec = build_avae_spec(image_size, width_div)
    need = ('encoder.conv2.conv1.conv.weight_orig', 'style.1.linear.weight_orig', 'generator.to_rgb.conv.weight_orig',
            f'generator.progression.{len(spec.blocks) - 1}.adain2.style.linear.weight_orig')
    missing = [k for k in need if k not in sd]
    if missing:
        raise KeyError(f'not a StyledGenerator({image_size}) state dict: {missing} missing')
    return AvaeWeights({k: v.detach().float() for k, v in sd.items() if torch.is_tensor(v) and v.is_floating_point()}, image_size, width_div)


class AVaeDefenseModel(torch.nn.Module, _EngineOwner):

    def __init__(self, base_classifier: BaseClassificationModel, purifier: AvaeWeights, kernel_size: int):
        torch.nn.Module.__init__(self)
        self.base_classifier = base_classifier
        self.purifier = purifier
        self.kernel_size = int(kernel_size)
        self._init_engines(base_classifier.device)
        self.bpda = False

    def _make_engine(self, rows: int, rep: int, with_noise: bool = True) -> Engine:
        av, clf = self.purifier, self.base_classifier.classifier
        D = av.output_size
        eng = Engine.bare(rows, device=self.device, store=self._store, rep=rep, resolution=(3, D, D), alphas=[])
        return eng.build_avae_defense(av.state_dict, av.spec, self.kernel_size, clf.state_dict, clf.spec)

    def forward_rows(self, batch: torch.Tensor, rep: int = 1, preds_only: bool = True):
        logits, purified = self._run(batch, rep, not preds_only)
        return logits if preds_only else (logits, purified)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.forward_rows(x, 1)

    def purify(self, x: torch.Tensor) -> torch.Tensor:
        """avg_pool2d(x * 2 - 1, kernel_size) -> StyledGenerator(inference=True) -> (x + 1) / 2 (purification_model.py:16-20)"""
        return self._run(x, 1, True)[1]
