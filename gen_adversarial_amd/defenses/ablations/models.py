"""
Noise-only / blur-only "purifiers" used as ablations, with the reference's surface
(src/defenses/ablations/models.py:13-66): `GaussianNoiseDefenseModel(base_classifier, eps=0.5)`,
`GaussianBlurDefenseModel(base_classifier)`, each with `.purify(x)` and `forward(x) -> logits`.
They reuse the pre-processing stage of the HIP engine (noise scaled to a per-row L2 of eps + clamp — the same
arithmetic as MLVGMDefenseModel.add_gaussian_noise — and the Gaussian-blur kernel) in front of the classifier.
"""
from __future__ import annotations

import torch

from ...engine import Engine
from ..ours.abstract_models import BaseClassificationModel, _EngineOwner


class _PreprocessDefense(torch.nn.Module, _EngineOwner):
    noise_eps = 0.0
    blur = False

    def __init__(self, base_classifier: BaseClassificationModel):
        torch.nn.Module.__init__(self)
        self.base_classifier = base_classifier
        self._init_engines(base_classifier.device)
        self._store = base_classifier._store                      # share the folded classifier weights

    def _make_engine(self, rows: int, rep: int, with_noise: bool = True) -> Engine:
        w = self.base_classifier.classifier
        r = getattr(self, 'image_size', 64)
        return Engine(None, None, (3, r, r), w.state_dict, w.spec, rows=rows, rep=rep, alphas=[], device=self.device,
                      store=self._store, noise_eps=self.noise_eps, blur=self.blur)

    def forward_rows(self, batch: torch.Tensor, rep: int = 1) -> torch.Tensor:
        self.image_size = batch.shape[-1]
        return self._run(batch, rep, False)[0]

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.forward_rows(x, 1)

    @torch.no_grad()
    def purify(self, x: torch.Tensor) -> torch.Tensor:
        """the pre-processed image the classifier sees (fresh noise draw for the noise ablation)"""
        self.image_size = x.shape[-1]
        self._run(x, 1, False)
        eng = self._engine(x.shape[0], 1)
        return eng.input_image_nchw()


class GaussianNoiseDefenseModel(_PreprocessDefense):
    def __init__(self, base_classifier: BaseClassificationModel, eps: float = 0.5):
        super().__init__(base_classifier)
        self.eps = eps
        self.noise_eps = eps


class GaussianBlurDefenseModel(_PreprocessDefense):
    blur = True
