"""
Alpha-learning objective on the fast forward path (reference: src/experiments/alpha_learning/common_utils.py:15-103).
`AlphaEvaluator.objective_function(alphas)` = EoT-32 accuracy of the defender, with the given interpolation alphas, on
a pre-computed adversarial set.  The reference walks the set one image at a time; here `batch_images` images x 32 EoT
rows go through the engine per call (the objective is forward-only and embarrassingly parallel over images).
BoTorch-based Bayesian optimisation stays third-party and is not reproduced; `random_search` is grid_search.py:44-72.
"""
from __future__ import annotations

import math
from typing import Sequence

import numpy as np
import torch

from ...defenses.ours.models import CelebaIdentityClassifier, NVAEDefenseModel
from ...defenses.wrappers import EoTWrapper


def get_linear_alphas(n: int) -> list:
    return [i / n for i in range(1, n + 1)]


def get_cosine_alphas(n: int) -> list:
    return [0.5 * (1 - math.cos(math.pi * (i / n))) for i in range(1, n + 1)]


def get_best_combination(folder: str) -> np.ndarray:
    alphas = np.load(f'{folder}/alphas.npy')
    accuracies = np.load(f'{folder}/accuracies.npy')[:, 0]
    return alphas[accuracies.argmax()]


class AlphaEvaluator:
    def __init__(self, args, device, images: torch.Tensor = None, labels: torch.Tensor = None, batch_images: int = 8):
        """args: classifier_type ('vgg-11' is the built path), classifier_path, autoencoder_path, [adv_images_path]."""
        self.device = device
        self.eot_steps = 32
        self.batch_images = batch_images
        if args.classifier_type != 'vgg-11':
            raise NotImplementedError(f"classifier type {args.classifier_type}: StyleGAN paths are next rows")
        args.image_size = 64
        self.alpha_attenuation = 0.7
        base = CelebaIdentityClassifier(args.classifier_path, device)
        n = len(getattr(args, 'initial_alphas', [0.] * 24))
        self.defense_model = NVAEDefenseModel(base, args.autoencoder_path, [0. for _ in range(n)],
                                              alpha_attenuation=0.7, device=device).eval()
        self.defense_model = EoTWrapper(self.defense_model, getattr(args, 'eot_steps', self.eot_steps)).eval()
        if images is None:
            from ..test_defense import folder_dataset
            images, labels = folder_dataset(args.adv_images_path, args.image_size)
        self.images, self.labels = images.to(device), labels.to(device)

    @torch.no_grad()
    def per_image_verdicts(self, alphas: Sequence[float]) -> torch.Tensor:
        """the (N,) bool tensor the objective averages: EoT-mean prediction == label, per image of the adversarial set"""
        alphas = alphas.cpu().tolist() if isinstance(alphas, torch.Tensor) else list(alphas)
        self.defense_model.model.interpolation_alphas = [a * self.alpha_attenuation for a in alphas]
        hits = []
        for i in range(0, self.images.shape[0], self.batch_images):
            x, y = self.images[i:i + self.batch_images], self.labels[i:i + self.batch_images]
            hits.append(torch.eq(self.defense_model(x).argmax(dim=1), y))
        return torch.cat(hits)

    @torch.no_grad()
    def objective_function(self, alphas: Sequence[float]) -> float:
        return torch.mean(self.per_image_verdicts(alphas).to(torch.float32)).item()


@torch.no_grad()
def random_search(evaluator: AlphaEvaluator, n_steps: int, seed: int = 0):
    """uniform random alphas, keep all (alphas, accuracy) pairs — grid_search.py:44-72"""
    n = len(evaluator.defense_model.model.interpolation_alphas)
    g = torch.Generator().manual_seed(seed)
    all_alphas, all_acc = [], []
    for _ in range(n_steps):
        alphas = torch.rand(n, generator=g)
        all_alphas.append(alphas)
        all_acc.append(evaluator.objective_function(alphas))
    return torch.stack(all_alphas).numpy(), np.asarray(all_acc, dtype=np.float32).reshape(-1, 1)
