"""
Concrete classifier / defender classes with the reference's names and constructor signatures
(src/defenses/ours/models.py:17-353).  The CelebA-identities pair (VGG-11 classifier + NVAE defender) and the gender
ResNet-50 classifier are built; the e4e/StyleGAN2 purifier and the cars pair (ResNeXt-50 + Style-Transformer) are "next" rows.
"""
from __future__ import annotations

import torch

from ...engine import Engine
from ..loading_utils import load_NVAE, load_ResNet50, load_ResNext50, load_Vgg11, NVAEWeights
from .abstract_models import BaseClassificationModel, MLVGMDefenseModel


class CelebaIdentityClassifier(BaseClassificationModel, torch.nn.Module):
    """CelebA-64 identities VGG-11 (models.py:38-56)."""

    def __init__(self, model_path: str, device: str):
        torch.nn.Module.__init__(self)
        BaseClassificationModel.__init__(self, model_path, device, (0.5, 0.5, 0.5), (0.5, 0.5, 0.5))

    def load_classifier(self, model_path: str, device: str):
        return load_Vgg11(model_path, device)


class CelebaGenderClassifier(BaseClassificationModel, torch.nn.Module):
    """CelebA-HQ 256 gender ResNet-50 (models.py:17-35)."""

    def __init__(self, model_path: str, device: str):
        torch.nn.Module.__init__(self)
        self.image_size = 256
        BaseClassificationModel.__init__(self, model_path, device, (0.5, 0.5, 0.5), (0.5, 0.5, 0.5))

    def load_classifier(self, model_path: str, device: str):
        return load_ResNet50(model_path, device)


class CarsTypeClassifier(BaseClassificationModel, torch.nn.Module):
    """Stanford-Cars 128 type classifier, ResNeXt-50 32x4d (models.py:59-77)."""

    def __init__(self, model_path: str, device: str):
        torch.nn.Module.__init__(self)
        self.image_size = 128
        BaseClassificationModel.__init__(self, model_path, device, (0.5, 0.5, 0.5), (0.5, 0.5, 0.5))

    def load_classifier(self, model_path: str, device: str):
        return load_ResNext50(model_path, device)


class NVAEDefenseModel(MLVGMDefenseModel, torch.nn.Module):
    """NVAE purifier (models.py:135-274)."""

    def __init__(self, classifier: BaseClassificationModel, autoencoder_path: str,
                 interpolation_alphas: tuple, alpha_attenuation: float = 1.0, initial_noise_eps: float = 0.0,
                 apply_gaussian_blur: bool = False, device: str = 'cpu', temperature: float = 0.6):
        torch.nn.Module.__init__(self)
        self.temperature = temperature
        MLVGMDefenseModel.__init__(self, classifier, autoencoder_path, interpolation_alphas, alpha_attenuation,
                                   initial_noise_eps, apply_gaussian_blur, device)

    def load_autoencoder(self, model_path: str, device: str) -> NVAEWeights:
        return load_NVAE(model_path, device, self.temperature)

    def _make_engine(self, rows: int, rep: int, with_noise: bool = True) -> Engine:
        ae, clf = self.autoencoder, self.classifier.classifier
        return Engine(ae.state_dict, ae.config, ae.resolution, clf.state_dict, clf.spec, rows=rows, rep=rep,
                      alphas=self.interpolation_alphas, temperature=self.temperature,
                      noise_eps=self.eps if with_noise else 0.0, blur=self.blur_input and with_noise,
                      share_encoder=True,      # EoT replicas share the encoder pass whenever no input noise is drawn
                      device=self.device, store=self._store)


def _next(name, what):
    class _NotBuilt:
        def __init__(self, *a, **k):
            raise NotImplementedError(f'{name}: {what} is a "next" row of SURVEY.md §8, not built yet')
    _NotBuilt.__name__ = name
    return _NotBuilt


E4EStyleGanDefenseModel = _next('E4EStyleGanDefenseModel', 'e4e + StyleGAN2 purifier (models.py:80-132)')
TransStyleGanDefenseModel = _next('TransStyleGanDefenseModel', 'Style-Transformer purifier (models.py:277-353)')
