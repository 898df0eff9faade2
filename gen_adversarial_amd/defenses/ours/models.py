"""
Concrete classifier / defender classes with the reference's names and constructor signatures
(src/defenses/ours/models.py:17-353).  The CelebA-identities pair (VGG-11 classifier + NVAE defender) and the gender
ResNet-50 classifier with the e4e + StyleGAN2 purifier, and the cars ResNeXt-50 classifier with the Style-Transformer + StyleGAN2 purifier.
"""
from __future__ import annotations

import torch

from ...engine import Engine
from ..loading_utils import (E4EWeights, NVAEWeights, TransWeights, load_E4EStyleGan, load_NVAE, load_ResNet50, load_ResNext50,
                             load_TranStyleGan, load_Vgg11)
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

    @property
    def supports_class_jacobian(self) -> bool:
        from ...vgg_spec import VggSpec
        return isinstance(self.classifier.classifier.spec, VggSpec)

    supports_forward_only = True

    def _make_engine(self, rows: int, rep: int, with_noise: bool = True, cot_rep: int = 1, need_backward: bool = True) -> Engine:
        ae, clf = self.autoencoder, self.classifier.classifier
        return Engine(ae.state_dict, ae.config, ae.resolution, clf.state_dict, clf.spec, rows=rows, rep=rep,
                      alphas=self.interpolation_alphas, temperature=self.temperature,
                      noise_eps=self.eps if with_noise else 0.0, blur=self.blur_input and with_noise,
                      share_encoder=True,      # EoT replicas share the encoder pass whenever no input noise is drawn
                      device=self.device, store=self._store, cot_rep=cot_rep, need_backward=need_backward)


class E4EStyleGanDefenseModel(MLVGMDefenseModel, torch.nn.Module):
    """e4e encoder + StyleGAN2 purifier (models.py:80-132): encode, mix every latent index with a freshly mapped style
    (alpha per index), decode with the fixed noise buffers, face_pool; (0.5, 0.5) normalisation around the autoencoder.
    One HIP plan pair per (rows, EoT) including the ResNet classifier (engine_stylegan.build_e4e_defense).
    Deliberate differences: generator sizes below 256 px (reduced test checkpoints) skip face_pool instead of being enlarged
    to 256."""

    def __init__(self, classifier: BaseClassificationModel, autoencoder_path: str,
                 interpolation_alphas: tuple, alpha_attenuation: float = 1.0, initial_noise_eps: float = 0.0,
                 apply_gaussian_blur: bool = False, device: str = 'cpu'):
        torch.nn.Module.__init__(self)
        MLVGMDefenseModel.__init__(self, classifier, autoencoder_path, interpolation_alphas, alpha_attenuation,
                                   initial_noise_eps, apply_gaussian_blur, device, (0.5, 0.5, 0.5), (0.5, 0.5, 0.5))
        n = self.autoencoder.decoder_spec.n_latent
        if len(self.interpolation_alphas) != n:
            raise ValueError(f'{n} interpolation alphas expected (one per latent index), got {len(self.interpolation_alphas)}')

    def load_autoencoder(self, model_path: str, device: str) -> E4EWeights:
        return load_E4EStyleGan(model_path, device)

    def _make_engine(self, rows: int, rep: int, with_noise: bool = True) -> Engine:
        ae, clf = self.autoencoder, self.classifier.classifier
        res = getattr(self, 'image_size', 256)
        eng = Engine.bare(rows, device=self.device, store=self._store, rep=rep, resolution=(3, res, res),
                          alphas=self.interpolation_alphas, noise_eps=self.eps if with_noise else 0.0,
                          blur=self.blur_input and with_noise,
                          share_encoder=True)      # EoT replicas share the encoder pass whenever no input noise is drawn
        size = ae.decoder_spec.size
        return eng.build_e4e_defense(ae.encoder_sd, ae.encoder_spec, ae.decoder_sd, ae.decoder_spec, ae.latent_avg,
                                     clf.state_dict, clf.spec, pool_to=min(256, size))

    def forward_rows(self, batch: torch.Tensor, rep: int = 1, preds_only: bool = True):
        self.image_size = batch.shape[-1]
        return super().forward_rows(batch, rep, preds_only)

    def purify(self, batch: torch.Tensor) -> torch.Tensor:
        """normalised images (B,3,H,W) in [-1, 1] -> normalised reconstructions, as the reference's purify (models.py:105-132);
        values outside [-1, 1] are clamped at the engine's image boundary"""
        self.image_size = batch.shape[-1]
        return self._run(batch * 0.5 + 0.5, 1, True, with_noise=False)[1] * 2.0 - 1.0


class TransStyleGanDefenseModel(MLVGMDefenseModel, torch.nn.Module):
    """Style-Transformer + StyleGAN2 purifier of the cars experiment (models.py:277-353): resize to 256 and crop rows 32:-32,
    encode with the 16 learned queries (GradualStyleEncoder), add latent_avg, mix every latent index with a freshly mapped
    N(0, 0.8) style (alpha per index), decode with the fixed noise buffers, face_pool, paint the cropped band -1, resize back to
    the input size; (0.5, 0.5) normalisation around the autoencoder.  One HIP plan pair per (rows, EoT) including the ResNeXt
    classifier (engine_trans.build_trans_defense).  Deliberate differences: generators smaller than the input image (reduced
    test checkpoints) keep their own resolution instead of being enlarged by face_pool."""

    def __init__(self, classifier: BaseClassificationModel, autoencoder_path: str,
                 interpolation_alphas: tuple, alpha_attenuation: float = 1.0, initial_noise_eps: float = 0.0,
                 apply_gaussian_blur: bool = False, device: str = 'cpu'):
        torch.nn.Module.__init__(self)
        MLVGMDefenseModel.__init__(self, classifier, autoencoder_path, interpolation_alphas, alpha_attenuation,
                                   initial_noise_eps, apply_gaussian_blur, device, (0.5, 0.5, 0.5), (0.5, 0.5, 0.5))
        n = self.autoencoder.encoder_spec.n_query
        if len(self.interpolation_alphas) != n:
            raise ValueError(f'{n} interpolation alphas expected (one per latent index), got {len(self.interpolation_alphas)}')

    def load_autoencoder(self, model_path: str, device: str) -> TransWeights:
        return load_TranStyleGan(model_path, device)

    def _make_engine(self, rows: int, rep: int, with_noise: bool = True) -> Engine:
        ae, clf = self.autoencoder, self.classifier.classifier
        res = getattr(self, 'image_size', 128)
        eng = Engine.bare(rows, device=self.device, store=self._store, rep=rep, resolution=(3, res, res),
                          alphas=self.interpolation_alphas, noise_eps=self.eps if with_noise else 0.0,
                          blur=self.blur_input and with_noise, share_encoder=True)
        # reference sizes: resize to 256 (2 x the 128-px cars images), crop 32 rows top and bottom; scaled with the image for
        # reduced test inputs
        return eng.build_trans_defense(ae.encoder_sd, ae.encoder_spec, ae.decoder_sd, ae.decoder_spec, ae.latent_avg,
                                       clf.state_dict, clf.spec, pool_to=min(res, ae.decoder_spec.size), mid=2 * res, crop=res // 4)

    def forward_rows(self, batch: torch.Tensor, rep: int = 1, preds_only: bool = True):
        self.image_size = batch.shape[-1]
        return super().forward_rows(batch, rep, preds_only)

    def purify(self, batch: torch.Tensor) -> torch.Tensor:
        """normalised images (B,3,H,W) in [-1, 1] -> normalised reconstructions, as the reference's purify (models.py:299-353)"""
        self.image_size = batch.shape[-1]
        return self._run(batch * 0.5 + 0.5, 1, True, with_noise=False)[1] * 2.0 - 1.0
