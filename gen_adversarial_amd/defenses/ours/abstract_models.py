"""
Defender API with the reference's names, constructor arguments and call semantics
(src/defenses/ours/abstract_models.py:13-193), executed by the HIP engine.

  BaseClassificationModel(model_path, device, mean=None, std=None)   .classifier  .set_device(d)  __call__(batch)->logits
  MLVGMDefenseModel(classifier, autoencoder_path, interpolation_alphas, alpha_attenuation=1.0, initial_noise_eps=0.0,
                    apply_gaussian_blur=False, device='cpu', mean=None, std=None)
      .interpolation_alphas (mutable list, already multiplied by the attenuation)  .autoencoder  .classifier
      .purify(batch)   __call__(batch, preds_only=True) -> logits | (logits, purified)
Both are differentiable w.r.t. their input (torch.autograd.Function over the engine's backward plan); the backward may
be invoked several times per forward (`retain_graph=True` call sites, src/attacks/untargeted.py:529-535, :625).

Differences that are deliberate: `device` must be a GPU ('cuda:N') — there is no CPU execution path; the classifier
normalisation constants must be the (0.5, 0.5) pair every reference subclass uses (models.py:24-25,45-46,66-67).
"""
from __future__ import annotations

import math
from abc import ABC, abstractmethod
from typing import Dict, List, Optional, Tuple

import torch

from ...engine import Engine, WeightStore


class _EngineFn(torch.autograd.Function):
    """One forward of an Engine; backward replays the engine's backward plan (dX only)."""

    @staticmethod
    def forward(ctx, x: torch.Tensor, owner, eng: Engine, want_purified: bool):
        eng.x_in.copy_(x.detach())
        owner._fill_noise(eng)
        eng.forward()
        eng.version += 1
        ctx.eng, ctx.owner, ctx.version = eng, owner, eng.version
        ctx.saved_noise = owner._snapshot_noise(eng)
        ctx.x = x.detach().clone()
        logits = eng.logits.clone()
        if want_purified and eng.purified is not None:
            return logits, eng.purified.clone()
        if want_purified and getattr(eng, 'purified_s2d', None) is not None:     # e4e defender: classifier-layout image
            return logits, eng.purified_nchw()
        if want_purified and getattr(eng, 'purified_nhwc', None) is not None:    # A-VAE: the to_rgb output is the NHWC image
            return logits, eng.purified_nhwc.t[..., :3].permute(0, 3, 1, 2).contiguous()
        return logits, x.new_zeros(())

    @staticmethod
    def backward(ctx, dlogits, dpurified):
        eng: Engine = ctx.eng
        if eng.version != ctx.version:
            # another forward ran on this engine since: recompute with the saved inputs and noise
            eng.x_in.copy_(ctx.x)
            ctx.owner._restore_noise(eng, ctx.saved_noise)
            eng.forward()
            eng.version += 1
            ctx.version = eng.version
        use_purified = dpurified is not None and dpurified.dim() == 4 and eng.dpurified is not None
        use_logits = dlogits is not None
        if use_logits:
            eng.dlogits.view_as(eng.logits).copy_(dlogits)
        if use_purified:
            eng.dpurified.copy_(dpurified)
        if not use_logits and not use_purified:
            return torch.zeros_like(ctx.x), None, None, None
        if getattr(ctx.owner, 'bpda', False):               # BPDA: identity in place of the purifier's Jacobian
            if not use_logits:
                return torch.zeros_like(ctx.x), None, None, None
            eng.backward(identity_purifier=True)
        else:
            eng.backward(from_logits=use_logits, from_purified=use_purified)
        return eng.dx.clone(), None, None, None


class _EngineJacobian:
    """Per-class input gradients of the EoT-mean logits from ONE forward and ceil(n_columns / K) backward replays (the engine's
    K-cotangent backward plan).  `logits`: (B, n) EoT means; `grads()`: (B, n_columns, 3, H, W) with
    d mean_logit[b, classes[b, j]] / d x_b — what DeepFool (src/attacks/untargeted.py:526-560) and FAB (:605-635) obtain by one
    `.backward(retain_graph=True)` per class."""

    def __init__(self, owner, eng: Engine, x: torch.Tensor, rep: int, classes):
        self.owner, self.eng, self.rep, self.classes = owner, eng, rep, classes
        self.x = x.detach().clone()
        eng.x_in.copy_(self.x)
        owner._fill_noise(eng)
        eng.forward()
        eng.version += 1
        self.version, self.noise = eng.version, owner._snapshot_noise(eng)
        self.logits = eng.logits.view(x.shape[0], rep, -1).mean(dim=1)

    def grads(self) -> torch.Tensor:
        eng, B, rep, K = self.eng, self.x.shape[0], self.rep, self.eng.cot_rep
        if eng.version != self.version:                      # another forward ran on this engine since: same inputs, same noise
            eng.x_in.copy_(self.x)
            self.owner._restore_noise(eng, self.noise)
            eng.forward()
            eng.version += 1
            self.version = eng.version
        n = self.logits.shape[1]
        cols = self.classes if self.classes is not None else torch.arange(n, device=self.x.device).expand(B, n)
        out = []
        dl = eng.dlogits.view(B, rep, K, n)                  # cotangent k of defender row (b, j) at row (b * rep + j) * K + k
        for j0 in range(0, cols.shape[1], K):
            idx = cols[:, j0:j0 + K]                         # (B, k) class of cotangent k for image b
            k = idx.shape[1]
            dl.zero_()
            # d mean_j logits[b, j, c] / d logits[b, j, c] = 1 / rep for every replica j
            dl[:, :, :k].scatter_(3, idx.view(B, 1, k, 1).expand(B, rep, k, 1), 1.0 / rep)
            eng.backward()
            out.append(eng.dx.view(B, K, *eng.dx.shape[1:])[:, :k].clone())
        return torch.cat(out, dim=1)


class _EngineOwner:
    """Engines are built lazily per (rows, rep) and share one device copy of the folded weights."""

    # cotangent rows one K-cotangent backward replay may carry (rows x K): the 32-row reference protocol gets K = 16, i.e. all
    # 10 DeepFool classes in one replay and the 100 classes of the ids experiment in 7 instead of 100
    jacobian_cot_rows = 512

    def _init_engines(self, device):
        self.device = device
        dev = torch.device(device)
        if dev.type != 'cuda':
            raise RuntimeError("gen_adversarial_amd runs on the GPU only (device='cuda:N'); there is no CPU fallback")
        self._engines: Dict[Tuple[int, int], Engine] = {}
        self._store = WeightStore(dev)
        self._fixed_noise = None

    # owners whose _make_engine takes need_backward=False (the NVAE defender): calls that cannot be differentiated (torch.no_grad(),
    # or an input that does not require grad: clean predictions, get_purified, the alpha-learning objective) run on a FORWARD-ONLY
    # engine — no gradient buffers, and the post-processing decoder cells as ga_dec_cell_halo launches, whose forward kernel is
    # faster than the launches it replaces while its backward is not; logits and purified image are bitwise those
    # of the differentiable engine
    supports_forward_only = False

    def _engine(self, rows: int, rep: int, with_noise: bool = True, cot_rep: int = 1, forward_only: bool = False) -> Engine:
        forward_only = forward_only and self.supports_forward_only and cot_rep == 1
        key = (rows, rep, with_noise) if cot_rep == 1 else (rows, rep, with_noise, cot_rep)
        if forward_only:
            key = key + ('forward_only',)
        if key not in self._engines:
            self._engines[key] = (self._make_engine(rows, rep, with_noise, need_backward=False) if forward_only else
                                  self._make_engine(rows, rep, with_noise) if cot_rep == 1 else
                                  self._make_engine(rows, rep, with_noise, cot_rep=cot_rep))
        eng = self._engines[key]
        alphas = self._current_alphas()
        if alphas is not None:
            eng.set_alphas(alphas)
        return eng

    def _current_alphas(self):
        return None

    # noise handling -------------------------------------------------------------------------------------------
    def fixed_noise(self, eps: Optional[List[torch.Tensor]] = None, input_noise: Optional[torch.Tensor] = None):
        """Use the given N(0,1) draws for the next calls instead of fresh ones (parity tests); None resets.  A pinned draw may hold
        more rows than a call needs (attacks that drop finished images call with fewer rows): row r of a call uses draw r."""
        self._fixed_noise = None if eps is None and input_noise is None else (eps, input_noise)

    def _fill_noise(self, eng: Engine):
        eps, inp = self._fixed_noise if self._fixed_noise is not None else (None, None)
        for i, e in enumerate(eng.eps):                      # one draw per latent group, even when alpha == 0
            if eps is not None:
                e.copy_(eps[i][:e.shape[0]])
            else:
                e.normal_()
                if getattr(eng, 'eps_std', 1.0) != 1.0:      # TransStyleGanDefenseModel draws N(0, 0.8) (models.py:331)
                    e.mul_(eng.eps_std)
        if eng.noise is not None:                            # abstract_models.py:132-138
            if inp is not None:
                eng.noise.copy_(inp[:eng.noise.shape[0]])
            else:
                eng.noise.normal_()
            if getattr(eng, 'noise_is_std', False):          # competitors: x + randn * std (nd_vae/purification_model.py:21)
                eng.noise_coef.fill_(eng.noise_eps)
            else:
                eng.noise_coef.copy_(eng.noise_eps / eng.noise.flatten(1).norm(dim=1))

    @staticmethod
    def _snapshot_noise(eng: Engine):
        return [e.clone() for e in eng.eps], (None if eng.noise is None else (eng.noise.clone(), eng.noise_coef.clone()))

    @staticmethod
    def _restore_noise(eng: Engine, snap):
        for e, s in zip(eng.eps, snap[0]):
            e.copy_(s)
        if snap[1] is not None:
            eng.noise.copy_(snap[1][0])
            eng.noise_coef.copy_(snap[1][1])

    supports_class_jacobian = False          # owners whose _make_engine takes cot_rep (NVAE + VGG defender, VGG classifier)

    def class_jacobian_rows(self, batch: torch.Tensor, rep: int, classes=None):
        """_EngineJacobian of (B,3,H,W) images under EoT `rep`, or None when this owner has no K-cotangent plan (callers then
        fall back to one autograd backward per class)."""
        if not self.supports_class_jacobian or getattr(self, 'bpda', False):
            return None
        batch = batch.to(self.device, dtype=torch.float32).contiguous()
        rows = batch.shape[0] * rep
        n_cols = classes.shape[1] if classes is not None else None
        K = max(1, self.jacobian_cot_rows // rows)
        if n_cols is not None:
            K = min(K, n_cols)
        if K < 2:
            return None
        eng = self._engine(rows, rep, True, cot_rep=K)
        if tuple(batch.shape[2:]) != tuple(eng.resolution[1:]):
            raise ValueError(f'expected {eng.resolution[1]}x{eng.resolution[2]} images, got {tuple(batch.shape[2:])}')
        return _EngineJacobian(self, eng, batch, rep, classes)

    def _run(self, batch: torch.Tensor, rep: int, want_purified: bool, with_noise: bool = True):
        if batch.dim() != 4 or batch.shape[1] != 3:
            raise ValueError('expected a (B, 3, H, W) image batch')
        differentiable = torch.is_grad_enabled() and batch.requires_grad
        batch = batch.to(self.device, dtype=torch.float32).contiguous()
        eng = self._engine(batch.shape[0] * rep, rep, with_noise, forward_only=not differentiable)
        if tuple(batch.shape[2:]) != tuple(eng.resolution[1:]):
            raise ValueError(f'expected {eng.resolution[1]}x{eng.resolution[2]} images, got {tuple(batch.shape[2:])}')
        return _EngineFn.apply(batch, self, eng, want_purified)


class BaseClassificationModel(ABC, _EngineOwner):

    def __init__(self, model_path: str, device: str, mean: tuple = None, std: tuple = None):
        super().__init__()
        if (mean is not None and std is None) or (mean is None and std is not None):
            raise ValueError("to apply Normalization, please specify both mean and std.")
        if mean is not None and (tuple(mean) != (0.5, 0.5, 0.5) or tuple(std) != (0.5, 0.5, 0.5)):
            raise NotImplementedError('only the (0.5, 0.5) normalisation of the reference classifiers is built')
        self.mean = torch.tensor(mean, device=device) if mean is not None else None
        self.std = torch.tensor(std, device=device) if std is not None else None
        self.preprocess = self.mean is not None
        if not self.preprocess:
            raise NotImplementedError('classifier without input normalisation is not built')
        self._init_engines(device)
        self.classifier = self.load_classifier(model_path, device)

    @abstractmethod
    def load_classifier(self, model_path: str, device: str):
        pass

    def set_device(self, device: str):
        if torch.device(device) != torch.device(self.device):
            self._init_engines(device)

    def input_resolution(self) -> int:
        return getattr(self, 'image_size', 64)

    @property
    def supports_class_jacobian(self) -> bool:
        from ...vgg_spec import VggSpec
        return isinstance(self.classifier.spec, VggSpec)

    def _make_engine(self, rows: int, rep: int, with_noise: bool = True, cot_rep: int = 1) -> Engine:
        w = self.classifier
        r = self.input_resolution()
        return Engine(None, None, (3, r, r), w.state_dict, w.spec, rows=rows, rep=rep, alphas=[], device=self.device,
                      store=self._store, cot_rep=cot_rep)

    def class_jacobian(self, batch: torch.Tensor, classes=None):
        self.image_size = batch.shape[-1]
        return self.class_jacobian_rows(batch, 1, classes)

    def forward_rows(self, batch: torch.Tensor, rep: int = 1) -> torch.Tensor:
        return self._run(batch, rep, False)[0]

    def __call__(self, batch: torch.Tensor) -> torch.Tensor:
        """(B,3,H,W) in [0,1] -> un-normalised predictions (B, n_classes)."""
        self.image_size = batch.shape[-1]
        return self.forward_rows(batch, 1)


class MLVGMDefenseModel(ABC, _EngineOwner):

    def __init__(self, classifier: BaseClassificationModel, autoencoder_path: str,
                 interpolation_alphas: tuple, alpha_attenuation: float = 1.0,
                 initial_noise_eps: float = 0.0, apply_gaussian_blur: bool = False, device: str = 'cpu',
                 mean: tuple = None, std: tuple = None):
        super().__init__()
        self.eps = initial_noise_eps
        self.blur_input = apply_gaussian_blur
        self._init_engines(device)
        self.classifier = classifier
        self.classifier.set_device(device)
        if (mean is not None and std is None) or (mean is None and std is not None):
            raise ValueError("to apply Normalization/Denormalization, please specify both mean and std.")
        self.mean = torch.tensor(mean, device=device) if mean is not None else None
        self.std = torch.tensor(std, device=device) if std is not None else None
        self.preprocess = self.mean is not None
        self.postprocess = self.mean is not None
        self.interpolation_alphas = [a * alpha_attenuation for a in interpolation_alphas]
        self.autoencoder = self.load_autoencoder(autoencoder_path, device)
        # BPDA switch (not in the reference, whose attacks are fully white-box): when True, autograd through this defender
        # treats the purifier as the identity (Engine.backward(identity_purifier=True)); see attacks.pgd.bpda
        self.bpda = False

    @abstractmethod
    def load_autoencoder(self, model_path: str, device: str):
        pass

    def _current_alphas(self):
        return [float(a) for a in self.interpolation_alphas]      # alpha learning overwrites the list in place

    def forward_rows(self, batch: torch.Tensor, rep: int = 1, preds_only: bool = True):
        logits, purified = self._run(batch, rep, not preds_only)
        return logits if preds_only else (logits, purified)

    def purify(self, batch: torch.Tensor) -> torch.Tensor:
        """(B,3,H,W) pre-processed images -> purified reconstructions (no input noise / blur: those belong to
        __call__, abstract_models.py:172-181)."""
        return self._run(batch, 1, True, with_noise=False)[1]

    def __call__(self, batch: torch.Tensor, preds_only: bool = True):
        return self.forward_rows(batch, 1, preds_only)

    def class_jacobian(self, batch: torch.Tensor, classes=None):
        return self.class_jacobian_rows(batch, 1, classes)
