"""
ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/nvae_oracle.py header for the rules).

CPU restatement of the classifier, the defender wrapper and EoT on the purification path:
  src/classifier/model.py:31-49               Vgg (torchvision vgg11_bn + projector head)
  src/defenses/ours/abstract_models.py:53-62  BaseClassificationModel.__call__
  src/defenses/ours/abstract_models.py:129-193 MLVGMDefenseModel (noise, blur, purify, classify)
  src/defenses/wrappers.py:15-24              EoTWrapper.forward

Third-party arithmetic restated here (absent from the reference tree and from this image, versions unpinned in
environment.yml:10,17): torchvision `vgg11_bn` topology; kornia `normalize` ((x-mean)/std per channel) and
`filters.gaussian_blur2d(x, k, (1,1))` (separable, 'reflect' border, kernel = normalised exp(-(i-(k-1)/2)^2/2)).
These third-party pieces are parity-unpinned by any reference fixture; the golden vectors pin everything
that lives in the reference tree itself.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence

import torch
import torch.nn.functional as F

from oracle import kinks as K     # relu / leaky_relu / prelu / clamp / max_pool2d: the torch functions unless a test flips near-ties

from gen_adversarial_amd.vgg_spec import VggSpec
from gen_adversarial_amd.nvae_spec import NVAESpec
from oracle.nvae_oracle import nvae_purify

SD = Dict[str, torch.Tensor]


def vgg_forward(sd: SD, spec: VggSpec, x: torch.Tensor) -> torch.Tensor:
    """Vgg.forward — src/classifier/model.py:47-49 (torchvision VGG.forward: features, avgpool(7,7), flatten, classifier)."""
    for op in spec.program:
        if op[0] == 'pool':
            x = K.max_pool2d(x, 2, 2)
        else:
            _, i, _, _ = op
            x = F.conv2d(x, sd[f'model.features.{i}.weight'], sd[f'model.features.{i}.bias'], padding=1)
            b = f'model.features.{i + 1}'
            x = F.batch_norm(x, sd[f'{b}.running_mean'], sd[f'{b}.running_var'], sd[f'{b}.weight'], sd[f'{b}.bias'],
                             False, 0.0, 1e-5)
            x = K.relu(x)
    x = F.adaptive_avg_pool2d(x, (7, 7)).flatten(1)
    x = F.linear(x, sd['model.classifier.0.weight'])
    c = 'model.classifier.1'
    x = F.batch_norm(x, sd[f'{c}.running_mean'], sd[f'{c}.running_var'], sd[f'{c}.weight'], sd[f'{c}.bias'],
                     False, 0.0, 1e-5)
    x = K.relu(x)
    return F.linear(x, sd['model.classifier.3.weight'], sd['model.classifier.3.bias'])


def resnet_forward(sd: SD, spec, x: torch.Tensor) -> torch.Tensor:
    """ResNet.forward / ResNext.forward — src/classifier/model.py:10-28, 52-70: torchvision resnet50 / resnext50_32x4d
    (published definition, see gen_adversarial_amd/resnet_spec.py) with the projector head of :19-24, eval mode."""
    def bn(p, t):
        return F.batch_norm(t, sd[f'{p}.running_mean'], sd[f'{p}.running_var'], sd[f'{p}.weight'], sd[f'{p}.bias'],
                            False, 0.0, 1e-5)
    x = K.relu(bn('model.bn1', F.conv2d(x, sd['model.conv1.weight'], stride=2, padding=3)))
    x = K.max_pool2d(x, 3, 2, 1)
    for b in spec.blocks:
        p = b.prefix
        o = K.relu(bn(f'{p}.bn1', F.conv2d(x, sd[f'{p}.conv1.weight'])))
        o = K.relu(bn(f'{p}.bn2', F.conv2d(o, sd[f'{p}.conv2.weight'], stride=b.stride, padding=1, groups=b.groups)))
        o = bn(f'{p}.bn3', F.conv2d(o, sd[f'{p}.conv3.weight']))
        idt = bn(f'{p}.downsample.1', F.conv2d(x, sd[f'{p}.downsample.0.weight'], stride=b.stride)) if b.downsample else x
        x = K.relu(o + idt)
    x = torch.flatten(F.adaptive_avg_pool2d(x, (1, 1)), 1)
    x = F.linear(x, sd['model.fc.0.weight'])
    x = K.relu(F.batch_norm(x, sd['model.fc.1.running_mean'], sd['model.fc.1.running_var'], sd['model.fc.1.weight'],
                            sd['model.fc.1.bias'], False, 0.0, 1e-5))
    return F.linear(x, sd['model.fc.3.weight'], sd['model.fc.3.bias'])


def resnet_classifier_call(sd: SD, spec, batch: torch.Tensor, mean=(0.5, 0.5, 0.5), std=(0.5, 0.5, 0.5)):
    """BaseClassificationModel.__call__ with the ResNet-50 of CelebaGenderClassifier — abstract_models.py:53-62,
    ours/models.py:17-35."""
    m = torch.tensor(mean, dtype=batch.dtype).view(1, 3, 1, 1)
    s = torch.tensor(std, dtype=batch.dtype).view(1, 3, 1, 1)
    return resnet_forward(sd, spec, (batch - m) / s)


def classifier_call(sd: SD, spec: VggSpec, batch: torch.Tensor, mean=(0.5, 0.5, 0.5), std=(0.5, 0.5, 0.5)):
    """BaseClassificationModel.__call__ — abstract_models.py:53-62 with the CelebaIdentityClassifier
    constants (models.py:45-47)."""
    m = torch.tensor(mean).view(1, 3, 1, 1)
    s = torch.tensor(std).view(1, 3, 1, 1)
    return vgg_forward(sd, spec, (batch - m) / s)


def add_gaussian_noise(x: torch.Tensor, noise: torch.Tensor, eps: float) -> torch.Tensor:
    """MLVGMDefenseModel.add_gaussian_noise — abstract_models.py:129-143, with the N(0,1) draw passed in."""
    norm = torch.norm(noise.view(noise.size(0), -1), dim=1, keepdim=True)
    scaled = noise * (eps / norm.view(-1, 1, 1, 1))
    return K.clamp(x + scaled, 0.0, 1.0)


def gaussian_kernel1d(k: int, sigma: float = 1.0) -> torch.Tensor:
    xs = torch.arange(k, dtype=torch.float32) - (k - 1) / 2.0 if k % 2 == 1 else torch.arange(k, dtype=torch.float32) - k // 2 + 0.5
    g = torch.exp(-xs.pow(2) / (2 * sigma * sigma))
    return g / g.sum()


def blur_kernel_size(h: int) -> int:
    """abstract_models.py:150-156: k = int(2**(sqrt(h)//2) - 1)."""
    return int(2 ** (math.sqrt(h) // 2) - 1)


def apply_gaussian_blur(x: torch.Tensor) -> torch.Tensor:
    """MLVGMDefenseModel.apply_gaussian_blur — abstract_models.py:145-159 (kornia gaussian_blur2d, sigma (1,1),
    default border_type='reflect', separable)."""
    b, c, h, w = x.shape
    k = blur_kernel_size(h)
    g = gaussian_kernel1d(k).to(x)
    p = k // 2
    y = F.pad(x, (p, p, p, p), mode='reflect')
    y = F.conv2d(y, g.view(1, 1, 1, k).expand(c, 1, 1, k), groups=c)
    y = F.conv2d(y, g.view(1, 1, k, 1).expand(c, 1, k, 1), groups=c)
    return y


def nvae_defender(nvae_sd: SD, nvae_spec: NVAESpec, vgg_sd: SD, vgg_spec: VggSpec, batch: torch.Tensor,
                  alphas: Sequence[float], eps: List[torch.Tensor], input_noise: torch.Tensor,
                  noise_eps: float = 0.0, blur: bool = False, temperature: float = 0.6):
    """MLVGMDefenseModel.__call__ for NVAEDefenseModel (mean/std None => no extra normalisation) —
    abstract_models.py:161-193.  Returns (logits, purified)."""
    if blur:
        batch = apply_gaussian_blur(batch)
    batch = add_gaussian_noise(batch, input_noise, noise_eps)
    purified = nvae_purify(nvae_sd, nvae_spec, batch, alphas, eps, temperature)
    logits = classifier_call(vgg_sd, vgg_spec, purified)
    return logits, purified


def eot_defender(nvae_sd, nvae_spec, vgg_sd, vgg_spec, image: torch.Tensor, eot_steps: int, alphas, eps,
                 input_noise, noise_eps=0.0, blur=False, temperature=0.6):
    """EoTWrapper.forward — wrappers.py:15-24: repeat the single image eot_steps times, mean logits over rows."""
    x = image.repeat(eot_steps, 1, 1, 1)
    logits, purified = nvae_defender(nvae_sd, nvae_spec, vgg_sd, vgg_spec, x, alphas, eps, input_noise,
                                     noise_eps, blur, temperature)
    return torch.mean(logits, dim=0, keepdim=True), purified


def e4e_purify(esd, espec, gsd, gspec, latent_avg, x01, alphas, z, pool_to: int):
    """MLVGMDefenseModel.__call__ around E4EStyleGanDefenseModel.purify up to the de-normalised purified image
    (abstract_models.py:176-185; src/defenses/ours/models.py:105-132; psp.py:89-118): see e4e_defender_call"""
    from oracle.e4e_oracle import e4e_encode
    from oracle import stylegan_oracle as S
    codes = e4e_encode(esd, espec, (x01 - 0.5) / 0.5)
    if latent_avg is not None:
        codes = codes + latent_avg.unsqueeze(0)
    styles = S.mapping_network(gsd, z)
    a = torch.tensor(list(alphas), dtype=codes.dtype).view(1, -1, 1)
    codes = (1 - a) * codes + a * styles
    img = S.generator_forward(gsd, gspec, codes)
    img = F.adaptive_avg_pool2d(img, (pool_to, pool_to))
    return img * 0.5 + 0.5


def e4e_defender_call(esd, espec, gsd, gspec, latent_avg, csd, cspec, x01, alphas, z, pool_to: int):
    """E4EStyleGanDefenseModel (src/defenses/ours/models.py:80-132) behind MLVGMDefenseModel.__call__ (abstract_models.py:161-193)
    for a batch already repeated / noised / clamped to [0, 1]:
      normalize(0.5, 0.5) -> pSp.encode (encoder + latent_avg, psp.py:89-103) -> codes mixed with mapping(z) per latent index
      (models.py:116-127; z [B, n_latent, D] ~ N(0, 1) supplied by the caller) -> pSp.decode (generator, fixed noise buffers,
      face_pool; psp.py:112-118) -> denormalize -> classifier.  face_pool = AdaptiveAvgPool2d to pool_to (256 in the reference;
      a k x k mean when the generator size is a multiple of it).  Returns (logits, purified image in [~0, ~1]).
    Pinned by tests/golden/e4e_purify.npz (the reference's own E4EStyleGanDefenseModel, tests/golden/make_stylegan_full_golden.py)."""
    purified = e4e_purify(esd, espec, gsd, gspec, latent_avg, x01, alphas, z, pool_to)
    return resnet_classifier_call(csd, cspec, purified), purified


class EoTDefenderOracle(torch.nn.Module):
    """EoTWrapper(NVAEDefenseModel(classifier)) as ONE differentiable CPU callable with every random draw pinned — what the
    reference's attacks see as `net` (src/attacks/untargeted.py: every `net(x)`; src/defenses/wrappers.py:15-24;
    src/defenses/ours/abstract_models.py:161-193), restated on the oracle so that the SAME attack code can be run against the HIP
    defender and against this restatement (tests/test_attack_parity_gpu.py, tools/robust_acc_attack.py).

    forward(x: (B,3,H,W)) -> (B, n_classes): image b becomes defender rows b*eot .. b*eot+eot-1 (the batched form of
    `x.repeat(eot_steps, 1, 1, 1)` for one image), row r of a call uses draw r of `eps[g]` / `input_noise` (as the HIP defender's
    fixed_noise does), logits are averaged over each image's rows."""

    def __init__(self, nvae_sd, nvae_spec, vgg_sd, vgg_spec, eot_steps: int, alphas, eps, input_noise, noise_eps: float = 0.0,
                 blur: bool = False, temperature: float = 0.6):
        super().__init__()
        self.a = (nvae_sd, nvae_spec, vgg_sd, vgg_spec)
        self.eot, self.alphas, self.eps, self.noise = eot_steps, list(alphas), eps, input_noise
        self.noise_eps, self.blur, self.temperature = noise_eps, blur, temperature
        self.calls = 0

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        B = x.shape[0]
        rows = B * self.eot
        self.calls += 1
        noise = self.noise[:rows] if self.noise is not None else torch.ones(rows, *x.shape[1:])
        logits, _ = nvae_defender(*self.a, x.repeat_interleave(self.eot, dim=0), self.alphas, [e[:rows] for e in self.eps], noise,
                                  self.noise_eps, self.blur, self.temperature)
        return logits.view(B, self.eot, -1).mean(dim=1)
