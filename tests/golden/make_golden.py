"""
Generates the golden vectors under tests/golden/ by IMPORTING THE REFERENCE (read-only at /root/reference).
Runs only in the build container; the reference never travels, only the .npz fixtures do.

    python tests/golden/make_golden.py

Shims (container-only, nothing from the reference is copied):
  * `kornia`, `kornia.enhance` (Normalize/Denormalize/normalize/denormalize), `kornia.filters`, `kornia.geometry`
    are absent from the image -> tiny stand-ins with the documented arithmetic ((x-mean)/std, x*std+mean);
  * `src.defenses.loading_utils` is replaced by an empty stub so that importing `src.defenses.ours.models` does NOT
    pull in torchvision (absent) nor the StyleGAN `op` package (whose import JIT-compiles CUDA sources and would
    write into the reference tree — SURVEY.md §0.4);
  * `builtins.Union` for the un-imported annotation at src/defenses/ours/abstract_models.py:162.
Weights come from gen_adversarial_amd's seeded initialiser and are loaded into the reference modules with
`load_state_dict(strict=True)`, which also pins our key names and shapes.
Random draws are made explicit: `Normal.sample` is patched to pop pre-drawn eps tensors (same mul/add order as
distributions.py:43-45) and the input-noise draw of abstract_models.py:132 is reproduced by re-seeding.
"""
import builtins
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference'
sys.dont_write_bytecode = True
sys.path.insert(0, REF)        # the reference's `src` package must win over the repo's import-compat `src/` shim
sys.path.insert(1, REPO)


def install_shims():
    k = types.ModuleType('kornia')
    ke = types.ModuleType('kornia.enhance')
    kf = types.ModuleType('kornia.filters')
    kg = types.ModuleType('kornia.geometry')

    def _bc(v, x):
        v = torch.as_tensor(v, dtype=x.dtype, device=x.device)
        return v.view(1, -1, 1, 1) if v.ndim == 1 else v

    def normalize(x, mean, std):
        return (x - _bc(mean, x)) / _bc(std, x)

    def denormalize(x, mean, std):
        return x * _bc(std, x) + _bc(mean, x)

    class Normalize(torch.nn.Module):
        def __init__(self, mean, std):
            super().__init__()
            self.mean, self.std = mean, std

        def forward(self, x):
            return normalize(x, self.mean, self.std)

    class Denormalize(torch.nn.Module):
        def __init__(self, mean, std):
            super().__init__()
            self.mean, self.std = mean, std

        def forward(self, x):
            return denormalize(x, self.mean, self.std)

    def _absent(*a, **kw):
        raise RuntimeError('kornia is absent: this third-party op is not pinned by the goldens')

    ke.Normalize, ke.Denormalize, ke.normalize, ke.denormalize = Normalize, Denormalize, normalize, denormalize
    kf.gaussian_blur2d = _absent
    kg.resize = _absent
    k.enhance, k.filters, k.geometry = ke, kf, kg
    sys.modules.update({'kornia': k, 'kornia.enhance': ke, 'kornia.filters': kf, 'kornia.geometry': kg})

    lu = types.ModuleType('src.defenses.loading_utils')
    for n in ('load_ResNet50', 'load_Vgg11', 'load_ResNext50', 'load_NVAE', 'load_E4EStyleGan', 'load_TranStyleGan'):
        setattr(lu, n, _absent)
    sys.modules['src.defenses.loading_utils'] = lu

    class _U:
        def __class_getitem__(cls, item):
            return cls
    builtins.Union = _U


install_shims()

from src.mlvgms_autoencoders.NVAE.model import AutoEncoder                      # noqa: E402
from src.mlvgms_autoencoders.NVAE.modules import distributions as ref_dist     # noqa: E402
from src.mlvgms_autoencoders.NVAE.modules.architecture import (                # noqa: E402
    ResidualCellEncoder, ResidualCellDecoder, SE, EncCombinerCell, DecCombinerCell)
from src.defenses.ours.models import NVAEDefenseModel                           # noqa: E402
from src.defenses.ours.abstract_models import BaseClassificationModel          # noqa: E402
from src.defenses.wrappers import EoTWrapper                                    # noqa: E402

from gen_adversarial_amd.nvae_spec import build_spec, init_nvae_state_dict     # noqa: E402
from gen_adversarial_amd.vgg_spec import build_vgg_spec, init_vgg_state_dict   # noqa: E402

assert not any(f.endswith('.hip') for _, _, fs in os.walk(REF) for f in fs), 'reference tree was modified'


# ------------------------------------------------------------------------------------------------------------
class EpsFeeder:
    """Replaces Normal.sample: z = eps * sigma + mu with eps popped from a list (distributions.py:37-45)."""

    def __init__(self):
        self.queue = []

    def install(self):
        feeder = self

        def sample(self_normal):
            eps = feeder.queue.pop(0).clone()
            assert eps.shape == self_normal.mu.shape, (eps.shape, self_normal.mu.shape)
            z = eps.mul_(self_normal.sigma).add_(self_normal.mu)
            return z, eps
        ref_dist.Normal.sample = sample


FEED = EpsFeeder()
FEED.install()


class VggLike(torch.nn.Module):
    """torchvision-vgg11_bn-shaped module (torchvision is absent): same state-dict keys as src/classifier/model.py:31-49.
    This stands in for third-party code; the golden therefore pins our restatement of torchvision only against
    this file's own transcription of the published VGG-11-BN configuration."""

    def __init__(self, n_classes, width_div):
        super().__init__()
        spec = build_vgg_spec(n_classes, width_div)
        layers = []
        for op in spec.program:
            if op[0] == 'pool':
                layers.append(torch.nn.MaxPool2d(2, 2))
            else:
                _, i, cin, cout = op
                layers += [torch.nn.Conv2d(cin, cout, 3, padding=1), torch.nn.BatchNorm2d(cout), torch.nn.ReLU(True)]
        m = torch.nn.Module()
        m.features = torch.nn.Sequential(*layers)
        m.avgpool = torch.nn.AdaptiveAvgPool2d((7, 7))
        d = spec.head_dim
        m.classifier = torch.nn.Sequential(torch.nn.Linear(d, d, bias=False), torch.nn.BatchNorm1d(d),
                                           torch.nn.ReLU(True), torch.nn.Linear(d, n_classes))
        self.model = m

    def forward(self, x):
        x = self.model.features(x)
        x = self.model.avgpool(x)
        return self.model.classifier(torch.flatten(x, 1))


def make_defender(cfg, resolution, nvae_seed, vgg_seed, n_classes, width_div, alphas, attenuation, noise_eps):
    ae = AutoEncoder(cfg, resolution)
    ae.load_state_dict(init_nvae_state_dict(cfg, resolution, nvae_seed), strict=True)
    ae.eval()
    vgg = VggLike(n_classes, width_div)
    vgg.load_state_dict(init_vgg_state_dict(n_classes, width_div, vgg_seed), strict=True)
    vgg.eval()

    class Clf(BaseClassificationModel, torch.nn.Module):
        def __init__(self):
            BaseClassificationModel.__init__(self, '', 'cpu', (0.5, 0.5, 0.5), (0.5, 0.5, 0.5))

        def load_classifier(self, model_path, device):
            return vgg

    class Def(NVAEDefenseModel):
        def load_autoencoder(self, model_path, device):
            return ae

    d = Def(Clf(), '', alphas, attenuation, noise_eps, False, 'cpu')
    return d, ae, vgg


CFG_A = {'initial_channels': 8, 'num_pre-post_process_blocks': 2, 'num_pre-post_process_cells': 2, 'num_scales': 2,
         'num_groups_per_scale': 2, 'is_adaptive': False, 'min_groups_per_scale': 1, 'num_cells_per_group': 2,
         'num_latent_per_group': 4, 'num_logistic_mixtures': 10, 'num_nf_cells': None}
RES_A = (3, 32, 32)
CFG_B = {'initial_channels': 4, 'num_pre-post_process_blocks': 1, 'num_pre-post_process_cells': 3, 'num_scales': 3,
         'num_groups_per_scale': 4, 'is_adaptive': True, 'min_groups_per_scale': 1, 'num_cells_per_group': 1,
         'num_latent_per_group': 6, 'num_logistic_mixtures': 10, 'num_nf_cells': None}
RES_B = (3, 32, 32)


def cosine_alphas(n):
    # same shape as the yaml's cosine schedule: monotone 0 -> 1
    return [float(0.5 * (1 - np.cos(np.pi * i / (n - 1)))) for i in range(n)]


def run_case(name, cfg, res, rows, alphas, attenuation, noise_eps, eot):
    spec = build_spec(cfg, res)
    n_classes, width_div = 10, 16
    torch.manual_seed(1234)
    g = torch.Generator().manual_seed(99)
    x = torch.rand((rows,) + tuple(res), generator=g)
    eps = [torch.randn(rows, spec.num_latent, gs.res, gs.res, generator=g) for gs in spec.groups]
    defender, ae, vgg = make_defender(cfg, res, 11, 12, n_classes, width_div, alphas, attenuation, noise_eps)
    out = {'x': x.numpy(), 'alphas': np.asarray(alphas, np.float64), 'attenuation': attenuation,
           'noise_eps': noise_eps, 'cfg_keys': np.asarray(list(cfg.keys())),
           'cfg_vals': np.asarray([repr(v) for v in cfg.values()]), 'res': np.asarray(res),
           'nvae_seed': 11, 'vgg_seed': 12, 'n_classes': n_classes, 'width_div': width_div}
    for i, e in enumerate(eps):
        out[f'eps_{i}'] = e.numpy()
    # the reference state-dict layout we loaded strictly
    sd = ae.state_dict()
    out['sd_keys'] = np.asarray(list(sd.keys()))
    out['sd_shapes'] = np.asarray([repr(tuple(v.shape)) for v in sd.values()])
    vsd = vgg.state_dict()
    out['vgg_keys'] = np.asarray(list(vsd.keys()))
    out['vgg_shapes'] = np.asarray([repr(tuple(v.shape)) for v in vsd.values()])

    # 1. purify alone (models.py:160-274)
    FEED.queue = [e for e in eps]
    with torch.no_grad():
        out['purified_only'] = defender.purify(x).numpy()
    assert not FEED.queue

    # 2. whole defender (abstract_models.py:161-193): input noise reproduced by re-seeding
    torch.manual_seed(777)
    noise = torch.ones_like(x).normal_(0., 1.)
    out['input_noise'] = noise.numpy()
    torch.manual_seed(777)
    FEED.queue = [e for e in eps]
    xg = x.clone().requires_grad_(True)
    logits, purified = defender(xg, preds_only=False)
    out['logits'] = logits.detach().numpy()
    out['purified'] = purified.detach().numpy()
    # 3. input gradient for a fixed cotangent (what every attack consumes, untargeted.py:146,201)
    cot = torch.randn(logits.shape, generator=g)
    out['cotangent'] = cot.numpy()
    (gx,) = torch.autograd.grad((logits * cot).sum(), [xg])
    out['grad_x'] = gx.numpy()

    # 4. EoT wrapper on the first image (wrappers.py:15-24)
    wrapper = EoTWrapper(defender, eot)
    x1 = x[:1].clone().requires_grad_(True)
    eps1 = [torch.randn(eot, spec.num_latent, gs.res, gs.res, generator=g) for gs in spec.groups]
    torch.manual_seed(778)
    noise1 = torch.ones(eot, *res).normal_(0., 1.)
    torch.manual_seed(778)
    FEED.queue = [e for e in eps1]
    pl = wrapper(x1)
    label = pl.argmax(dim=1)
    loss = torch.nn.functional.cross_entropy(pl, label)
    (g1,) = torch.autograd.grad(loss, [x1])
    for i, e in enumerate(eps1):
        out[f'eot_eps_{i}'] = e.numpy()
    out['eot_steps'] = eot
    out['eot_noise'] = noise1.numpy()
    out['eot_logits'] = pl.detach().numpy()
    out['eot_ce_grad'] = g1.numpy()
    np.savez_compressed(os.path.join(HERE, f'nvae_{name}.npz'), **out)
    print(name, 'logits', out['logits'][0, :4], 'purified mean', out['purified'].mean(), '|grad|', np.abs(out['grad_x']).max())


def run_modules():
    """Per-module goldens: reference modules with seeded weights, output and dX for a cotangent."""
    from gen_adversarial_amd.nvae_spec import _Rng, _enc_cell, _dec_cell, EncCellSpec, DecCellSpec
    g = torch.Generator().manual_seed(5)
    out = {}

    def strip(sd, prefix):
        return {k[len(prefix) + 1:]: v for k, v in sd.items() if k.startswith(prefix + '.')}

    cases = [('enc_same', EncCellSpec('c', 8, 8, False), (3, 8, 8, 8)),
             ('enc_down', EncCellSpec('c', 8, 16, True), (3, 8, 8, 8)),
             ('dec_same', DecCellSpec('c', 8, 8, False, 6), (3, 8, 4, 4)),
             ('dec_up', DecCellSpec('c', 8, 4, True, 3), (3, 8, 4, 4))]
    for i, (name, cell, shape) in enumerate(cases):
        sd = {}
        rng = _Rng(100 + i)
        if isinstance(cell, EncCellSpec):
            _enc_cell(sd, rng, cell)
            m = ResidualCellEncoder(cell.cin, cell.cout, cell.down, True)
        else:
            _dec_cell(sd, rng, cell)
            m = ResidualCellDecoder(cell.cin, cell.cout, cell.up, True, cell.hidden_mul)
        m.load_state_dict(strip(sd, 'c'), strict=True)
        m.eval()
        x = torch.randn(shape, generator=g).requires_grad_(True)
        y = m(x)
        cot = torch.randn(y.shape, generator=g)
        (gx,) = torch.autograd.grad((y * cot).sum(), [x])
        out[f'{name}_x'], out[f'{name}_y'] = x.detach().numpy(), y.detach().numpy()
        out[f'{name}_cot'], out[f'{name}_gx'] = cot.numpy(), gx.numpy()
        out[f'{name}_seed'] = 100 + i

    # DiscMixLogistic.mean (distributions.py:103-129, 231-254)
    lg = (torch.randn(2, 100, 6, 6, generator=g) * 1.5).requires_grad_(True)
    y = ref_dist.DiscMixLogistic(lg, img_channels=3, num_bits=8).mean()
    cot = torch.randn(y.shape, generator=g)
    (gl,) = torch.autograd.grad((y * cot).sum(), [lg])
    out['dml_logits'], out['dml_mean'], out['dml_cot'], out['dml_glogits'] = \
        lg.detach().numpy(), y.detach().numpy(), cot.numpy(), gl.numpy()

    # Normal (soft-clamped mu, sigma) + sample_given_eps (distributions.py:32-48)
    mu, ls, e = (torch.randn(2, 4, 3, 3, generator=g) * 4 for _ in range(3))
    n = ref_dist.Normal(mu, ls, temp=0.6)
    out['normal_mu_in'], out['normal_ls_in'], out['normal_eps'] = mu.numpy(), ls.numpy(), e.numpy()
    out['normal_mu'], out['normal_sigma'] = n.mu.numpy(), n.sigma.numpy()
    out['normal_z'] = n.sample_given_eps(e).numpy()
    np.savez_compressed(os.path.join(HERE, 'nvae_modules.npz'), **out)
    print('modules done')


CFG_NF = dict(CFG_A, **{'num_nf_cells': 2})


if __name__ == '__main__':
    torch.set_num_threads(8)
    run_case('A_nf2', CFG_NF, RES_A, rows=2, alphas=cosine_alphas(4), attenuation=0.7, noise_eps=0.0, eot=2)
    run_case('A_cos07', CFG_A, RES_A, rows=3, alphas=cosine_alphas(4), attenuation=0.7, noise_eps=0.0, eot=4)
    run_case('A_zero_noise2', CFG_A, RES_A, rows=2, alphas=[0.0] * 4, attenuation=1.0, noise_eps=2.0, eot=2)
    nB = len(build_spec(CFG_B, RES_B).groups)
    run_case('B_adaptive', CFG_B, RES_B, rows=2, alphas=cosine_alphas(nB), attenuation=0.7, noise_eps=0.5, eot=3)
    run_modules()
    assert not any(f.endswith('.hip') for _, _, fs in os.walk(REF) for f in fs), 'reference tree was modified'
