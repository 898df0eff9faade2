"""
Golden vectors for the StyleGAN2 synthesis network and the e4e defender (SURVEY.md §8 rows a14 tail, a16, a17), produced by
IMPORTING THE REFERENCE (read-only at /root/reference) in the build container.  Only the .npz files travel.

    python tests/golden/make_stylegan_full_golden.py            (~2 min on 8 cores)

What runs is the reference's own Python:
  * `op/upfirdn2d.py::upfirdn2d_native` (upfirdn2d.py:150-184) — the module is loaded from its file with
    `torch.utils.cpp_extension.load` patched to a no-op (the real call JIT-compiles the CUDA sources and would write into the
    reference tree) and with the `F` it forgets to import (upfirdn2d.py:157) injected;
  * `op/fused_act.py`'s autograd Functions (`FusedLeakyReLU`, `fused_leaky_relu`) over a Python statement of the ONE switch of
    `fused_bias_act_kernel.cu:34-45` (act 3: grad 0 / 1) in place of the CUDA extension object `fused`;
  * `stylegan2/generator.py` (`StyledConv`, `ToRGB`, `Generator`), `encoding/helpers.py` (`bottleneck_IR_SE`),
    `encoding/encoder.py` (`GradualStyleBlock`, `Encoder4Editing`), `psp.py` (`pSp`) and
    `src/defenses/ours/models.py::E4EStyleGanDefenseModel` (`purify`, `__call__`) on top of them.
The `op` package stub handed to the generator wraps `upfirdn2d_native` exactly as `UpFirDn2d.forward` wraps the CUDA op
(upfirdn2d.py:87-117: reshape to (-1, H, W, 1), call, view back).
Weights: gen_adversarial_amd's seeded initialisers, loaded with load_state_dict(strict=True) (also pins key names / shapes).
Random draws are made explicit: `torch.normal` is patched while `purify` runs so that models.py:119 receives a recorded tensor.
"""
import builtins
import importlib.util
import os
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference'
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
sys.path.insert(1, REPO)

OP_DIR = os.path.join(REF, 'src/mlvgms_autoencoders/StyleGan_E4E/stylegan2/op')
OP_PKG = 'src.mlvgms_autoencoders.StyleGan_E4E.stylegan2.op'


# ---------------------------------------------------------------------------------------------------- shims
class _FusedStub:
    """stands in for the JIT-built extension object `fused` of op/fused_act.py: the arithmetic of
    fused_bias_act_kernel.cu:18-49 for the only (act, grad) pairs the module uses"""

    @staticmethod
    def fused_bias_act(x, b, ref, act, grad, alpha, scale):
        assert act == 3 and grad in (0, 1)
        if b.numel():
            x = x + b.view(1, -1, *([1] * (x.ndim - 2)))                  # p_b[(xi / step_b) % size_b], .cu:28-30
        if grad == 0:
            y = torch.where(x > 0, x, x * alpha)                          # case 30
        else:
            y = torch.where(ref > 0, x, x * alpha)                        # case 31
        return y * scale


def _load_op_module(fname, modname):
    import torch.utils.cpp_extension as cpp
    real = cpp.load
    cpp.load = lambda *a, **k: _FusedStub()            # no compilation, nothing written
    try:
        spec = importlib.util.spec_from_file_location(modname, os.path.join(OP_DIR, fname))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    finally:
        cpp.load = real
    return mod


ref_up = _load_op_module('upfirdn2d.py', '_ref_upfirdn2d')
ref_up.F = torch.nn.functional                          # upfirdn2d.py:157 uses F without importing it
ref_act = _load_op_module('fused_act.py', '_ref_fused_act')


def upfirdn2d(input, kernel, up=1, down=1, pad=(0, 0)):
    """the wrapper of UpFirDn2d.forward (upfirdn2d.py:87-117) around the reference's native statement"""
    _, channel, in_h, in_w = input.shape
    out = ref_up.upfirdn2d_native(input.reshape(-1, in_h, in_w, 1), kernel.to(input.dtype), up, up, down, down,
                                  pad[0], pad[1], pad[0], pad[1])
    return out.reshape(-1, channel, out.shape[1], out.shape[2])


op = types.ModuleType(OP_PKG)
op.FusedLeakyReLU, op.fused_leaky_relu, op.upfirdn2d = ref_act.FusedLeakyReLU, ref_act.fused_leaky_relu, upfirdn2d
sys.modules[OP_PKG] = op


def install_defense_shims():
    """kornia normalize / denormalize stand-ins, loading_utils stub, builtins.Union — as tests/golden/make_golden.py"""
    k, ke, kf, kg = (types.ModuleType(n) for n in ('kornia', 'kornia.enhance', 'kornia.filters', 'kornia.geometry'))

    def _bc(v, x):
        v = torch.as_tensor(v, dtype=x.dtype, device=x.device)
        return v.view(1, -1, 1, 1) if v.ndim == 1 else v

    def _absent(*a, **kw):
        raise RuntimeError('absent third-party op: not pinned by the goldens')

    ke.normalize = lambda x, mean, std: (x - _bc(mean, x)) / _bc(std, x)
    ke.denormalize = lambda x, mean, std: x * _bc(std, x) + _bc(mean, x)
    ke.Normalize = ke.Denormalize = _absent
    kf.gaussian_blur2d, kg.resize = _absent, _absent
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


install_defense_shims()

from src.mlvgms_autoencoders.StyleGan_E4E.stylegan2.generator import Generator, StyledConv, ToRGB, make_kernel   # noqa: E402
from src.mlvgms_autoencoders.StyleGan_E4E.encoding.helpers import bottleneck_IR_SE                                # noqa: E402
from src.mlvgms_autoencoders.StyleGan_E4E.encoding.encoder import GradualStyleBlock                               # noqa: E402
from src.mlvgms_autoencoders.StyleGan_E4E.psp import pSp                                                          # noqa: E402
from src.defenses.ours.models import E4EStyleGanDefenseModel                                                      # noqa: E402

from gen_adversarial_amd.stylegan_spec import (StyledConvSpec, build_stylegan_spec, init_styled_conv_state_dict,  # noqa: E402
                                               init_stylegan_state_dict)
from gen_adversarial_amd.e4e_spec import build_e4e_spec, init_e4e_state_dict                                      # noqa: E402

assert not any(f.endswith('.hip') for _, _, fs in os.walk(REF) for f in fs), 'reference tree was modified'


def _np(**kw):
    return {k: (v.detach().numpy() if isinstance(v, torch.Tensor) else v) for k, v in kw.items()}


# ---------------------------------------------------------------------------------------------------- cases
def golden_ops(out):
    g = torch.Generator().manual_seed(11)
    k4 = make_kernel([1, 3, 3, 1]) * 4
    x = torch.randn(2, 3, 5, 7, generator=g).requires_grad_(True)
    y_up = upfirdn2d(x, k4, up=2, down=1, pad=(2, 1))                       # Upsample of the ToRGB skip (generator.py:29-46)
    cot = torch.randn(y_up.shape, generator=g)
    (g_up,) = torch.autograd.grad((y_up * cot).sum(), [x])
    xb = torch.randn(2, 3, 11, 13, generator=g)
    y_blur = upfirdn2d(xb, k4, pad=(1, 1))                                  # Blur after the transposed conv (generator.py:133-139)
    out.update(_np(**{'up.x': x, 'up.y': y_up, 'up.cot': cot, 'up.gx': g_up, 'blur.x': xb, 'blur.y': y_blur}))
    # fused_leaky_relu through the reference's autograd Functions
    a = torch.randn(3, 6, 4, 4, generator=g).requires_grad_(True)
    b = torch.randn(6, generator=g)
    ya = ref_act.fused_leaky_relu(a, b)
    ca = torch.randn(ya.shape, generator=g)
    (ga,) = torch.autograd.grad((ya * ca).sum(), [a])
    out.update(_np(**{'lrelu.x': a, 'lrelu.b': b, 'lrelu.y': ya, 'lrelu.cot': ca, 'lrelu.gx': ga}))
    print('ops: up', tuple(y_up.shape), 'blur', tuple(y_blur.shape), 'lrelu', tuple(ya.shape))


LAYERS = {'styled': StyledConvSpec('conv1', 32, 64, 3, 64, 8, True, True),
          'styled_up': StyledConvSpec('convs.0', 32, 64, 3, 64, 16, True, True, True),
          'torgb': StyledConvSpec('to_rgbs.0', 64, 3, 1, 64, 16, False, False)}


def golden_layers(out):
    for name, sp in LAYERS.items():
        sd = init_styled_conv_state_dict(sp, 41)
        g = torch.Generator().manual_seed(13)
        r_in = sp.res // 2 if sp.upsample else sp.res
        x = torch.randn(2, sp.cin, r_in, r_in, generator=g).requires_grad_(True)
        w = torch.randn(2, sp.style_dim, generator=g).requires_grad_(True)
        own = {k[len(sp.prefix) + 1:]: v for k, v in sd.items()}
        if sp.activate:
            m = StyledConv(sp.cin, sp.cout, 3, sp.style_dim, upsample=sp.upsample)
            missing = m.load_state_dict(own, strict=False)                       # the constant blur kernel buffer is not ours
            assert all(k.endswith('blur.kernel') for k in missing.missing_keys) and not missing.unexpected_keys, missing
            noise = torch.randn(1, 1, sp.res, sp.res, generator=g)
            y = m(x, w, noise=noise)
            out[f'{name}.noise'] = noise.numpy()
            extra = []
        else:
            m = ToRGB(sp.cin, sp.style_dim)
            missing = m.load_state_dict(own, strict=False)
            assert all(k.endswith('upsample.kernel') for k in missing.missing_keys) and not missing.unexpected_keys, missing
            skip = torch.randn(2, 3, sp.res // 2, sp.res // 2, generator=g).requires_grad_(True)
            y = m(x, w, skip)
            out[f'{name}.skip'] = skip.detach().numpy()
            extra = [skip]
        cot = torch.randn(y.shape, generator=g)
        grads = torch.autograd.grad((y * cot).sum(), [x, w] + extra)
        out.update(_np(**{f'{name}.x': x, f'{name}.w': w, f'{name}.y': y, f'{name}.cot': cot, f'{name}.gx': grads[0], f'{name}.gw': grads[1]}))
        if extra:
            out[f'{name}.gskip'] = grads[2].numpy()
        print(name, 'y', tuple(y.shape), float(y.abs().max()))


GEN_SIZE, GEN_SEED = 32, 43


def _generator(size=GEN_SIZE, seed=GEN_SEED):
    spec = build_stylegan_spec(size)
    sd = init_stylegan_state_dict(spec, seed)
    gen = Generator(size, 512, 8, channel_multiplier=2)
    res = gen.load_state_dict(sd, strict=False)
    assert not res.unexpected_keys and all(k.endswith('.kernel') for k in res.missing_keys), res
    return gen.eval(), spec, sd


def golden_generator(out):
    gen, spec, _ = _generator()
    g = torch.Generator().manual_seed(17)
    lat = (0.7 * torch.randn(2, spec.n_latent, 512, generator=g)).requires_grad_(True)
    img, _ = gen([lat], input_is_latent=True, randomize_noise=False)
    cot = torch.randn(img.shape, generator=g)
    (glat,) = torch.autograd.grad((img * cot).sum(), [lat])
    z = torch.randn(5, 512, generator=g)
    styles = gen.style(z)
    out.update(_np(**{'gen.latent': lat, 'gen.image': img, 'gen.cot': cot, 'gen.glatent': glat, 'map.z': z, 'map.styles': styles}))
    print('generator: image', tuple(img.shape), float(img.abs().max()), '|glat|', float(glat.abs().max()), 'styles', float(styles.abs().max()))


def golden_encoder_blocks(out):
    g = torch.Generator().manual_seed(19)
    for name, (cin, depth, stride) in {'irse_same': (16, 16, 2), 'irse_proj': (16, 32, 2), 'irse_s1': (32, 32, 1)}.items():
        m = bottleneck_IR_SE(cin, depth, stride).eval()
        with torch.no_grad():
            for p in m.parameters():
                p.copy_(0.3 * torch.randn(p.shape, generator=g))
            for n_, b in m.named_buffers():
                if n_.endswith('running_var'):
                    b.copy_(0.5 + torch.rand(b.shape, generator=g))
                elif n_.endswith('running_mean'):
                    b.copy_(0.1 * torch.randn(b.shape, generator=g))
        x = torch.randn(2, cin, 8, 8, generator=g).requires_grad_(True)
        y = m(x)
        cot = torch.randn(y.shape, generator=g)
        (gx,) = torch.autograd.grad((y * cot).sum(), [x])
        out.update(_np(**{f'{name}.x': x, f'{name}.y': y, f'{name}.cot': cot, f'{name}.gx': gx}))
        for k, v in m.state_dict().items():
            if not k.endswith('num_batches_tracked'):
                out[f'{name}.sd.{k}'] = v.numpy()
    m = GradualStyleBlock(32, 32, 8).eval()                # 3 stride-2 convs + LeakyReLU, then EqualLinear
    with torch.no_grad():
        for p in m.parameters():
            p.copy_(0.3 * torch.randn(p.shape, generator=g))
    x = torch.randn(2, 32, 8, 8, generator=g).requires_grad_(True)
    y = m(x)
    cot = torch.randn(y.shape, generator=g)
    (gx,) = torch.autograd.grad((y * cot).sum(), [x])
    out.update(_np(**{'gsb.x': x, 'gsb.y': y, 'gsb.cot': cot, 'gsb.gx': gx}))
    for k, v in m.state_dict().items():
        out[f'gsb.sd.{k}'] = v.numpy()
    print('encoder blocks done')


class _MeanClassifier:
    """stand-in for the BaseClassificationModel argument (torchvision is absent): logits = per-channel mean of the purified
    image.  Only the purifier is pinned by this golden."""

    def set_device(self, device):
        pass

    def __call__(self, batch):
        return batch.mean(dim=(2, 3))


PURIFY = dict(size=32, res=64, enc_seed=47, gen_seed=48, rows=2)


def golden_purify(out):
    size, res = PURIFY['size'], PURIFY['res']
    espec = build_e4e_spec(size)
    esd = init_e4e_state_dict(size, 1, PURIFY['enc_seed'])
    gspec = build_stylegan_spec(size)
    gsd = init_stylegan_state_dict(gspec, PURIFY['gen_seed'])
    g = torch.Generator().manual_seed(23)
    latent_avg = 0.5 * torch.randn(gspec.n_latent, 512, generator=g)
    ref_gen_sd = Generator(size, 512, 8).state_dict()
    dec = {k: gsd.get(k, v) for k, v in ref_gen_sd.items()}               # + the constant kernel buffers strict=True asks for
    ck = {'state_dict': {**{'encoder.' + k: v for k, v in esd.items()}, **{'decoder.' + k: v for k, v in dec.items()}},
          'latent_avg': latent_avg}
    tmp = tempfile.NamedTemporaryFile(suffix='.pt', delete=False)
    tmp.close()
    torch.save(ck, tmp.name)
    opts = types.SimpleNamespace(stylegan_size=size, encoder_type='Encoder4Editing', checkpoint_path=tmp.name,
                                 start_from_latent_avg=True, device='cpu')

    class Defender(E4EStyleGanDefenseModel):
        def load_autoencoder(self, model_path, device):                   # loading_utils.load_E4EStyleGan needs torchvision-free imports
            return pSp(opts).to(device).eval()

    alphas = [0.05 * (j % 5) for j in range(gspec.n_latent)]
    model = Defender(_MeanClassifier(), tmp.name, alphas, alpha_attenuation=0.8, initial_noise_eps=0.0, device='cpu')
    os.unlink(tmp.name)
    rows = PURIFY['rows']
    x = torch.rand(rows, 3, res, res, generator=g).requires_grad_(True)
    z = torch.randn(gspec.n_latent, rows, 512, generator=g)              # models.py:119 draws (n_codes, b, d)
    real_normal = torch.normal

    def fed_normal(mean, std, size_, **kw):
        assert tuple(size_) == tuple(z.shape) and mean == 0 and std == 1
        return z.clone()
    torch.normal = fed_normal
    try:
        preds, purified = model(x, preds_only=False)
    finally:
        torch.normal = real_normal
    assert purified.shape == (rows, 3, 256, 256)                           # face_pool: AdaptiveAvgPool2d((256, 256)), psp.py:26
    small = purified[:, :, ::8, ::8]
    assert torch.equal(small.repeat_interleave(8, 2).repeat_interleave(8, 3), purified), 'face_pool of a 32 px image is a replication'
    cot = torch.randn(small.shape, generator=g)
    (gx,) = torch.autograd.grad((small * cot).sum(), [x])
    codes = model.autoencoder.encode((x.detach() - 0.5) / 0.5)
    out.update(_np(**{'purify.x': x, 'purify.z': z.permute(1, 0, 2).contiguous(), 'purify.latent_avg': latent_avg,
                      'purify.alphas': np.asarray(model.interpolation_alphas, dtype=np.float64), 'purify.codes': codes,
                      'purify.purified32': small, 'purify.preds': preds, 'purify.cot': cot, 'purify.gx': gx}))
    print('purify: purified', tuple(purified.shape), float(small.min()), float(small.max()), '|gx|', float(gx.abs().max()))


if __name__ == '__main__':
    torch.set_num_threads(8)
    out = {}
    golden_ops(out)
    golden_layers(out)
    golden_generator(out)
    golden_encoder_blocks(out)
    np.savez_compressed(os.path.join(HERE, 'stylegan_full.npz'), gen_size=GEN_SIZE, gen_seed=GEN_SEED, **out)
    out2 = {}
    golden_purify(out2)
    np.savez_compressed(os.path.join(HERE, 'e4e_purify.npz'), **{k: v for k, v in PURIFY.items()}, **out2)
    assert not any(f.endswith('.hip') for _, _, fs in os.walk(REF) for f in fs), 'reference tree was modified'
