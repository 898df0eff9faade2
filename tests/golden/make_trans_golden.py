"""
Golden vectors for the Style-Transformer encoder and defender, produced by IMPORTING THE REFERENCE
(read-only at /root/reference) in the build container.  Only the .npz travels.

    python tests/golden/make_trans_golden.py          (~2 min on 8 cores)

What runs is the reference's own Python:
  * `StyleGan_Trans/models/transformer.py::TransformerDecoderLayer` (importable as-is);
  * `StyleGan_Trans/models/encoders/style_transformer_encoders.py::GradualStyleEncoder`, `models/style_transformer.py::
    StyleTransformer`, `models/stylegan2/model.py::Generator` — these import the STALE package name `src.hl_autoencoders`
    (SURVEY.md §0.3): aliased here to `src.mlvgms_autoencoders`; their `stylegan2.op` package (import-time CUDA JIT, §0.4) is
    pre-seeded with the stub of make_stylegan_full_golden.py (the reference's own upfirdn2d_native / FusedLeakyReLU Python);
  * `src/defenses/ours/models.py::TransStyleGanDefenseModel.__call__ / purify` with `load_autoencoder` overridden to build the
    StyleTransformer in memory (load_TranStyleGan lives in loading_utils, which needs torchvision).
Shims for absent third parties: kornia normalize / denormalize ((x - m) / s, x * s + m) and kornia.geometry.resize =
F.interpolate(bilinear, align_corners=False) — kornia's documented default; this piece is third-party and stays parity-unpinned.
Weights: gen_adversarial_amd's seeded initialisers, load_state_dict(strict=True).  `torch.normal` is patched while purify runs.
"""
import os
import sys

sys.dont_write_bytecode = True     # importing the reference must not leave __pycache__ in /root/reference (read-only tree)
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_stylegan_full_golden as G          # installs the E4E op stub, kornia / loading_utils / Union shims   # noqa: E402

REF = G.REF

# the stale package name of StyleGan_Trans's imports -> the real package
import src.mlvgms_autoencoders as _pkg       # noqa: E402
sys.modules['src.hl_autoencoders'] = _pkg
OP_T = 'src.hl_autoencoders.StyleGan_Trans.models.stylegan2.op'
op = types.ModuleType(OP_T)
op.FusedLeakyReLU, op.fused_leaky_relu, op.upfirdn2d = G.ref_act.FusedLeakyReLU, G.ref_act.fused_leaky_relu, G.upfirdn2d
sys.modules[OP_T] = op


def _resize(x, size, **kw):
    return torch.nn.functional.interpolate(x, size=(size, size) if isinstance(size, int) else size, mode='bilinear', align_corners=False)


sys.modules['kornia.geometry'].resize = _resize

from src.mlvgms_autoencoders.StyleGan_Trans.models.transformer import TransformerDecoderLayer                  # noqa: E402
from src.hl_autoencoders.StyleGan_Trans.models.encoders.style_transformer_encoders import GradualStyleEncoder    # noqa: E402
from src.hl_autoencoders.StyleGan_Trans.models.style_transformer import StyleTransformer                         # noqa: E402
import src.defenses.ours.models as ref_models                                                                   # noqa: E402
ref_models.resize = _resize                  # `from kornia.geometry import resize` was bound when the module was first imported

from gen_adversarial_amd.trans_spec import build_trans_spec, init_trans_state_dict        # noqa: E402
from gen_adversarial_amd.stylegan_spec import build_stylegan_spec, init_stylegan_state_dict   # noqa: E402

ENC_SEED, GEN_SEED, GEN_SIZE = 51, 52, 32


def _np(**kw):
    return {k: (v.detach().numpy() if isinstance(v, torch.Tensor) else v) for k, v in kw.items()}


def golden_layer(out):
    sd = init_trans_state_dict(1, ENC_SEED)
    m = TransformerDecoderLayer(d_model=512, nhead=4, dim_feedforward=1024).eval()
    m.load_state_dict({k[len('transformerlayer_medium.'):]: v for k, v in sd.items() if k.startswith('transformerlayer_medium.')}, strict=True)
    g = torch.Generator().manual_seed(29)
    tgt = torch.randn(16, 2, 512, generator=g).requires_grad_(True)             # (T, B, C): the reference's layout
    mem = torch.randn(40, 2, 512, generator=g).requires_grad_(True)
    y = m(tgt, mem)
    cot = torch.randn(y.shape, generator=g)
    gt, gm = torch.autograd.grad((y * cot).sum(), [tgt, mem])
    tr = lambda t: t.detach().transpose(0, 1).contiguous()                       # stored batch-first          # noqa: E731
    out.update(_np(**{'layer.tgt': tr(tgt), 'layer.mem': tr(mem), 'layer.y': tr(y), 'layer.cot': tr(cot), 'layer.gtgt': tr(gt), 'layer.gmem': tr(gm)}))
    print('decoder layer: y', tuple(y.shape), float(y.abs().max()))


def golden_encoder(out):
    sd = init_trans_state_dict(1, ENC_SEED)
    enc = GradualStyleEncoder(50, 'ir_se', types.SimpleNamespace(input_nc=3)).eval()
    enc.load_state_dict(sd, strict=True)
    g = torch.Generator().manual_seed(31)
    x = torch.randn(2, 3, 48, 64, generator=g).requires_grad_(True)
    q = torch.randn(2, 16, 512, generator=g).requires_grad_(True)
    codes = enc(x, q)
    cot = torch.randn(codes.shape, generator=g)
    gx, gq = torch.autograd.grad((codes * cot).sum(), [x, q])
    out.update(_np(**{'enc.x': x, 'enc.q': q, 'enc.codes': codes, 'enc.cot': cot, 'enc.gx': gx, 'enc.gq': gq}))
    print('encoder: codes', tuple(codes.shape), float(codes.abs().max()), '|gx|', float(gx.abs().max()))


class _MeanClassifier:
    def set_device(self, device):
        pass

    def __call__(self, batch):
        return batch.mean(dim=(2, 3))


def golden_purify(out):
    sd = init_trans_state_dict(1, ENC_SEED)
    gspec = build_stylegan_spec(GEN_SIZE)
    gsd = init_stylegan_state_dict(gspec, GEN_SEED)
    g = torch.Generator().manual_seed(37)
    latent_avg = 0.5 * torch.randn(16, 512, generator=g)
    opts = types.SimpleNamespace(output_size=GEN_SIZE, input_nc=3, start_from_latent_avg=True, learn_in_w=False, device='cpu',
                                 checkpoint_path=None)

    class Defender(ref_models.TransStyleGanDefenseModel):
        def load_autoencoder(self, model_path, device):
            net = StyleTransformer(opts)
            net.encoder.load_state_dict(sd, strict=True)
            res = net.decoder.load_state_dict(gsd, strict=False)
            assert not res.unexpected_keys and all(k.endswith('.kernel') for k in res.missing_keys), res
            net.latent_avg = latent_avg
            return net.eval()

    alphas = [0.05 * (j % 5) for j in range(16)]
    model = Defender(_MeanClassifier(), 'unused', alphas, alpha_attenuation=0.7, initial_noise_eps=0.0, device='cpu')
    x = torch.rand(2, 3, 128, 128, generator=g).requires_grad_(True)
    z = 0.8 * torch.randn(16, 2, 512, generator=g)                                 # models.py:331: torch.normal(0, 0.8, (n_codes, b, d))
    real = torch.normal

    def fed(mean, std, size_, **kw):
        assert tuple(size_) == tuple(z.shape) and mean == 0 and std == 0.8
        return z.clone()
    torch.normal = fed
    try:
        preds, purified = model(x, preds_only=False)
    finally:
        torch.normal = real
    assert purified.shape == (2, 3, 128, 128)
    small = purified[:, :, ::4, ::4]                                                 # Generator(32) -> face_pool 256 -> resize 128: replication x4
    assert torch.allclose(small.repeat_interleave(4, 2).repeat_interleave(4, 3), purified, atol=1e-6)
    assert float(small[:, :, :4].abs().max()) == 0.0 and float(small[:, :, -4:].abs().max()) == 0.0   # the -1 band, de-normalised
    cot = torch.randn(small.shape, generator=g)
    (gx,) = torch.autograd.grad((small * cot).sum(), [x])
    out.update(_np(**{'purify.x': x, 'purify.z': z.permute(1, 0, 2).contiguous(), 'purify.latent_avg': latent_avg,
                      'purify.alphas': np.asarray(model.interpolation_alphas, dtype=np.float64), 'purify.purified32': small,
                      'purify.preds': preds, 'purify.cot': cot, 'purify.gx': gx}))
    print('purify: purified', tuple(purified.shape), float(small.min()), float(small.max()), '|gx|', float(gx.abs().max()))


if __name__ == '__main__':
    torch.set_num_threads(8)
    out = {}
    golden_layer(out)
    golden_encoder(out)
    golden_purify(out)
    np.savez_compressed(os.path.join(HERE, 'trans_full.npz'), enc_seed=ENC_SEED, gen_seed=GEN_SEED, gen_size=GEN_SIZE, **out)
    assert not any(f.endswith('.hip') for _, _, fs in os.walk(REF) for f in fs), 'reference tree was modified'
