"""
GPU parity of the Style-Transformer defender (BASELINE.json configs[4]):
  * the new ops through the C-ABI against plain torch: ga_attn (16 queries x Tk keys, forward + backward), ga_layernorm,
    ga_resize2_crop (bilinear x2 + crop and its adjoint), ga_pool_denorm's -1 band;
  * one TransformerDecoderLayer, the whole GradualStyleEncoder and TransStyleGanDefenseModel.__call__ against goldens produced by
    the REFERENCE's own modules (tests/golden/make_trans_golden.py), forward at 1e-3 and input gradients by decision replay;
  * the drop-in API: experiment 'cars', defense_type 'ours' through load(args) on a reduced checkpoint, against the oracle.
"""
import math
import os
from argparse import Namespace

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

if not torch.cuda.is_available():
    pytest.skip('needs a GPU', allow_module_level=True)

from gen_adversarial_amd import _lib as L   # noqa: E402
from gen_adversarial_amd.engine import Engine   # noqa: E402
from gen_adversarial_amd.engine_core import Act   # noqa: E402
from gen_adversarial_amd.engine_trans import TokenView   # noqa: E402
from gen_adversarial_amd.trans_spec import build_trans_spec, init_trans_state_dict   # noqa: E402
from gen_adversarial_amd.stylegan_spec import build_stylegan_spec, init_stylegan_state_dict   # noqa: E402
from gen_adversarial_amd.resnet_spec import build_resnet_spec, init_resnet_state_dict   # noqa: E402

DEV = 'cuda:0'


def close(got, ref, tol, what):
    e, s = (got.cpu() - ref).abs().max().item(), max(1.0, ref.abs().max().item())
    print(f'   {what}: err {e:.2e} of {s:.2e}')
    assert e < tol * s, what


import contextlib   # noqa: E402


@contextlib.contextmanager
def on_stream(side):
    """side: run the body on a fresh non-default HIP stream (its handle is a real 64-bit pointer, unlike the NULL stream's 0:
    the direct per-op binding must pass it untruncated)"""
    torch.cuda.synchronize()
    if not side:
        yield
        return
    s = torch.cuda.Stream(device=DEV)
    assert s.cuda_stream != 0
    with torch.cuda.stream(s):
        yield
    s.synchronize()


def golden():
    z = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'trans_full.npz'))
    return {k: z[k] for k in z.files}


@pytest.mark.parametrize('Tk,dh,side', [(16, 128, False), (192, 128, True), (3072, 128, False), (50, 32, True)])
def test_attention_core_forward_and_backward(Tk, dh, side):
    with on_stream(side):
        _attention_core(Tk, dh)


def _attention_core(Tk, dh):
    N, heads, Tq = 3, 4, 16
    E = heads * dh
    g = torch.Generator().manual_seed(1)
    q = torch.randn(N, Tq, E, generator=g).requires_grad_(True)
    kv = torch.randn(N, Tk, 2 * E, generator=g).requires_grad_(True)          # k and v as the halves of one projection output
    k, v = kv[..., :E], kv[..., E:]

    def hd(t):
        return t.reshape(N, t.shape[1], heads, dh).transpose(1, 2)
    att = torch.softmax(hd(q) @ hd(k).transpose(-1, -2) / math.sqrt(dh), dim=-1)
    ref = (att @ hd(v)).transpose(1, 2).reshape(N, Tq, E)
    cot = torch.randn(ref.shape, generator=g)
    gq, gkv = torch.autograd.grad((ref * cot).sum(), [q, kv])
    qd, kvd, cd = q.detach().to(DEV), kv.detach().to(DEV), cot.to(DEV)
    out, p = torch.zeros(N, Tq, E, device=DEV), torch.zeros(N, heads, Tq, Tk, device=DEV)
    d = L.AttnDesc()
    d.q, d.k, d.v, d.out, d.p = qd.data_ptr(), kvd.data_ptr(), kvd.data_ptr() + 4 * E, out.data_ptr(), p.data_ptr()
    d.ldq, d.ldk, d.ldv, d.ldo = E, 2 * E, 2 * E, E
    d.N, d.Tq, d.Tk, d.heads, d.dh, d.scale, d.backward = N, Tq, Tk, heads, dh, 1.0 / math.sqrt(dh), 0
    L.run(d, torch.cuda.current_stream().cuda_stream)
    close(out, ref.detach(), 1e-5, f'attention forward Tk={Tk}')
    close(p, att.detach(), 1e-5, 'probabilities')
    dq, dkv, ds = torch.zeros_like(qd), torch.zeros_like(kvd), torch.zeros_like(p)
    d.dout, d.ds, d.dq, d.dk, d.dv = cd.data_ptr(), ds.data_ptr(), dq.data_ptr(), dkv.data_ptr(), dkv.data_ptr() + 4 * E
    d.lddq, d.lddk, d.lddv, d.backward = E, 2 * E, 2 * E, 1
    L.run(d, torch.cuda.current_stream().cuda_stream)
    close(dq, gq, 2e-5, 'dq')
    close(dkv, gkv, 2e-5, 'dk | dv')


@pytest.mark.parametrize('side', [False, True])
def test_layernorm_forward_and_backward(side):
    with on_stream(side):
        _layernorm()


def _layernorm():
    rows, C = 37, 512
    g = torch.Generator().manual_seed(2)
    a = torch.randn(rows, C, generator=g).requires_grad_(True)
    b = torch.randn(rows, C, generator=g).requires_grad_(True)
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
    ref = F.layer_norm(a + b, (C,), gamma, beta, 1e-5)
    cot = torch.randn(ref.shape, generator=g)
    ga, gb = torch.autograd.grad((ref * cot).sum(), [a, b])
    ad, bd, gd, bed, cd = (t.detach().to(DEV) for t in (a, b, gamma, beta, cot))
    y, stats, dx = torch.zeros(rows, C, device=DEV), torch.zeros(rows, 2, device=DEV), torch.ones(rows, C, device=DEV)
    d = L.LayernormDesc()
    d.a, d.b, d.gamma, d.beta, d.y, d.stats = ad.data_ptr(), bd.data_ptr(), gd.data_ptr(), bed.data_ptr(), y.data_ptr(), stats.data_ptr()
    d.rows, d.C, d.eps, d.backward = rows, C, 1e-5, 0
    L.run(d, torch.cuda.current_stream().cuda_stream)
    close(y, ref.detach(), 1e-5, 'layernorm forward')
    d.dy, d.dx, d.backward, d.accumulate = cd.data_ptr(), dx.data_ptr(), 1, 1
    L.run(d, torch.cuda.current_stream().cuda_stream)
    close(dx - 1.0, ga, 1e-5, 'layernorm backward (accumulating)')
    assert torch.allclose(ga, gb)


@pytest.mark.parametrize('H,W,crop,side', [(128, 128, 32, True), (16, 24, 4, False), (8, 8, 0, False)])
def test_resize2_crop_forward_and_adjoint(H, W, crop, side):
    with on_stream(side):
        _resize2_crop(H, W, crop)


def _resize2_crop(H, W, crop):
    N, C = 2, 8
    g = torch.Generator().manual_seed(3)
    x = torch.randn(N, C, H, W, generator=g).requires_grad_(True)
    up = F.interpolate(x, size=(2 * H, 2 * W), mode='bilinear', align_corners=False)
    ref = up[:, :, crop:2 * H - crop] if crop else up
    cot = torch.randn(ref.shape, generator=g)
    (gx,) = torch.autograd.grad((ref * cot).sum(), [x])
    xd = x.detach().permute(0, 2, 3, 1).contiguous().to(DEV)
    y = torch.zeros(N, 2 * H - 2 * crop, 2 * W, C, device=DEV)
    d = L.Resize2CropDesc()
    d.x, d.y, d.N, d.H, d.W, d.C, d.crop, d.backward = xd.data_ptr(), y.data_ptr(), N, H, W, C, crop, 0
    L.run(d, torch.cuda.current_stream().cuda_stream)
    close(y.permute(0, 3, 1, 2), ref.detach(), 1e-6, 'resize x2 + crop')
    cd = cot.permute(0, 2, 3, 1).contiguous().to(DEV)
    dx = torch.zeros_like(xd)
    d.dy, d.dx, d.backward, d.accumulate = cd.data_ptr(), dx.data_ptr(), 1, 0
    L.run(d, torch.cuda.current_stream().cuda_stream)
    close(dx.permute(0, 3, 1, 2), gx, 1e-6, 'adjoint')


def _layer_engine(sd, spec, rows, Tm, precision):
    eng = Engine.bare(rows, device=DEV, precision=precision)
    tgt = Act(eng, rows, spec.n_query, 1, spec.d_model, 'tgt')
    mem = Act(eng, rows, Tm, 1, spec.d_model, 'mem')
    out = eng._decoder_layer(sd, spec, 'transformerlayer_medium', tgt, mem, tgt_needs_grad=True)
    eng.finish()
    return eng, tgt, mem, out


@pytest.mark.parametrize('precision,tol', [('fp32', 1e-4), ('bf16x3', 1e-3)])
def test_decoder_layer_matches_the_reference_golden(precision, tol):
    from gradcheck import assert_grad_given_engine_decisions
    from oracle import trans_oracle as T
    g = golden()
    spec, sd = build_trans_spec(1), init_trans_state_dict(1, int(g['enc_seed']))
    tgt, mem, cot = (torch.from_numpy(g[f'layer.{k}']) for k in ('tgt', 'mem', 'cot'))
    rows, Tm = tgt.shape[0], mem.shape[1]
    eng, at, am, out = _layer_engine(sd, spec, rows, Tm, precision)
    at.t.view(rows, 16, -1).copy_(tgt.to(DEV))
    am.t.view(rows, Tm, -1).copy_(mem.to(DEV))
    eng.forward()
    print(f'TransformerDecoderLayer vs reference golden [{precision}]')
    close(out.t.view(rows, 16, -1), torch.from_numpy(g['layer.y']), tol, 'y')
    out.g.view(rows, 16, -1).copy_(cot.to(DEV))
    eng.bwd.run(eng.stream())
    torch.cuda.synchronize()
    for act, key, arg, other in ((at, 'gtgt', tgt, mem), (am, 'gmem', mem, tgt)):
        ref = torch.from_numpy(g[f'layer.{key}'])
        fn = (lambda t: (T.decoder_layer(sd, 'transformerlayer_medium', t, other, spec.nhead) * cot).sum()) if key == 'gtgt' else \
             (lambda t: (T.decoder_layer(sd, 'transformerlayer_medium', other, t, spec.nhead) * cot).sum())
        ar = arg.clone().requires_grad_(True)
        (g0,) = torch.autograd.grad(fn(ar), [ar])
        got = act.g.view(rows, arg.shape[1], -1)
        # the FFN's ReLU input is stored as [rows, 16, 1, dff]: candidates are matched as 4-D NCHW tensors
        assert_grad_given_engine_decisions(eng, fn, arg, got, tol, f'layer.{key}', min_matched=0, golden=(ref, g0))


@pytest.mark.parametrize('precision,tol', [('fp32', 2e-4), ('bf16x3', 1e-3)])
def test_style_transformer_encoder_matches_the_reference_golden(precision, tol):
    """the reference's GradualStyleEncoder (full-width IR-SE50 + FPN + 3 decoder layers) on 48 x 64 inputs: codes and d/dx"""
    from gradcheck import assert_grad_given_engine_decisions
    from oracle import trans_oracle as T
    g = golden()
    spec, sd = build_trans_spec(1), init_trans_state_dict(1, int(g['enc_seed']))
    x, q, cot = (torch.from_numpy(g[f'enc.{k}']) for k in ('x', 'q', 'cot'))
    rows = x.shape[0]
    eng = Engine.bare(rows, device=DEV, precision=precision)
    img = Act(eng, rows, x.shape[2], x.shape[3], 8, 'img')
    es = spec.trunk
    xin = eng._e4e_input_layer(sd, es, img, normalize=False)
    c3, p2, p1 = eng._e4e_body_fpn(sd, es, xin)
    tgt = Act(eng, rows, 16, 1, spec.d_model, 'queries')
    cur = tgt
    from gen_adversarial_amd.trans_spec import LAYERS
    for i, (name, mem) in enumerate(zip(LAYERS, (c3, p2, p1))):
        cur = eng._decoder_layer(sd, spec, name, cur, TokenView(mem), tgt_needs_grad=True)
    eng.finish()
    img.t.zero_()
    img.t[..., :3].copy_(x.permute(0, 2, 3, 1).to(DEV))
    tgt.t.view(rows, 16, -1).copy_(q.to(DEV))
    eng.forward()
    print(f'GradualStyleEncoder vs reference golden [{precision}]: {len(eng.fwd)} + {len(eng.bwd)} ops')
    close(cur.t.view(rows, 16, -1), torch.from_numpy(g['enc.codes']), tol, 'codes')
    cur.g.view(rows, 16, -1).copy_(cot.to(DEV))
    eng.bwd.run(eng.stream())
    torch.cuda.synchronize()
    ref = torch.from_numpy(g['enc.gx'])
    xr = x.clone().requires_grad_(True)
    (g0,) = torch.autograd.grad((T.encode(sd, spec, xr, q) * cot).sum(), [xr])
    got = img.g[..., :3].permute(0, 3, 1, 2)
    assert_grad_given_engine_decisions(eng, lambda t: (T.encode(sd, spec, t, q) * cot).sum(), x, got, 1e-3, 'encoder d/dx', min_matched=20,
                                       golden=(ref, g0))
    close(tgt.g.view(rows, 16, -1), torch.from_numpy(g['enc.gq']), 5e-3, 'd/dquery (near-tie flips included)')


def _defense_engine(rows, rep, tsd, tspec, gsd, gspec, avg, csd, cspec, alphas, res, precision, share=False):
    eng = Engine.bare(rows, device=DEV, precision=precision, rep=rep, resolution=(3, res, res), alphas=alphas, share_encoder=share)
    return eng.build_trans_defense(tsd, tspec, gsd, gspec, avg, csd, cspec, pool_to=min(res, gspec.size), mid=2 * res, crop=res // 4)


@pytest.mark.parametrize('precision,tol', [('fp32', 2e-4), ('bf16x3', 1e-3)])
def test_trans_defender_matches_the_reference_purify_golden(precision, tol):
    """tests/golden/trans_full.npz: the REFERENCE's TransStyleGanDefenseModel.__call__(x, preds_only=False) (128-px inputs, resize
    256, crop, full-width GradualStyleEncoder, latent_avg, alphas x attenuation, recorded N(0, 0.8) draw, Generator(32), face_pool,
    -1 band, resize 128): purified image and the input gradient through it"""
    from gradcheck import assert_grad_given_engine_decisions
    from oracle import trans_oracle as T
    g = golden()
    tspec, tsd = build_trans_spec(1), init_trans_state_dict(1, int(g['enc_seed']))
    gspec = build_stylegan_spec(int(g['gen_size']))
    gsd = init_stylegan_state_dict(gspec, int(g['gen_seed']))
    cspec, csd = build_resnet_spec(4, 2, (1, 1, 1, 1), 4, 8), init_resnet_state_dict(4, 2, 3, (1, 1, 1, 1), 4, 8)     # not compared
    x, z, avg, cot = (torch.from_numpy(g[f'purify.{k}']) for k in ('x', 'z', 'latent_avg', 'cot'))
    alphas = [float(a) for a in g['purify.alphas']]
    rows = x.shape[0]
    eng = _defense_engine(rows, 1, tsd, tspec, gsd, gspec, avg, csd, cspec, alphas, 128, precision)
    eng.x_in.copy_(x.to(DEV))
    eng.eps[0].copy_(z.to(DEV))
    eng.forward()
    ref = torch.from_numpy(g['purify.purified32'])
    got = eng.purified_nchw().cpu()
    print(f'trans defender vs reference purify golden [{precision}]: {len(eng.fwd)} + {len(eng.bwd)} ops')
    close(got, ref, tol, 'purified')
    assert got[:, :, :4].abs().max().item() == 0.0 and got[:, :, -4:].abs().max().item() == 0.0        # the -1 band, de-normalised
    eng.dpurified.copy_(cot.to(DEV))
    eng.backward(from_logits=False, from_purified=True)
    gx = torch.from_numpy(g['purify.gx'])

    def loss(t):
        return (T.trans_purify(tsd, tspec, gsd, gspec, avg, t, alphas, z)[:, :, ::4, ::4] * cot).sum()
    xr = x.clone().requires_grad_(True)
    (g0,) = torch.autograd.grad(loss(xr), [xr])
    assert_grad_given_engine_decisions(eng, loss, x, eng.dx, 1e-3, 'input gradient through the returned purified image', min_matched=20)
    rel = ((eng.dx.cpu() - gx).double().norm() / gx.double().norm()).item()
    print(f'   vs the golden gradient: relL2 {rel:.2e} (near-tie flips between nn.MultiheadAttention and its restatement included)')
    assert rel < 1e-2
    del g0


def _small_case():
    tspec, tsd = build_trans_spec(4, (1, 1, 1, 1)), init_trans_state_dict(4, 1, (1, 1, 1, 1))
    gspec = build_stylegan_spec(64, width_div=8, style_dim=tspec.d_model)
    gsd = init_stylegan_state_dict(gspec, 2)
    cspec, csd = build_resnet_spec(4, 2, (1, 1, 1, 1), 4, 8), init_resnet_state_dict(4, 2, 3, (1, 1, 1, 1), 4, 8)
    avg = 0.3 * torch.randn(16, tspec.d_model, generator=torch.Generator().manual_seed(4))
    alphas = [0.05 * (j % 5) for j in range(16)]
    return tspec, tsd, gspec, gsd, cspec, csd, avg, alphas


@pytest.mark.parametrize('share', [False, True])
def test_reduced_trans_defender_matches_oracle(share):
    """quarter-width encoder, 64-px generator (12 of the 16 codes read), 32-px inputs, ResNeXt classifier: logits, purified
    image, input gradient from the logits; EoT replicas sharing the encoder pass"""
    from gradcheck import assert_grad_given_engine_decisions
    from oracle import defender_oracle as D, trans_oracle as T
    tspec, tsd, gspec, gsd, cspec, csd, avg, alphas = _small_case()
    rows, rep, res = 4, 2, 32
    gen = torch.Generator().manual_seed(5)
    x = torch.rand(rows // rep, 3, res, res, generator=gen)
    z = 0.8 * torch.randn(rows, 16, tspec.d_model, generator=gen)

    def call(t):
        p = T.trans_purify(tsd, tspec, gsd, gspec, avg, t.repeat_interleave(rep, dim=0), alphas, z, out_size=res, mid=2 * res, crop=res // 4,
                           pool_to=2 * res)
        return D.resnet_classifier_call(csd, cspec, p), p
    logits, purified = call(x)
    cot = torch.randn(logits.shape, generator=gen)
    eng = _defense_engine(rows, rep, tsd, tspec, gsd, gspec, avg, csd, cspec, alphas, res, 'bf16x3', share)
    assert eng.enc_rows == (rows // rep if share else rows)
    eng.x_in.copy_(x.to(DEV))
    eng.eps[0].copy_(z.to(DEV))
    eng.forward()
    close(eng.purified_nchw(), purified.detach(), 1e-3, 'purified')
    close(eng.logits.view(rows, -1), logits.detach(), 1e-3, 'logits')
    eng.dlogits.view(rows, -1).copy_(cot.to(DEV))
    eng.backward()
    assert_grad_given_engine_decisions(eng, lambda t: (call(t)[0] * cot).sum(), x, eng.dx, 1e-3, f'trans defender input gradient (share={share})',
                                       min_matched=20)


def test_trans_defender_through_the_reference_api(tmp_path):
    """experiment 'cars', defense_type 'ours' through load(args) (src/experiments/load_defense.py:59-73,132-142): checkpoint
    layout of StyleTransformer.load_weights ('encoder.module.*', 'decoder.module.*', 'latent_avg', 'opts'), EoT wrapper, autograd to
    the input, get_purified, Gaussian blur of the input (configs/ours_*_blur_cars.yaml)"""
    import yaml
    from gen_adversarial_amd.experiments.load_defense import load
    from oracle import defender_oracle as D, trans_oracle as T
    tspec, tsd, gspec, gsd, cspec, csd, avg, alphas = _small_case()
    ck = {'state_dict': {**{'encoder.module.' + k: v for k, v in tsd.items()}, **{'decoder.module.' + k: v for k, v in gsd.items()}},
          'latent_avg': avg, 'opts': {'output_size': gspec.size, 'input_nc': 3, 'start_from_latent_avg': True, 'learn_in_w': False}}
    torch.save(ck, tmp_path / 'trans.pt')
    torch.save({'state_dict': csd}, tmp_path / 'resnext.pt')
    with open(tmp_path / 'cfg.yaml', 'w') as f:
        yaml.safe_dump({'classifier_path': str(tmp_path / 'resnext.pt'), 'autoencoder_path': str(tmp_path / 'trans.pt'),
                        'interpolation_alphas': [a / 0.5 for a in alphas], 'alpha_attenuation': 0.5, 'initial_noise_eps': 0.0,
                        'gaussian_blur_input': True}, f)
    eot, res = 3, 64
    args, model = load(Namespace(config=str(tmp_path / 'cfg.yaml'), experiment='cars', defense_type='ours', eot_steps=eot, device=DEV))
    assert args.image_size == 128
    gen = torch.Generator().manual_seed(6)
    x = torch.rand(1, 3, res, res, generator=gen)
    z = 0.8 * torch.randn(eot, 16, tspec.d_model, generator=gen)

    def call(t):
        p = T.trans_purify(tsd, tspec, gsd, gspec, avg, D.apply_gaussian_blur(t).repeat(eot, 1, 1, 1), alphas, z, out_size=res, mid=2 * res,
                           crop=res // 4, pool_to=2 * res)
        return D.resnet_classifier_call(csd, cspec, p).mean(dim=0, keepdim=True), p
    mean, purified = call(x)
    model.model.fixed_noise([z.to(DEV)], None)
    xd = x.to(DEV).requires_grad_(True)
    out = model(xd)
    assert out.shape == (1, 4)
    close(out.detach(), mean.detach(), 1e-3, 'EoT logits through load(args)')
    (gd,) = torch.autograd.grad(out[0, 1], [xd])
    from gradcheck import assert_grad_given_engine_decisions
    assert_grad_given_engine_decisions(model.model._engine(eot, eot), lambda t: call(t)[0][0, 1], x, gd, 1e-3, 'API input gradient', min_matched=20)
    model.model.fixed_noise([z[:1].to(DEV)], None)
    p = model.get_purified(x.to(DEV))
    close(p, purified[:1].detach(), 1e-3, 'get_purified')
    model.model.fixed_noise(None, None)
    a = model(x.to(DEV))
    assert torch.isfinite(a).all()
    eng = model.model._engine(eot, eot)
    assert abs(eng.eps[0].std().item() - 0.8) < 0.05                      # fresh draws are N(0, 0.8) (models.py:331)
