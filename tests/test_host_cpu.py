"""CPU: host logic — weight folding against the oracle, plan construction (dry run), FLOP accounting against SURVEY.md,
the PGD step and the product path's refusal to run without a GPU."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import golden_cfg
from gen_adversarial_amd import _lib as L
from gen_adversarial_amd import folding as FO
from gen_adversarial_amd.attacks.pgd import PGDLinf
from gen_adversarial_amd.engine import Engine, WeightStore
from gen_adversarial_amd.nvae_spec import (ASSUMED_NVAE_CONFIG, ASSUMED_NVAE_RESOLUTION, DecCellSpec, EncCellSpec, _Rng,
                                           _dec_cell, _enc_cell, build_spec, init_nvae_state_dict)
from gen_adversarial_amd.vgg_spec import adaptive_avgpool_matrix, build_vgg_spec, init_vgg_state_dict
from oracle import defender_oracle as D
from oracle import nvae_oracle as O


def _conv_from_layout(x, w_flat, b, cout, cin, k, stride=1):
    w = w_flat.reshape(cout, k, k, cin).permute(0, 3, 1, 2)
    return F.conv2d(x, w, b, stride=stride, padding=k // 2)


@pytest.mark.parametrize('down', [False, True])
def test_enc_cell_folding(down):
    cell = EncCellSpec('c', 8, 16 if down else 8, down)
    sd = {}
    _enc_cell(sd, _Rng(1), cell)
    w = FO.fold_enc_cell(sd, cell)
    x = torch.randn(2, 8, 8, 8)
    st = 2 if down else 1
    a = F.silu(x * w['pro_scale'].view(1, -1, 1, 1) + w['pro_shift'].view(1, -1, 1, 1))
    t1 = _conv_from_layout(a, w['w1'], w['b1'], cell.cout, 8, 3, st)
    t2 = _conv_from_layout(F.silu(t1), w['w2'], w['b2'], cell.cout, cell.cout, 3)
    gate = torch.sigmoid(F.linear(F.relu(F.linear(t2.mean(dim=[2, 3]), w['se_w1'], w['se_b1'])), w['se_w2'], w['se_b2']))
    skip = _conv_from_layout(F.silu(x), w['ws'], w['bs'], cell.cout, 8, 1, 2) if down else x
    out = skip + 0.1 * gate.view(2, -1, 1, 1) * t2
    np.testing.assert_allclose(out.numpy(), O.enc_cell(sd, cell, x).numpy(), atol=2e-6)


@pytest.mark.parametrize('up', [False, True])
def test_dec_cell_folding_and_commuted_upsampling(up):
    cell = DecCellSpec('c', 8, 4 if up else 8, up, 6)
    sd = {}
    _dec_cell(sd, _Rng(2), cell)
    w = FO.fold_dec_cell(sd, cell)
    x = torch.randn(2, 8, 4, 4)
    t1 = F.conv2d(x, w['w1'].view(cell.hidden, 8, 1, 1), w['b1'])               # low resolution: 1x1 before nearest-up
    a = F.silu(t1)
    if up:
        a = F.interpolate(a, scale_factor=2, mode='nearest')
    wd = w['wd'].t().reshape(cell.hidden, 1, 5, 5)
    t2 = F.conv2d(a, wd, w['bd'], padding=2, groups=cell.hidden)
    t3 = F.conv2d(F.silu(t2), w['w2'].view(cell.cout, cell.hidden, 1, 1), w['b2'])
    gate = torch.sigmoid(F.linear(F.relu(F.linear(t3.mean(dim=[2, 3]), w['se_w1'], w['se_b1'])), w['se_w2'], w['se_b2']))
    if up:
        low = F.conv2d(x, w['ws'].view(cell.cout, 8, 1, 1), w['bs'])            # 1x1 before the bilinear interpolation
        skip = F.interpolate(low, scale_factor=2, mode='bilinear', align_corners=True)
    else:
        skip = x
    out = skip + 0.1 * gate.view(2, -1, 1, 1) * t3
    np.testing.assert_allclose(out.numpy(), O.dec_cell(sd, cell, x).numpy(), atol=3e-6)
    # backward layouts are the transposes / flips of the forward ones
    assert torch.equal(w['w1_bwd'], w['w1'].t()) and torch.equal(w['w2_bwd'], w['w2'].t())
    assert torch.equal(w['wd_bwd'].t().reshape(-1, 5, 5), wd[:, 0].flip(1, 2))


def test_vgg_head_folding_is_exact_pooling():
    spec = build_vgg_spec(10, 16)
    sd = init_vgg_state_dict(10, 16, 3)
    for f in (1, 2, 3):
        feat = torch.randn(3, spec.feat_channels, f, f)
        x = F.adaptive_avg_pool2d(F.relu(feat), (7, 7)).flatten(1)
        x = F.linear(x, sd['model.classifier.0.weight'])
        c = 'model.classifier.1'
        ref = F.batch_norm(x, sd[f'{c}.running_mean'], sd[f'{c}.running_var'], sd[f'{c}.weight'], sd[f'{c}.bias'], False, 0.0, 1e-5)
        h = FO.fold_vgg_head(sd, spec.feat_channels, f)
        nhwc = F.relu(feat).permute(0, 2, 3, 1).reshape(3, -1)
        np.testing.assert_allclose((nhwc @ h['w_head'].t() + h['b_head']).numpy(), ref.numpy(), atol=1e-5)
    m = adaptive_avgpool_matrix(2, 7)
    assert m.shape == (7, 2) and torch.allclose(m.sum(dim=1), torch.ones(7, dtype=torch.float64))
    assert m[3].tolist() == [0.5, 0.5] and m[0].tolist() == [1.0, 0.0] and m[6].tolist() == [0.0, 1.0]


def _flops(plan):
    tot = 0
    for d in plan.descs:
        if isinstance(d, L.ConvDesc):
            pix = d.N * d.Ho * d.Wo if d.sd == 1 else d.N * d.Hi * d.Wi
            tot += 2 * pix * d.KH * d.KW * (d.C1 + d.C2) * d.Cout
    return tot


def test_assumed_config_plan_matches_survey_flop_count(monkeypatch):
    """15.15 GFLOP/row forward for the assumed NVAE config (we skip the unused log-sigma half of the
    encoder samplers, apply the up-cell 1x1 convs before upsampling and fold the prior half of combiner_0:0,
    so the plan is slightly BELOW the reference's count)."""
    vspec = build_vgg_spec(100, 8)
    vsd = init_vgg_state_dict(100, 8, 0)
    sd = init_nvae_state_dict(ASSUMED_NVAE_CONFIG, ASSUMED_NVAE_RESOLUTION, 0)
    n = len(build_spec(ASSUMED_NVAE_CONFIG, ASSUMED_NVAE_RESOLUTION).groups)
    assert n == 24
    eng = Engine(sd, ASSUMED_NVAE_CONFIG, ASSUMED_NVAE_RESOLUTION, vsd, vspec, rows=1, rep=1, alphas=[0.5] * n,
                 device='cpu', dry_run=True)
    assert not any(isinstance(d, L.DecCellDesc) for d in eng.fwd.descs)   # one row: the fused cell would leave 255 CUs idle
    monkeypatch.setattr(Engine, 'fuse_min_workgroups', 0)
    eng = Engine(sd, ASSUMED_NVAE_CONFIG, ASSUMED_NVAE_RESOLUTION, vsd, vspec, rows=1, rep=1, alphas=[0.5] * n,
                 device='cpu', dry_run=True)
    from bench import dec_cell_algorithmic_flops
    nvae_fwd = sum(2 * (d.N * d.Ho * d.Wo) * d.KH * d.KW * (d.C1 + d.C2) * d.Cout
                   for d, nm in zip(eng.fwd.descs, eng.fwd.names) if isinstance(d, L.ConvDesc) and not nm.startswith('vgg'))
    assert any(isinstance(d, L.DecCellDesc) for d in eng.fwd.descs)       # the 16 x 16 decoder cells run as fused launches
    nvae_fwd += dec_cell_algorithmic_flops(eng.fwd)
    assert 13.5e9 < nvae_fwd < 15.2e9, nvae_fwd
    nvae_bwd = sum((2 * (d.N * d.Ho * d.Wo if d.sd == 1 else d.N * d.Hi * d.Wi) * d.KH * d.KW * (d.C1 + d.C2) * d.Cout)
                   for d, nm in zip(eng.bwd.descs, eng.bwd.names) if isinstance(d, L.ConvDesc) and not nm.startswith('vgg'))
    nvae_bwd += dec_cell_algorithmic_flops(eng.bwd)
    assert 0.9 * nvae_fwd < nvae_bwd < 1.1 * nvae_fwd
    with pytest.raises(RuntimeError):
        eng.forward()


def test_plans_build_for_every_golden_config_and_share_weights(golden_cases):
    for name, g in golden_cases.items():
        cfg, res = golden_cfg(g)
        sd = init_nvae_state_dict(cfg, res, 1)
        vspec = build_vgg_spec(10, 16)
        vsd = init_vgg_state_dict(10, 16, 2)
        n = len(build_spec(cfg, res).groups)
        store = WeightStore('cpu')
        e1 = Engine(sd, cfg, res, vsd, vspec, rows=2, rep=1, alphas=[0.3] * n, device='cpu', dry_run=True, store=store)
        b = store.bytes
        e2 = Engine(sd, cfg, res, vsd, vspec, rows=4, rep=2, alphas=[0.3] * n, device='cpu', dry_run=True, store=store)
        assert store.bytes == b and len(e1.fwd) == len(e2.fwd) and 0 < e1.bwd_split < len(e1.bwd)
        e2.set_alphas([0.1] * n)
        assert all(abs(d.alpha - 0.1) < 1e-7 for d, _ in e2._sampler_descs)
        with pytest.raises(ValueError):
            e2.set_alphas([0.1])


def test_no_cpu_fallback():
    cfg = dict(ASSUMED_NVAE_CONFIG)
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        Engine({}, cfg, ASSUMED_NVAE_RESOLUTION, {}, build_vgg_spec(100, 8), rows=1, rep=1, alphas=[0.0] * 24, device='cpu')
    from gen_adversarial_amd.defenses.ours.models import CelebaIdentityClassifier, TransStyleGanDefenseModel
    with pytest.raises(RuntimeError, match='GPU only'):
        CelebaIdentityClassifier('/nonexistent', 'cpu')
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        Engine.bare(2, device='cpu')
    with pytest.raises(RuntimeError, match='GPU only'):          # the Style-Transformer defender is a GPU path like the others
        TransStyleGanDefenseModel(None, '/nonexistent', [0.0] * 16, device='cpu')


def test_pgd_step_and_protocol_on_a_toy_net():
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Flatten(), torch.nn.Linear(3 * 8 * 8, 5))
    x = torch.rand(4, 3, 8, 8)
    y = net(x).argmax(dim=1)
    atk = PGDLinf(eps=8 / 255, step_size=2 / 255, steps=20)
    success, bound, adv = atk(x, y, net)
    assert adv.shape == x.shape and float(bound.max()) <= 8 / 255 + 1e-6
    assert float(adv.min()) >= 0 and float(adv.max()) <= 1
    assert torch.equal(net(adv).argmax(dim=1) != y, success)
    s1, b1, a1 = atk(x[:1], y[:1], net)
    assert isinstance(s1, bool) and isinstance(b1, float)
    g = torch.ones_like(x)
    nxt = PGDLinf.step(x, x, g, 0.01, 0.05)
    assert float((nxt - x).abs().max()) <= 0.01 + 1e-7


def test_oracle_noise_and_blur_helpers():
    x = torch.rand(2, 3, 16, 16)
    n = torch.randn(2, 3, 16, 16)
    y = D.add_gaussian_noise(x, n, 2.0)
    assert float(y.min()) >= 0 and float(y.max()) <= 1
    assert D.blur_kernel_size(64) == 15 and D.blur_kernel_size(256) == 255   # abstract_models.py:153-156
    b = D.apply_gaussian_blur(torch.rand(1, 3, 64, 64))
    assert b.shape == (1, 3, 64, 64)


def test_nf_cells_reduce_to_a_constant_shift():
    """The fold the engine uses for normalizing-flow cells (folding.nf_constant_shift) equals the oracle's masked-conv
    evaluation of architecture.py:221-253 on random latents."""
    import torch
    from gen_adversarial_amd import folding
    from gen_adversarial_amd.nvae_spec import build_spec, init_nvae_state_dict
    from oracle import nvae_oracle
    cfg = {'initial_channels': 8, 'num_pre-post_process_blocks': 1, 'num_pre-post_process_cells': 1, 'num_scales': 2,
           'num_groups_per_scale': 2, 'is_adaptive': False, 'min_groups_per_scale': 1, 'num_cells_per_group': 1,
           'num_latent_per_group': 4, 'num_logistic_mixtures': 3, 'num_nf_cells': 3}
    spec = build_spec(cfg, (3, 16, 16))
    sd = init_nvae_state_dict(cfg, (3, 16, 16), 5)
    for gs in spec.groups:
        key = f'{gs.s}:{gs.g}'
        z = torch.randn(2, 4, 4, 4)
        c = folding.nf_constant_shift(sd, key, 3, 4)
        torch.testing.assert_close(nvae_oracle.nf_blocks(sd, spec, key, z), z - c.float().view(1, -1, 1, 1),
                                   rtol=0, atol=1e-6)
    sd['nf_cells.nf_0:0.0.cell1.layers.4.mask'].fill_(1.0)
    with pytest.raises(NotImplementedError):
        folding.nf_constant_shift(sd, '0:0', 3, 4)


def test_resnet_plans_build_and_checkpoint_layout(tmp_path):
    """ResNet-50 row (a13): the loader reads width/depth off a checkpoint in the reference layout
    (loading_utils.py:10-16), the plans build (dry run) for every block kind, and the BN fold is exact."""
    import torch
    import torch.nn.functional as TF
    from gen_adversarial_amd import folding
    from gen_adversarial_amd.defenses.loading_utils import load_ResNet50
    from gen_adversarial_amd.resnet_spec import build_resnet_spec, init_resnet_state_dict
    blocks, wd = (2, 1, 2, 1), 8
    sd = init_resnet_state_dict(2, wd, 3, blocks)
    torch.save({'state_dict': sd}, tmp_path / 'r.pt')
    w = load_ResNet50(str(tmp_path / 'r.pt'), 'cpu')
    assert (w.width_div, w.blocks, w.n_classes) == (wd, blocks, 2)
    spec = w.spec
    assert [b.stride for b in spec.blocks] == [1, 1, 2, 2, 1, 2] and [b.downsample for b in spec.blocks] == [True, False, True, True, False, True]
    full = build_resnet_spec(2)
    assert len(full.blocks) == 16 and full.feat_channels == 2048 and full.blocks[3].cin == 256 and full.blocks[3].stride == 2
    eng = Engine(None, None, (3, 64, 64), sd, spec, rows=2, rep=1, alphas=[], device='cpu', dry_run=True)
    kinds = {type(d).__name__ for d in eng.fwd.descs} | {type(d).__name__ for d in eng.bwd.descs}
    assert {'Maxpool3s2Desc', 'AvgpoolActDesc', 'Interleave2Desc', 'ConvDesc'} <= kinds
    # conv + eval BN == folded conv + bias
    blk = spec.blocks[2]
    f = folding.fold_resnet_block(sd, blk)
    x = torch.randn(1, blk.width, 6, 6)
    p = blk.prefix
    ref = TF.batch_norm(TF.conv2d(x, sd[f'{p}.conv2.weight'], stride=2, padding=1), sd[f'{p}.bn2.running_mean'],
                        sd[f'{p}.bn2.running_var'], sd[f'{p}.bn2.weight'], sd[f'{p}.bn2.bias'], False, 0.0, 1e-5)
    wf = f['w2'].view(blk.width, 3, 3, blk.width).permute(0, 3, 1, 2)
    torch.testing.assert_close(TF.conv2d(x, wf, f['b2'], stride=2, padding=1), ref, rtol=1e-5, atol=1e-5)


def test_batched_evaluation_equals_the_one_image_protocol():
    """experiments/test_defense.evaluate_shard with batch_images > 1 (batched PGD) returns the per-image table of the
    reference's one-image-at-a-time loop (deterministic toy classifier)."""
    import torch
    from gen_adversarial_amd.attacks.pgd import PGDLinf
    from gen_adversarial_amd.experiments.test_defense import evaluate_shard
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 3, padding=1), torch.nn.ReLU(), torch.nn.AdaptiveAvgPool2d(1),
                              torch.nn.Flatten(), torch.nn.Linear(4, 3)).eval()
    for p in net.parameters():
        p.requires_grad_(False)
    x = torch.rand(7, 3, 8, 8)
    y = net(x).argmax(dim=1)
    y[2] = (y[2] + 1) % 3                      # one clean-misclassified image
    attacks = {'pgd': PGDLinf(eps=0.3, step_size=0.05, steps=12)}
    one = evaluate_shard(net, attacks, x, y, batch_images=1)
    many = evaluate_shard(net, attacks, x, y, batch_images=3)
    torch.testing.assert_close(many, one, rtol=0, atol=1e-6)
    assert one[2, 0] == 0 and one[:, 0].sum() == 6


def test_styled_conv_plans_build_without_a_gpu():
    """the StyleGAN2 modulated-conv builder (engine_stylegan.py) emits well-formed forward / backward plans"""
    from gen_adversarial_amd import _lib as L
    from gen_adversarial_amd.engine_core import Act
    from gen_adversarial_amd.stylegan_spec import StyledConvSpec, init_styled_conv_state_dict
    sp = StyledConvSpec('conv1', 32, 64, 3, 64, 8, True, True)
    rgb = StyledConvSpec('to_rgb1', 64, 3, 1, 64, 8, False, False)
    eng = Engine.bare(2, device='cpu', dry_run=True)
    x, w = Act(eng, 2, 8, 8, 32, 'x'), Act(eng, 2, 1, 1, 64, 'w')
    h = eng.styled_conv(init_styled_conv_state_dict(sp, 1), sp, x, w, noise=torch.randn(8, 8))
    img = eng.styled_conv(init_styled_conv_state_dict(rgb, 2), rgb, h, w)
    eng.finish()
    assert (img.c, h.c) == (4, 64)
    kinds = [type(d).__name__ for d in eng.fwd.descs]
    assert kinds.count('ConvDesc') == 5 and kinds.count('ModoutDesc') == 1 and kinds.count('UnaryDesc') == 2      # ToRGB's tail is its conv's bias
    # backward: the latent gradient is written by the last layer's modulation^T and accumulated by the first one's
    mods = [d for d, n in zip(eng.bwd.descs, eng.bwd.names) if n.endswith('modulation^T')]
    assert len(mods) == 2 and not mods[0].addend and mods[1].addend


def test_stylegan_generator_plans_build_without_a_gpu():
    """Generator wiring (engine_stylegan.build_stylegan): layer count, latent slices shared between ToRGB and the next
    up-sampling conv accumulate, the constant input gets no gradient op"""
    from gen_adversarial_amd.engine_core import Act
    from gen_adversarial_amd.stylegan_spec import build_stylegan_spec, init_stylegan_state_dict
    spec = build_stylegan_spec(32, width_div=16, style_dim=64)
    assert spec.n_latent == 8 and len(spec.convs) == 6 and len(spec.to_rgbs) == 3
    sd = init_stylegan_state_dict(spec, 3)
    eng = Engine.bare(2, device='cpu', dry_run=True)
    lat = Act(eng, 2, 1, 1, spec.n_latent * spec.style_dim, 'latent')
    img = eng.build_stylegan(sd, spec, lat)
    eng.finish()
    assert (img.h, img.w, img.c) == (32, 32, 4)
    names = eng.bwd.names
    # d x rides on the style-gradient reduction of its layer (one read of the conv^T output); the constant input gets none
    assert 'conv1.dstyle_conv' in names and 'convs.0.dstyle_conv+dx' in names and not any(n.endswith('.dx') for n in names)
    assert not any(n.endswith('interleave') for n in eng.fwd.names)      # t stays in the parity conv's depth-to-space form
    mods = {n: d for d, n in zip(eng.bwd.descs, names) if n.endswith('modulation^T')}
    assert len(mods) == 11
    assert not mods['to_rgbs.2.modulation^T'].addend            # latent[:, 7]: one reader
    # latent[:, 5] has two readers: the later layer's backward runs first and writes, the earlier one accumulates
    assert not mods['convs.4.modulation^T'].addend and mods['to_rgbs.1.modulation^T'].addend
    assert sum(n.endswith('skip_upsample') for n in eng.fwd.names) == 3


def _small_e4e_defense(rows=2, rep=1, device='cpu', dry_run=True, precision='bf16x3', share_encoder=False):
    from gen_adversarial_amd.e4e_spec import build_e4e_spec, init_e4e_state_dict
    from gen_adversarial_amd.resnet_spec import build_resnet_spec, init_resnet_state_dict
    from gen_adversarial_amd.stylegan_spec import build_stylegan_spec, init_stylegan_state_dict
    size, res = 64, 64
    espec = build_e4e_spec(size, 4, (1, 1, 1, 1))
    esd = init_e4e_state_dict(size, 4, 3, (1, 1, 1, 1))
    gspec = build_stylegan_spec(size, width_div=8, style_dim=espec.style_dim)
    gsd = init_stylegan_state_dict(gspec, 4)
    cspec = build_resnet_spec(2, 8, (1, 1, 1, 1))
    csd = init_resnet_state_dict(2, 8, 5, (1, 1, 1, 1))
    g = torch.Generator().manual_seed(6)
    avg = 0.5 * torch.randn(gspec.n_latent, gspec.style_dim, generator=g)
    alphas = [0.1 * (j % 4) for j in range(gspec.n_latent)]
    eng = Engine.bare(rows, device=device, dry_run=dry_run, precision=precision, rep=rep, resolution=(3, res, res), alphas=alphas,
                      share_encoder=share_encoder)
    eng.build_e4e_defense(esd, espec, gsd, gspec, avg, csd, cspec, pool_to=32)
    return eng, (esd, espec, gsd, gspec, avg, csd, cspec, alphas)


def test_e4e_defense_plans_build_without_a_gpu():
    """encoder -> latent mixing -> synthesis -> face_pool -> classifier as one plan pair (engine_stylegan.build_e4e_defense)"""
    eng, _ = _small_e4e_defense()
    f, b = eng.fwd.names, eng.bwd.names
    assert f[0] == 'image_in' and b[-1] == 'image_in^T'
    order = [f.index(n) for n in ('e4e.input.conv', 'sg.mapping.pixelnorm', 'latent_mix', 'conv1.modulation', 'face_pool_denorm', 'resnet.conv1')]
    assert order == sorted(order)
    rorder = [b.index(n) for n in ('resnet.conv1^T', 'face_pool_denorm^T', 'conv1.modulation^T', 'latent_mix^T', 'e4e.w0.grad', 'e4e.input.conv^T')]
    assert rorder == sorted(rorder)
    assert not any(n.startswith('sg.mapping') for n in b)          # the mapping network sees noise only: no backward
    assert eng.logits.shape[0] == 2 and eng.eps[0].shape == (2, 10, 128)
    # EoT replicas share the encoder pass: its ops see one row per image, everything after the latent mixing sees all rows
    sh, _ = _small_e4e_defense(rows=6, rep=3, share_encoder=True)
    n_of = {n: d.N for d, n in zip(sh.fwd.descs, sh.fwd.names) if hasattr(d, 'N')}
    assert n_of['e4e.input.conv'] == 2 and n_of['conv1.conv'] == 6 and n_of['resnet.conv1'] == 6
    assert sh.x_in.shape[0] == 2 and sh.dx.shape[0] == 2 and sh.dlogits.shape[0] == 6


def test_every_reference_config_is_present_and_parses():
    """configs/*.yaml are the reference's files, copied as data: 45 files; each `ours_*` config has one alpha
    per latent index of its experiment (24 NVAE groups / 18 e4e styles / 16 Style-Transformer styles)"""
    import glob
    import os
    import yaml
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'configs')
    files = sorted(glob.glob(os.path.join(root, '*.yaml')))
    assert len(files) == 45
    n_alpha = {'ids': 24, 'gender': 18, 'cars': 16}
    for f in files:
        y = yaml.safe_load(open(f))
        assert 'classifier_path' in y
        name = os.path.basename(f)
        if name.startswith('ours_'):
            exp = name[:-5].split('_')[-1]
            assert len(y['interpolation_alphas']) == n_alpha[exp], name
            assert {'autoencoder_path', 'alpha_attenuation', 'initial_noise_eps', 'gaussian_blur_input'} <= set(y)


def test_trans_defense_plans_build_without_a_gpu():
    """resize + crop -> IR-SE trunk -> 3 transformer decoder layers -> latent mixing -> synthesis -> pool / band / resize ->
    ResNeXt as one plan pair (engine_trans.build_trans_defense), and the order of the decoder layer's backward ops"""
    from gen_adversarial_amd.trans_spec import build_trans_spec, init_trans_state_dict
    from gen_adversarial_amd.stylegan_spec import build_stylegan_spec, init_stylegan_state_dict
    from gen_adversarial_amd.resnet_spec import build_resnet_spec, init_resnet_state_dict
    tspec, tsd = build_trans_spec(4, (1, 1, 1, 1)), init_trans_state_dict(4, 1, (1, 1, 1, 1))
    gspec = build_stylegan_spec(32, width_div=8, style_dim=tspec.d_model)
    gsd = init_stylegan_state_dict(gspec, 2)
    cspec, csd = build_resnet_spec(4, 2, (1, 1, 1, 1), 4, 8), init_resnet_state_dict(4, 2, 3, (1, 1, 1, 1), 4, 8)
    eng = Engine.bare(4, device='cpu', dry_run=True, rep=2, resolution=(3, 64, 64), alphas=[0.1] * 16, share_encoder=True)
    eng.build_trans_defense(tsd, tspec, gsd, gspec, torch.zeros(16, tspec.d_model), csd, cspec, pool_to=32, mid=128, crop=16)
    f, b = eng.fwd.names, eng.bwd.names
    order = [f.index(n) for n in ('image_in', 'trans.resize_crop', 'e4e.input.conv', 'trans.transformerlayer_coarse.sa.in_proj',
                                  'trans.transformerlayer_fine.norm3', 'latent_mix', 'conv1.modulation', 'face_pool_band_resize_denorm', 'resnet.conv1')]
    assert order == sorted(order)
    p = 'trans.transformerlayer_fine.'
    want = ['norm3^T', 'norm3.residual^T', 'ff.linear2^T', 'ff.linear1^T', 'norm2^T', 'norm2.residual^T', 'ca.out_proj^T', 'ca.attn^T',
            'ca.kv_proj^T', 'ca.q_proj^T', 'norm1^T', 'norm1.residual^T', 'sa.out_proj^T', 'sa.attn^T', 'sa.in_proj^T']
    got = [n[len(p):] for n in b if n.startswith(p)]
    assert got == want
    assert not any(n.startswith('trans.transformerlayer_coarse.sa.in_proj^T') for n in b)      # the queries are a constant of the checkpoint
    assert b.index('latent_mix^T') < b.index(p + 'norm3^T') < b.index('trans.resize_crop^T') < b.index('image_in^T')
    n_of = {n: d.N for d, n in zip(eng.fwd.descs, eng.fwd.names) if hasattr(d, 'N')}
    assert n_of['e4e.input.conv'] == 2 and n_of['conv1.conv'] == 4                             # shared encoder: one pass per image


def test_channel_padding_helpers_keep_the_convolution():
    """folding.pad_conv_out / pad_cols / pad_rows: the engine rounds the 20 latent channels and the 100 mixture logits up to a multiple
    of 8 (so that the convs around them run on the split-bf16 kernels); padded weights on zero-padded tensors give the same numbers"""
    from gen_adversarial_amd.folding import conv_bwd_layout, conv_fwd_layout, pad_cols, pad_conv_out, pad_rows
    g = torch.Generator().manual_seed(3)
    cin, cout, cp = 6, 5, 8
    w = torch.randn(cout, cin, 3, 3, generator=g)
    b = torch.randn(cout, generator=g)
    f = {'w': conv_fwd_layout(w), 'w_bwd': conv_bwd_layout(w), 'b': b}
    fp = pad_conv_out(f, cout, cp)
    assert fp['w'].shape == (cp, 9 * cin) and fp['w_bwd'].shape == (cin, 9 * cp) and fp['b'].shape == (cp,)
    x = torch.randn(2, cin, 5, 5, generator=g)
    ref = torch.nn.functional.conv2d(x, w, b, padding=1)
    wp = fp['w'].view(cp, 3, 3, cin).permute(0, 3, 1, 2)                     # back from [Cout][taps * Cin]
    got = torch.nn.functional.conv2d(x, wp, fp['b'], padding=1)
    assert torch.equal(got[:, :cout], ref) and bool((got[:, cout:] == 0).all())
    # transposed conv of a cotangent whose pad channels are zero: same input gradient
    cot = torch.randn_like(ref)
    cotp = torch.zeros(2, cp, 5, 5)
    cotp[:, :cout] = cot
    wb = fp['w_bwd'].view(cin, 3, 3, cp).permute(0, 3, 1, 2)
    wb0 = f['w_bwd'].view(cin, 3, 3, cout).permute(0, 3, 1, 2)
    assert torch.allclose(torch.nn.functional.conv2d(cotp, wb, padding=1), torch.nn.functional.conv2d(cot, wb0, padding=1), atol=1e-6)
    # a 1x1 weight whose last n inputs get a padded pitch, and its transpose
    m = torch.randn(4, 10 + 3, generator=g)
    mp = pad_cols(m, 10, 3, 8)
    assert mp.shape == (4, 18) and torch.equal(mp[:, :13], m) and bool((mp[:, 13:] == 0).all())
    r = pad_rows(m[:, 10:].t().contiguous(), 8)
    assert r.shape == (8, 4) and torch.equal(r[:3], m[:, 10:].t()) and bool((r[3:] == 0).all())
    assert pad_cols(m, 10, 3, 3) is m and pad_rows(r, 8) is r


def test_fused_decoder_cells_are_chosen_by_launch_size():
    """Engine plans (dry run): a fused decoder-cell launch walks the hidden width serially, so it pays only when its grid fills
    the chip — 16 cells at 16 x 16 x 128 (one workgroup per image) from 160 rows, 16 more at 8 x 8 x 256 (two images per workgroup)
    from 320 rows; the reference protocol of one image x EoT 32 keeps the three unfused launches."""
    vspec = build_vgg_spec(100, 8)
    vsd = init_vgg_state_dict(100, 8, 0)
    sd = init_nvae_state_dict(ASSUMED_NVAE_CONFIG, ASSUMED_NVAE_RESOLUTION, 0)
    n = len(build_spec(ASSUMED_NVAE_CONFIG, ASSUMED_NVAE_RESOLUTION).groups)
    store = WeightStore('cpu')
    for rows, want in ((32, 0), (128, 0), (160, 16), (256, 16), (320, 32)):
        eng = Engine(sd, ASSUMED_NVAE_CONFIG, ASSUMED_NVAE_RESOLUTION, vsd, vspec, rows=rows, rep=32, alphas=[0.5] * n,
                     device='meta', dry_run=True, store=store)
        got = sum(isinstance(d, L.DecCellDesc) for d in eng.fwd.descs)
        assert got == want and sum(isinstance(d, L.DecCellDesc) for d in eng.bwd.descs) == want, (rows, got)


def test_competitor_defender_plans_build_without_a_gpu():
    """SURVEY.md §8 row f4 (dry run): the ND-VAE and A-VAE competitor defenders as one plan pair each, with a VGG and with a ResNet
    behind them; the backward plan replays the classifier first, the purifier's boundary ops last; the specs follow the
    reference's constructors (channel bookkeeping of Defence_NVAE, block list of StyledGenerator)."""
    from gen_adversarial_amd.avae_spec import build_avae_spec, init_avae_state_dict
    from gen_adversarial_amd.ndvae_spec import build_ndvae_spec, init_ndvae_h, init_ndvae_state_dict
    from gen_adversarial_amd.resnet_spec import build_resnet_spec, init_resnet_state_dict
    vgg = (build_vgg_spec(10, 16), init_vgg_state_dict(10, 16, 3))
    res = (build_resnet_spec(4, 8, (1, 1, 1, 1)), init_resnet_state_dict(4, 8, 3, (1, 1, 1, 1)))
    # ---- ND-VAE: configs/competitor_ndvae_ids.yaml (scales 1, groups 8, cells 2) and _gender.yaml (scales 2, groups 4) shapes
    ids = build_ndvae_spec({'x_channels': 3, 'encoding_channels': 32, 'pre_proc_groups': 2, 'scales': 1, 'groups': 8, 'cells': 2, 'input_dim': 64})
    assert (ids.top_channels, ids.top_res, ids.h_shape, ids.latent_shapes) == (128, 16, (128, 16, 16), [(128, 16), (128, 16)])
    assert len(ids.enc_scales[0]) == 16 and len(ids.dec_scales[0].groups) == 8 and [c.hidden for c in ids.post_cells] == [256, 4096, 128, 1024]
    gen = build_ndvae_spec({'x_channels': 3, 'encoding_channels': 16, 'pre_proc_groups': 2, 'scales': 2, 'groups': 4, 'cells': 2, 'input_dim': 256})
    assert gen.latent_shapes == [(128, 32), (128, 32), (64, 64)] and gen.dec_scales[1].up.cout == 64
    with pytest.raises(ValueError):          # Decoder_tower.h would not meet the encoder's top feature map (torch.cat fails in the reference)
        build_ndvae_spec({'x_channels': 3, 'encoding_channels': 4, 'pre_proc_groups': 1, 'scales': 2, 'groups': 1, 'cells': 1, 'input_dim': 64})
    cfg = {'x_channels': 3, 'encoding_channels': 4, 'pre_proc_groups': 2, 'scales': 2, 'groups': 2, 'cells': 2, 'input_dim': 32}
    spec, sd, h = build_ndvae_spec(cfg), init_ndvae_state_dict(cfg, 1), init_ndvae_h(cfg, 2)
    for cspec, csd in (vgg, res):
        eng = Engine.bare(4, device='cpu', dry_run=True, rep=2, resolution=(3, 32, 32), alphas=[], noise_eps=0.1)
        eng.build_ndvae_defense(sd, spec, h, csd, cspec)
        f, b = eng.fwd.names, eng.bwd.names
        assert f[0] == 'image_in' and f.index('nd.stem') < f.index('decoder.samplers.0.sample') < f.index('nd.dml_mean') < len(f) - 1
        assert b[-1] == 'image_in^T' and b.index('nd.dml_mean^T') >= eng.bwd_split > 0
        assert sum(isinstance(d, L.SamplerDesc) and d.mode == 1 for d in eng.fwd.descs) == 3
        assert [tuple(e.shape[1:]) for e in eng.eps] == [(c, r, r) for c, r in spec.latent_shapes] and eng.noise is not None
    # ---- A-VAE: the three output sizes of the reference (model.py:37-66), skip concatenated at 16 x 16
    assert [(b_.kind, b_.cin, b_.cout, b_.res, b_.skip) for b_ in build_avae_spec(64).blocks] == [
        ('initial', 512, 512, 4, False), ('up', 512, 512, 8, False), ('up', 512, 512, 16, False), ('fused', 768, 256, 32, True),
        ('fused', 256, 128, 64, False)]
    assert len(build_avae_spec(128).blocks) == 6 and len(build_avae_spec(256).blocks) == 7
    for size, k in ((64, 2), (128, 4)):
        aspec, asd = build_avae_spec(size, 8), init_avae_state_dict(size, 1, 8)
        for cspec, csd in (vgg, res):
            eng = Engine.bare(4, device='cpu', dry_run=True, rep=2, resolution=(3, size, size), alphas=[])
            eng.build_avae_defense(asd, aspec, k, csd, cspec)
            f, b = eng.fwd.names, eng.bwd.names
            assert f[:2] == ['image_in', 'avae.avgpool'] and b[-2:] == ['avae.avgpool^T', 'image_in^T']
            assert sum(isinstance(d, L.AvaeDesc) and d.mode == L.GA_AVAE_ADAIN for d in eng.fwd.descs) == 2 * len(aspec.blocks)
            assert b.index('avae.to_rgb^T') >= eng.bwd_split > 0 and b.index('avae.sample^T') > b.index('generator.progression.0.adain1^T')
            assert len(eng.eps) == 1 + len(aspec.blocks)


@pytest.mark.parametrize('cin,cout', [(32, 32), (32, 104), (64, 64), (64, 8)])
def test_thin_kernel_weight_fragments_follow_the_header(cin, cout):
    """include/ga_ops.h, ga_conv_desc.w_frag, tile 11: bf16 [ceil(Cout/32)][9 taps][C1/16 k steps][hi | lo][64 lanes][8] with element e
    of lane l = W[32 t + (l & 31)][tap * C1 + 16 kstep + 8 (l >> 5) + e], rows >= Cout zero, hi + lo the bf16 split of W"""
    w = torch.randn(cout, 9 * cin, generator=torch.Generator().manual_seed(cin + cout))
    store = WeightStore('cpu')
    f = store.frag_thin(w)
    nt, ks = (cout + 31) // 32, cin // 16
    assert f.dtype == torch.bfloat16 and tuple(f.shape) == (nt, 9, ks, 2, 2, 32, 8)        # [.., hi | lo, lane >> 5, lane & 31, e]
    hi, lo = store.split(w)
    gen = torch.Generator().manual_seed(1)
    for _ in range(200):
        t, tap, k, part = (int(torch.randint(n, (1,), generator=gen)) for n in (nt, 9, ks, 2))
        lane, e = int(torch.randint(64, (1,), generator=gen)), int(torch.randint(8, (1,), generator=gen))
        row, col = 32 * t + (lane & 31), tap * cin + 16 * k + 8 * (lane >> 5) + e
        want = (hi, lo)[part][row, col] if row < cout else torch.zeros((), dtype=torch.bfloat16)
        assert f[t, tap, k, part, lane >> 5, lane & 31, e] == want, (t, tap, k, part, lane, e)
    assert torch.equal(hi.float() + lo.float(), (w.to(torch.bfloat16).float() + (w - w.to(torch.bfloat16).float()).to(torch.bfloat16).float()))


def test_torgb_layers_have_no_tail_pass_and_no_backward_conv():
    """round 4 (engine_stylegan.py): ToRGB's tail is its conv's bias, its backward conv W^T dt is formed inside the style-gradient /
    dx reduction (ga_rowchan_reduce a_src / a_w); the up-sampling layers keep t in depth-to-space form (t_planes, no interleave)"""
    from gen_adversarial_amd.engine_core import Act
    from gen_adversarial_amd.stylegan_spec import build_stylegan_spec, init_stylegan_state_dict
    spec = build_stylegan_spec(32, width_div=16, style_dim=64)
    eng = Engine.bare(2, device='cpu', dry_run=True)
    lat = Act(eng, 2, 1, 1, spec.n_latent * spec.style_dim, 'latent')
    eng.build_stylegan(init_stylegan_state_dict(spec, 3), spec, lat)
    eng.finish()
    fwd, bwd = dict(zip(eng.fwd.names, eng.fwd.descs)), dict(zip(eng.bwd.names, eng.bwd.descs))
    assert not any(n.startswith('to_rgb') and n.endswith('.tail') for n in fwd) and 'convs.0.tail' in fwd
    assert not any(n.startswith('to_rgb') and (n.endswith('.tail^T') or n.endswith('.conv^T')) for n in bwd)
    r = bwd['to_rgbs.0.conv^T+dstyle_conv+dx']
    assert isinstance(r, L.ReduceDesc) and not r.a and r.a_src and r.a_w and r.scaled and r.gate
    up = fwd['convs.0.tail']                                    # convs.0 up-samples: its tail reads the parity conv's planes
    assert not up.t and all(up.t_planes[i] for i in range(4)) and up.ld_planes == 4 * up.C
    upb = bwd['convs.0.tail^T']
    assert not upb.dt and all(upb.dt_planes[i] for i in range(4))
    plain = fwd['convs.1.tail']                                 # convs.1 does not: interleaved t, interleaved dt
    assert plain.t and not plain.t_planes[0] and bwd['convs.1.tail^T'].dt


def test_se_squeeze_of_few_large_rows_is_reduced_over_the_chip():
    """Engine._se_wide_rows: the IR-SE50 rows of the e4e / Style-Transformer defenders (32 - 64 rows of >= 1 MB) go through
    ga_rowchan_reduce; the NVAE cells (hundreds of small rows, or 32 rows of <= 0.5 MB) keep the fused one-workgroup-per-row squeeze"""
    assert Engine._se_wide_rows(32, 128 * 128, 64) and Engine._se_wide_rows(64, 32 * 32, 256)
    assert not Engine._se_wide_rows(1024, 128 * 128, 64)        # many rows: the fused form fills the chip
    assert not Engine._se_wide_rows(32, 64 * 64, 32) and not Engine._se_wide_rows(32, 16 * 16, 128)      # NVAE at 32 rows
