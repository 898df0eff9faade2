"""
Load-time weight preparation (host side, runs once per model): folds weight-norm and eval-mode batch-norm into
the tensors the HIP kernels consume, and lays them out as include/ga_ops.h documents.

Reference semantics being folded:
  * weight_norm(Conv2d) — torch parametrization, w = g * v / ||v|| per output channel (keys
    `parametrizations.weight.original0/1`; NVAE/modules/architecture.py:75,89,122,125,193,213; NVAE/model.py:106,186,
    211,228,312)
  * SyncBatchNorm(eps=1e-5) in .eval() (loading_utils.py:65) == y = x*s + t with s = gamma/sqrt(var+eps),
    t = beta - mean*s (architecture.py:120,123,165-173)
  * BatchNorm2d / BatchNorm1d of the VGG (src/classifier/model.py:37-45)
All folds are computed in float64 and rounded once to float32.
"""
from __future__ import annotations

from typing import Dict, Tuple

import torch

from .vgg_spec import adaptive_avgpool_matrix

SD = Dict[str, torch.Tensor]


def wn_weight64(sd: SD, prefix: str) -> torch.Tensor:
    g = sd[f'{prefix}.parametrizations.weight.original0'].double()
    v = sd[f'{prefix}.parametrizations.weight.original1'].double()
    norm = v.flatten(1).norm(dim=1).view(-1, 1, 1, 1)
    return v * (g / norm)


def bn_affine64(sd: SD, prefix: str, eps: float = 1e-5) -> Tuple[torch.Tensor, torch.Tensor]:
    s = sd[f'{prefix}.weight'].double() / torch.sqrt(sd[f'{prefix}.running_var'].double() + eps)
    t = sd[f'{prefix}.bias'].double() - sd[f'{prefix}.running_mean'].double() * s
    return s, t


def conv_fwd_layout(w: torch.Tensor) -> torch.Tensor:
    """[Cout,Cin,KH,KW] -> [Cout][KH*KW*Cin] with k = (kh*KW+kw)*Cin + c."""
    co = w.shape[0]
    return w.permute(0, 2, 3, 1).reshape(co, -1).contiguous()


def conv_bwd_layout(w: torch.Tensor) -> torch.Tensor:
    """weights of the backward-to-input convolution: [Cin][KH*KW*Cout], spatially flipped."""
    ci = w.shape[1]
    return w.flip(2, 3).permute(1, 2, 3, 0).reshape(ci, -1).contiguous()


def dw_layout(w: torch.Tensor, flip: bool = False) -> torch.Tensor:
    """[C,1,5,5] -> [25][C]."""
    if flip:
        w = w.flip(2, 3)
    return w.reshape(w.shape[0], 25).t().contiguous()


def subpixel_weights(w: torch.Tensor) -> dict:
    """Backward-to-input of a stride-2 convolution (pad = (k-1)//2, k in {1, 3}) as four stride-1 convolutions, one per
    output parity (a, b) = (row, column parity of the input pixel): hi = 2*ho + kh - pad, so parity a only meets the
    taps kh with (a + pad - kh) even, at ho = i + (a + pad - kh)/2 for hi = 2i + a.  Each sub-kernel is returned in the
    forward layout of ga_conv2d — [Cin][th*KW_ab*Cout + tw*Cout + co], taps ordered by increasing offset, window
    anchored at (i, j), pad 0 — together with its (KH_ab, KW_ab); a parity that meets no tap is absent (its plane is
    zero).  w: [Cout, Cin, k, k] (the folded forward weights)."""
    co, ci, k, _ = w.shape
    pad = (k - 1) // 2

    def taps(a):      # [(offset, kh)] sorted by offset
        t = [((a + pad - kh) // 2, kh) for kh in range(k) if (a + pad - kh) % 2 == 0]
        t.sort()
        assert [o for o, _ in t] == list(range(len(t))), (k, a, t)          # offsets 0.. : a plain window with pad 0
        return [kh for _, kh in t]

    out = {}
    for a in (0, 1):
        for b in (0, 1):
            ta, tb = taps(a), taps(b)
            if not ta or not tb:
                continue
            sub = w[:, :, ta][:, :, :, tb]                                    # [co, ci, KHab, KWab]
            out[(a, b)] = (f32(sub.permute(1, 2, 3, 0).reshape(ci, -1)), len(ta), len(tb))
    return out


def f32(t: torch.Tensor) -> torch.Tensor:
    return t.to(torch.float32).contiguous()


def fold_enc_cell(sd: SD, cell) -> dict:
    p = cell.prefix
    s0, t0 = bn_affine64(sd, f'{p}.residual.0')
    w1 = wn_weight64(sd, f'{p}.residual.2')
    b1 = sd[f'{p}.residual.2.bias'].double()
    s1, t1 = bn_affine64(sd, f'{p}.residual.3')
    w1f = w1 * s1.view(-1, 1, 1, 1)
    b1f = b1 * s1 + t1
    w2 = wn_weight64(sd, f'{p}.residual.5')
    b2 = sd[f'{p}.residual.5.bias'].double()
    out = {'pro_scale': f32(s0), 'pro_shift': f32(t0),
           'w1': f32(conv_fwd_layout(w1f)), 'w1_bwd': f32(conv_bwd_layout(w1f)), 'b1': f32(b1f),
           'w2': f32(conv_fwd_layout(w2)), 'w2_bwd': f32(conv_bwd_layout(w2)), 'b2': f32(b2)}
    out.update(_se(sd, f'{p}.residual.6'))
    if cell.down:
        ws = wn_weight64(sd, f'{p}.skip_connection.conv')
        out['ws'] = f32(conv_fwd_layout(ws))
        out['ws_bwd'] = f32(conv_bwd_layout(ws))
        out['bs'] = f32(sd[f'{p}.skip_connection.conv.bias'].double())
        for key, wsrc in (('w1_sub', w1f), ('ws_sub', ws)):                  # sub-pixel kernels of the two stride-2 transposes
            for (a, b), (wm, kh, kw) in subpixel_weights(wsrc).items():
                out[f'{key}{a}{b}'] = wm
    return out


def _nd_se(sd: SD, prefix: str) -> dict:
    """SE_Block of the ND-VAE competitor (NVAE.py:57-69): nn.Sequential(Linear, ReLU, Linear, Sigmoid) under `.se`"""
    return {'se_w1': f32(sd[f'{prefix}.se.0.weight']), 'se_b1': f32(sd[f'{prefix}.se.0.bias']),
            'se_w2': f32(sd[f'{prefix}.se.2.weight']), 'se_b2': f32(sd[f'{prefix}.se.2.bias'])}


def fold_nd_res_cell(sd: SD, cell) -> dict:
    """Residual_Cell_NVAE (competitors/nd_vae/modules/models/NVAE.py:255-297) in the layout of fold_enc_cell: bn1 = the prologue
    affine of conv1, bn2 folded into conv1; FactorizedReduce (:117-135) = ONE 2x2 / stride-2 / pad-0 convolution whose four
    output-channel quarters each see one tap: conv_1 pixel (0,0), conv_2 (1,1), conv_3 (0,1), conv_4 (1,0) of every 2x2 block."""
    p = cell.prefix
    s0, t0 = bn_affine64(sd, f'{p}.bn1')
    w1 = sd[f'{p}.conv1.weight'].double()
    s1, t1 = bn_affine64(sd, f'{p}.bn2')
    w1f = w1 * s1.view(-1, 1, 1, 1)
    b1f = sd[f'{p}.conv1.bias'].double() * s1 + t1
    w2 = sd[f'{p}.conv2.weight'].double()
    out = {'pro_scale': f32(s0), 'pro_shift': f32(t0),
           'w1': f32(conv_fwd_layout(w1f)), 'w1_bwd': f32(conv_bwd_layout(w1f)), 'b1': f32(b1f),
           'w2': f32(conv_fwd_layout(w2)), 'w2_bwd': f32(conv_bwd_layout(w2)), 'b2': f32(sd[f'{p}.conv2.bias'].double())}
    out.update(_nd_se(sd, f'{p}.squeeze_excitation'))
    if cell.down:
        ws = torch.zeros(cell.cout, cell.cin, 2, 2, dtype=torch.float64)
        bs, o = [], 0
        for i, (kh, kw) in enumerate(((0, 0), (1, 1), (0, 1), (1, 0)), start=1):
            wi = sd[f'{p}.skip.conv_{i}.weight'].double()[:, :, 0, 0]
            ws[o:o + wi.shape[0], :, kh, kw] = wi
            bs.append(sd[f'{p}.skip.conv_{i}.bias'].double())
            o += wi.shape[0]
        assert o == cell.cout
        out['ws'] = f32(conv_fwd_layout(ws))
        out['bs'] = f32(torch.cat(bs))
        for key, wsrc in (('w1_sub', w1f), ('ws_sub', ws)):                  # sub-pixel kernels of the two stride-2 transposes
            for (a, b), (wm, kh, kw) in subpixel_weights(wsrc).items():
                out[f'{key}{a}{b}'] = wm
    return out


def fold_nd_gen_cell(sd: SD, cell) -> dict:
    """Generative_Cell_NVAE (NVAE.py:156-228) in the layout of fold_dec_cell plus the pointwise conv of its
    depthwise_separable_conv (`wp`): bn1 and bn_expanded1 folded into `expand`, the depthwise conv keeps its own bias,
    bn_expanded2 folded into the pointwise conv, bn2 into `expand2`; the up-sampling cell's skip is bilinear x2 then a 1x1 conv
    (the two commute: the engine convolves at low resolution)."""
    p = cell.prefix
    s0, t0 = bn_affine64(sd, f'{p}.bn1')
    w1 = sd[f'{p}.expand.weight'].double()[:, :, 0, 0]
    s1, t1 = bn_affine64(sd, f'{p}.bn_expanded1')
    w1f = s1.view(-1, 1) * w1 * s0.view(1, -1)
    b1f = s1 * (w1 @ t0 + sd[f'{p}.expand.bias'].double()) + t1
    wd = sd[f'{p}.dep_sep_conv.depthwise.weight'].double()
    wp = sd[f'{p}.dep_sep_conv.pointwise.weight'].double()[:, :, 0, 0]
    s2, t2 = bn_affine64(sd, f'{p}.bn_expanded2')
    wpf = s2.view(-1, 1) * wp
    bpf = s2 * sd[f'{p}.dep_sep_conv.pointwise.bias'].double() + t2
    w2 = sd[f'{p}.expand2.weight'].double()[:, :, 0, 0]
    s3, t3 = bn_affine64(sd, f'{p}.bn2')
    w2f = s3.view(-1, 1) * w2
    b2f = s3 * sd[f'{p}.expand2.bias'].double() + t3
    out = {'w1': f32(w1f), 'w1_bwd': f32(w1f.t()), 'b1': f32(b1f),
           'wd': f32(dw_layout(wd)), 'wd_bwd': f32(dw_layout(wd, flip=True)), 'bd': f32(sd[f'{p}.dep_sep_conv.depthwise.bias'].double()),
           'wp': f32(wpf), 'wp_bwd': f32(wpf.t()), 'bp': f32(bpf),
           'w2': f32(w2f), 'w2_bwd': f32(w2f.t()), 'b2': f32(b2f)}
    out.update(_nd_se(sd, f'{p}.squeeze_excitation'))
    if cell.up:
        ws = sd[f'{p}.skip.1.weight'].double()[:, :, 0, 0]
        out['ws'] = f32(ws)
        out['ws_bwd'] = f32(ws.t())
        out['bs'] = f32(sd[f'{p}.skip.1.bias'].double())
    return out


def _se(sd: SD, prefix: str) -> dict:
    return {'se_w1': f32(sd[f'{prefix}.linear_1.weight']), 'se_b1': f32(sd[f'{prefix}.linear_1.bias']),
            'se_w2': f32(sd[f'{prefix}.linear_2.weight']), 'se_b2': f32(sd[f'{prefix}.linear_2.bias'])}


def fold_dec_cell(sd: SD, cell) -> dict:
    p, o = cell.prefix, cell.ridx
    s0, t0 = bn_affine64(sd, f'{p}.residual.{o + 0}')
    w1 = sd[f'{p}.residual.{o + 1}.weight'].double()[:, :, 0, 0]             # [hid, cin]
    s1, t1 = bn_affine64(sd, f'{p}.residual.{o + 2}')
    w1f = s1.view(-1, 1) * w1 * s0.view(1, -1)
    b1f = s1 * (w1 @ t0) + t1
    wd = sd[f'{p}.residual.{o + 4}.weight'].double()                         # [hid,1,5,5]
    s2, t2 = bn_affine64(sd, f'{p}.residual.{o + 5}')
    wdf = wd * s2.view(-1, 1, 1, 1)
    w2 = sd[f'{p}.residual.{o + 7}.weight'].double()[:, :, 0, 0]             # [cout, hid]
    s3, t3 = bn_affine64(sd, f'{p}.residual.{o + 8}')
    w2f = s3.view(-1, 1) * w2
    out = {'w1': f32(w1f), 'w1_bwd': f32(w1f.t()), 'b1': f32(b1f),
           'wd': f32(dw_layout(wdf)), 'wd_bwd': f32(dw_layout(wdf, flip=True)), 'bd': f32(t2),
           'w2': f32(w2f), 'w2_bwd': f32(w2f.t()), 'b2': f32(t3)}
    out.update(_se(sd, f'{p}.residual.{o + 9}'))
    if cell.up:
        ws = wn_weight64(sd, f'{p}.skip_connection.conv')[:, :, 0, 0]
        out['ws'] = f32(ws)
        out['ws_bwd'] = f32(ws.t())
        out['bs'] = f32(sd[f'{p}.skip_connection.conv.bias'].double())
    return out


def fold_wn_conv(sd: SD, prefix: str, out_slice: slice = None, in_slice: slice = None) -> dict:
    w = wn_weight64(sd, prefix)
    b = sd[f'{prefix}.bias'].double()
    if out_slice is not None:
        w, b = w[out_slice], b[out_slice]
    if in_slice is not None:
        w = w[:, in_slice]
    return {'w': f32(conv_fwd_layout(w)), 'w_bwd': f32(conv_bwd_layout(w)), 'b': f32(b)}


def pad_conv_out(f: dict, cout: int, cout_pad: int) -> dict:
    """Zero-pads the OUTPUT-channel axis of a folded conv from `cout` to `cout_pad` (a multiple of 8): 'w' [cout][K] -> [cout_pad][K]
    (zero rows), 'b' zero-extended, 'w_bwd' [Cin][taps*cout] -> [Cin][taps*cout_pad] (zero columns).  The pad channels of the output
    are exact zeros and their cotangents meet zero weights; the point is the transposed conv, whose INPUT channel count must be a
    multiple of 8 to run on the split-bf16 kernels instead of the exact-fp32 one (the 100 mixture logits: 1.5 -> 0.4 ms)."""
    if cout_pad == cout:
        return f
    w, wb, b = f['w'], f['w_bwd'], f['b']
    wp = torch.zeros(cout_pad, w.shape[1], dtype=w.dtype)
    wp[:cout] = w
    bp = torch.zeros(cout_pad, dtype=b.dtype)
    bp[:cout] = b
    cin, taps = wb.shape[0], wb.shape[1] // cout
    wbp = torch.zeros(cin, taps, cout_pad, dtype=wb.dtype)
    wbp[:, :, :cout] = wb.view(cin, taps, cout)
    out = dict(f)
    out['w'], out['b'], out['w_bwd'] = wp.contiguous(), bp, wbp.reshape(cin, taps * cout_pad).contiguous()
    return out


def pad_cols(w: torch.Tensor, at: int, n: int, n_pad: int) -> torch.Tensor:
    """[R][... at, at+n ...] -> n_pad - n zero columns inserted after column at + n (1x1 weights whose last `n` inputs get a padded pitch)"""
    if n_pad == n:
        return w
    out = torch.zeros(w.shape[0], w.shape[1] + n_pad - n, dtype=w.dtype)
    out[:, :at + n] = w[:, :at + n]
    out[:, at + n_pad:] = w[:, at + n:]
    return out


def pad_rows(w: torch.Tensor, n_pad: int) -> torch.Tensor:
    if w.shape[0] == n_pad:
        return w
    out = torch.zeros(n_pad, w.shape[1], dtype=w.dtype)
    out[:w.shape[0]] = w
    return out


def pad_image_conv(f: dict, cin: int, ld: int) -> dict:
    """Zero-pads the input-channel axis of a folded image-consuming conv from `cin` to the NHWC image pitch `ld`:
    'w' [Cout][taps*cin] -> [Cout][taps*ld], 'w_bwd' [cin][taps*Cout] -> [ld][taps*Cout] (zero rows).  The pad channels of
    the image are zero and meet zero weights, so results are unchanged; the conv just sees a vectorisable tensor."""
    w, wb = f['w'], f['w_bwd']
    co, taps = w.shape[0], w.shape[1] // cin
    wp = torch.zeros(co, taps, ld, dtype=w.dtype)
    wp[:, :, :cin] = w.view(co, taps, cin)
    wbp = torch.zeros(ld, wb.shape[1], dtype=wb.dtype)
    wbp[:cin] = wb
    out = dict(f)
    out['w'], out['w_bwd'] = wp.reshape(co, taps * ld).contiguous(), wbp.contiguous()
    return out


# ---------------------------------------------------------------------------------------------------------------
# VGG
# ---------------------------------------------------------------------------------------------------------------

def fold_vgg_conv(sd: SD, i: int) -> dict:
    w = sd[f'model.features.{i}.weight'].double()
    b = sd[f'model.features.{i}.bias'].double()
    s, t = bn_affine64(sd, f'model.features.{i + 1}')
    wf = w * s.view(-1, 1, 1, 1)
    return {'w': f32(conv_fwd_layout(wf)), 'w_bwd': f32(conv_bwd_layout(wf)), 'b': f32(b * s + t)}


def fold_vgg_head(sd: SD, feat_channels: int, feat_hw: int, chunk: int = 2048) -> dict:
    """
    AdaptiveAvgPool2d((7,7)) + flatten + Linear(d,d,bias=False) + BatchNorm1d folded into one [d][f*f*C] matrix whose
    input is the NHWC-flattened f x f feature map (exact: the pool is linear).  src/classifier/model.py:39-45.
    """
    w0 = sd['model.classifier.0.weight']                                     # [d, C*49], input index c*49 + oh*7 + ow
    d = w0.shape[0]
    C, f = feat_channels, feat_hw
    A = adaptive_avgpool_matrix(f, 7)                                        # [7, f]
    K = torch.einsum('ai,bj->abij', A, A).reshape(49, f * f)                 # [(oh,ow), (ih,iw)]
    s, t = bn_affine64(sd, 'model.classifier.1')
    big = d * C * 49 > (1 << 26)
    Kc = K.float() if big else K
    out = torch.empty(d, f * f * C, dtype=torch.float32)
    for r0 in range(0, d, chunk):
        blk = w0[r0:r0 + chunk].reshape(-1, C, 49)
        blk = blk.float() if big else blk.double()
        fold = torch.matmul(blk, Kc)                                         # [r, C, f*f]
        fold = fold.permute(0, 2, 1).reshape(blk.shape[0], f * f * C)        # NHWC flatten: (ih*f+iw)*C + c
        out[r0:r0 + chunk] = (fold * s[r0:r0 + chunk].to(fold.dtype).view(-1, 1)).float()
    w3 = sd['model.classifier.3.weight']
    return {'w_head': out, 'w_head_bwd': out.t().contiguous(), 'b_head': f32(t),
            'w_out': f32(w3), 'w_out_bwd': f32(w3.t()), 'b_out': f32(sd['model.classifier.3.bias'])}


def nf_constant_shift(sd: SD, key: str, num_nf_cells: int, num_latent: int) -> torch.Tensor:
    """
    Normalizing-flow cells on the purification path (models.py:209-210, 253-254): NFCell(z) = z - layers(z)
    (architecture.py:238-239).  The last conv of `layers` is a 1x1 MaskedConv2d built with zero_diag=False, whose mask
    keeps (1*1)//2 + 0 = 0 taps (architecture.py:20-23): its output is its bias for every input, so each cell subtracts a
    per-channel constant and a whole group's flow is  z -> z - c,  c = sum of those biases.  Returned as float64 [NL];
    raises if a checkpoint's masks do not have that form (a genuine flow would need its own kernels).
    """
    c = torch.zeros(num_latent, dtype=torch.float64)
    for n in range(num_nf_cells):
        for cell in ('cell1', 'cell2'):
            p = f'nf_cells.nf_{key}.{n}.{cell}.layers.4'
            if float(sd[f'{p}.mask'].abs().sum()) != 0.0:
                raise NotImplementedError('normalizing-flow cell with a non-empty 1x1 mask: not built')
            c += sd[f'{p}.bias'].double()
    return c


# ---------------------------------------------------------------------------------------------------------------
# ResNet-50 (src/classifier/model.py:10-28; torchvision topology restated in resnet_spec.py)
# ---------------------------------------------------------------------------------------------------------------

def _conv_bn64(sd: SD, conv: str, bn: str) -> Tuple[torch.Tensor, torch.Tensor]:
    """bias-free conv followed by eval-mode BatchNorm2d -> (folded weights [Cout,Cin,k,k] float64, bias [Cout])"""
    s, t = bn_affine64(sd, bn)
    return sd[f'{conv}.weight'].double() * s.view(-1, 1, 1, 1), t


def fold_resnet_stem(sd: SD, ld: int) -> dict:
    """7x7 / stride 2 / pad 3 stem as a 4x4 / stride 1 convolution over the space-to-depth image (ga_image_io s2d):
    out[i] = sum_kh x[2i + kh - 3] w[kh]; with kh' = kh + 1 = 2q + r (a zero tap in front), x[2(i + q - 2) + r] is phase r
    of the image at position i + q - 2: four taps q with 'pad' 2 per axis, 4*ld input channels ((r_h*2 + r_w)*ld + c)."""
    w, b = _conv_bn64(sd, 'model.conv1', 'model.bn1')                          # [Cout, 3, 7, 7]
    co = w.shape[0]
    w8 = torch.zeros(co, 3, 8, 8, dtype=w.dtype)
    w8[:, :, 1:, 1:] = w
    w4 = torch.zeros(co, 2, 2, ld, 4, 4, dtype=w.dtype)                       # [co, r_h, r_w, c, q_h, q_w]
    w4[:, :, :, :3] = w8.view(co, 3, 4, 2, 4, 2).permute(0, 3, 5, 1, 2, 4)    # kh' = 2 q_h + r_h, kw' = 2 q_w + r_w
    w4 = w4.reshape(co, 4 * ld, 4, 4)
    return {'w': f32(conv_fwd_layout(w4)), 'w_bwd': f32(conv_bwd_layout(w4)), 'b': f32(b)}


def grouped_bwd_weights(w: torch.Tensor, groups: int) -> torch.Tensor:
    """[C, cg, k, k] grouped forward weights -> the grouped weights of the backward-to-input conv, same shape: within each
    group input and output channels swap, the kernel flips spatially."""
    c, cg, k, _ = w.shape
    return w.view(groups, c // groups, cg, k, k).transpose(1, 2).flip(3, 4).reshape(c, cg, k, k).contiguous()


def grouped_subpixel_weights(w: torch.Tensor, groups: int) -> dict:
    """subpixel_weights per group, stacked: {(a, b): ([C][taps*cg] in ga_gconv layout, KH, KW)}"""
    c, cg, k, _ = w.shape
    per = [subpixel_weights(w[g * cg:(g + 1) * cg]) for g in range(groups)]        # each: [cg_in][taps*cg_out]
    return {ab: (torch.cat([p[ab][0] for p in per], dim=0).contiguous(), per[0][ab][1], per[0][ab][2]) for ab in per[0]}


def fold_resnet_block(sd: SD, blk) -> dict:
    p = blk.prefix
    w1, b1 = _conv_bn64(sd, f'{p}.conv1', f'{p}.bn1')
    w2, b2 = _conv_bn64(sd, f'{p}.conv2', f'{p}.bn2')
    w3, b3 = _conv_bn64(sd, f'{p}.conv3', f'{p}.bn3')
    out = {'w1': f32(conv_fwd_layout(w1)), 'w1_bwd': f32(conv_bwd_layout(w1)), 'b1': f32(b1),
           'w2': f32(conv_fwd_layout(w2)), 'b2': f32(b2),          # grouped: [C][9*cg], the ga_gconv layout
           'w3': f32(conv_fwd_layout(w3)), 'w3_bwd': f32(conv_bwd_layout(w3)), 'b3': f32(b3)}
    if blk.groups > 1:
        if blk.stride == 1:
            out['w2_bwd'] = f32(conv_fwd_layout(grouped_bwd_weights(w2, blk.groups)))
        else:
            for (a, b), (wm, kh, kw) in grouped_subpixel_weights(w2, blk.groups).items():
                out[f'w2_sub{a}{b}'] = wm
    elif blk.stride == 1:
        out['w2_bwd'] = f32(conv_bwd_layout(w2))
    else:
        for (a, b), (wm, kh, kw) in subpixel_weights(w2).items():
            out[f'w2_sub{a}{b}'] = wm
    if blk.downsample:
        wd, bd = _conv_bn64(sd, f'{p}.downsample.0', f'{p}.downsample.1')
        out['wd'], out['bd'] = f32(conv_fwd_layout(wd)), f32(bd)
        if blk.stride == 1:
            out['wd_bwd'] = f32(conv_bwd_layout(wd))
        else:
            for (a, b), (wm, kh, kw) in subpixel_weights(wd).items():
                out[f'wd_sub{a}{b}'] = wm
    return out


def fold_resnet_head(sd: SD) -> dict:
    """Linear(d, d, bias=False) + BatchNorm1d folded, then Linear(d, n) — src/classifier/model.py:19-24."""
    s, t = bn_affine64(sd, 'model.fc.1')
    w0 = sd['model.fc.0.weight'].double() * s.view(-1, 1)
    w3 = sd['model.fc.3.weight']
    return {'w_h': f32(w0), 'w_h_bwd': f32(w0.t()), 'b_h': f32(t),
            'w_o': f32(w3), 'w_o_bwd': f32(w3.t()), 'b_o': f32(sd['model.fc.3.bias'])}


# ---------------------------------------------------------------------------------------------------------------
# e4e encoder (encoding/encoder.py, encoding/helpers.py; spec in e4e_spec.py)
# ---------------------------------------------------------------------------------------------------------------

def fold_e4e_input(sd: SD, ld: int) -> dict:
    """input_layer: Conv2d(3, 64, 3, 1, 1, bias=False) + BatchNorm2d folded; PReLU slopes kept apart (encoder.py:72-74)"""
    w, b = _conv_bn64(sd, 'input_layer.0', 'input_layer.1')
    out = pad_image_conv({'w': f32(conv_fwd_layout(w)), 'w_bwd': f32(conv_bwd_layout(w)), 'b': f32(b)}, 3, ld)
    out['slope'] = f32(sd['input_layer.2.weight'])
    return out


def fold_ir_se_unit(sd: SD, u) -> dict:
    """bottleneck_IR_SE (helpers.py:97-119): BN0 stays a prologue affine of conv1 (zero padding applies AFTER the BN, so its
    shift cannot move into a bias), BN4 folds into conv2, the SE FCs are bias-free 1x1 convs, the conv shortcut folds its BN."""
    p = u.prefix
    s0, t0 = bn_affine64(sd, f'{p}.res_layer.0')
    w1 = sd[f'{p}.res_layer.1.weight'].double()
    w2, b2 = _conv_bn64(sd, f'{p}.res_layer.3', f'{p}.res_layer.4')
    hid = sd[f'{p}.res_layer.5.fc1.weight'].shape[0]
    out = {'pro_scale': f32(s0), 'pro_shift': f32(t0),
           'w1': f32(conv_fwd_layout(w1)), 'w1_bwd': f32(conv_bwd_layout(w1)),
           'slope': f32(sd[f'{p}.res_layer.2.weight']),
           'w2': f32(conv_fwd_layout(w2)), 'b2': f32(b2),
           'se_w1': f32(sd[f'{p}.res_layer.5.fc1.weight'][:, :, 0, 0]), 'se_b1': torch.zeros(hid),
           'se_w2': f32(sd[f'{p}.res_layer.5.fc2.weight'][:, :, 0, 0]), 'se_b2': torch.zeros(u.depth)}
    if u.stride == 1:
        out['w2_bwd'] = f32(conv_bwd_layout(w2))
    else:
        for (a, b), (wm, kh, kw) in subpixel_weights(w2).items():
            out[f'w2_sub{a}{b}'] = wm
    if u.cin != u.depth:
        ws, bs = _conv_bn64(sd, f'{p}.shortcut_layer.0', f'{p}.shortcut_layer.1')
        out['ws'], out['bs'] = f32(conv_fwd_layout(ws)), f32(bs)
        if u.stride == 1:
            out['ws_bwd'] = f32(conv_bwd_layout(ws))
        else:
            for (a, b), (wm, kh, kw) in subpixel_weights(ws).items():
                out[f'ws_sub{a}{b}'] = wm
    return out


def fold_e4e_lateral(sd: SD, name: str) -> dict:
    w = sd[f'{name}.weight'].double()
    return {'w': f32(conv_fwd_layout(w)), 'w_bwd': f32(conv_bwd_layout(w)), 'b': f32(sd[f'{name}.bias'])}


def fold_e4e_style(sd: SD, j: int, pools: int) -> dict:
    """GradualStyleBlock (encoder.py:33-54): `pools` stride-2 3x3 convs with bias (LeakyReLU between: a consumer prologue)
    and the EqualLinear (lr_mul = 1: weight * 1/sqrt(in), bias as is; generator.py:85-98).  Per conv: forward weights,
    sub-pixel backward kernels, and the centre tap as a 1x1 conv for maps that are already 1x1 (small test inputs)."""
    out = {}
    for k in range(pools):
        w = sd[f'styles.{j}.convs.{2 * k}.weight'].double()
        out[f'w{k}'], out[f'b{k}'] = f32(conv_fwd_layout(w)), f32(sd[f'styles.{j}.convs.{2 * k}.bias'])
        for (a, b), (wm, kh, kw) in subpixel_weights(w).items():
            out[f'w{k}_sub{a}{b}'] = wm
        wc = w[:, :, 1:2, 1:2]
        out[f'w{k}_c'], out[f'w{k}_c_bwd'] = f32(conv_fwd_layout(wc)), f32(conv_bwd_layout(wc))
    wl = sd[f'styles.{j}.linear.weight'].double()
    wl = wl * (1.0 / (wl.shape[1] ** 0.5))
    out['wl'], out['wl_bwd'], out['bl'] = f32(wl), f32(wl.t()), f32(sd[f'styles.{j}.linear.bias'])
    return out


# ------------------------------------------------------------------------------------------------------------------
# StyleGAN2 modulated convolution (StyleGan_E4E/stylegan2/generator.py:108-207)
# ------------------------------------------------------------------------------------------------------------------
def fold_styled_conv(sd: SD, spec, noise: torch.Tensor = None, cout_pad: int = 0) -> dict:
    """ModulatedConv2d without resampling, rewritten so that the per-sample weights never exist:
        weight[n] = scale * W * s[n, ci] * demod[n, co]   (generator.py:166-176)
      ==> out[n] = demod[n] * conv(scale * W, x[n] * s[n]),   demod[n, co] = rsqrt(sum_ci W2[co, ci] s[n, ci]^2 + 1e-8)
    with W2 = scale^2 * sum_taps W^2.  Returns the shared conv weights (forward / backward layouts), W2 (both layouts), the
    modulation EqualLinear (lr_mul = 1: weight / sqrt(D), bias as is; generator.py:85-98) and the row-independent additive
    term of the layer's tail: noise.weight * noise[p] + activate.bias[c] (StyledConv, generator.py:258-265, fixed noise
    buffers) or ToRGB's bias (generator.py:282-283).  cout_pad > 0 pads Cout with zero filters (ToRGB: 3 -> 4 lanes)."""
    p = spec.prefix
    w = sd[f'{p}.conv.weight'][0].double()                           # [Cout, Cin, k, k]
    w = w * (1.0 / (w.shape[1] * w.shape[2] * w.shape[3]) ** 0.5)
    co = w.shape[0]
    if cout_pad > co:
        w = torch.cat([w, w.new_zeros(cout_pad - co, *w.shape[1:])], dim=0)
    wm = sd[f'{p}.conv.modulation.weight'].double()
    wm = wm * (1.0 / wm.shape[1] ** 0.5)
    out = {'w': f32(conv_fwd_layout(w)), 'w_bwd': f32(conv_bwd_layout(w)), 'w64': w,
           'wm': f32(wm), 'wm_bwd': f32(wm.t()), 'bm': f32(sd[f'{p}.conv.modulation.bias'])}
    if spec.demodulate:
        w2 = w.pow(2).sum(dim=(2, 3))                                # [Cout, Cin]
        out['w2'], out['w2_bwd'] = f32(w2), f32(w2.t())
    P = spec.res * spec.res
    if spec.activate:
        add = sd[f'{p}.activate.bias'].double().view(1, -1).expand(P, -1).clone()
        if noise is not None:
            add = add + sd[f'{p}.noise.weight'].double() * noise.double().reshape(P, 1)
    else:
        add = sd[f'{p}.bias'].double().view(1, -1).expand(P, -1).clone()
    if cout_pad > co:
        add = torch.cat([add, add.new_zeros(P, cout_pad - co)], dim=1)
    out['add'] = f32(add)
    return out


def upsample_conv_weights(w: torch.Tensor) -> dict:
    """StyledConv(upsample=True): conv_transpose2d(stride 2, no padding) followed by the [1,3,3,1] (x4) blur with pad (1, 1)
    (generator.py:133-139,178-189) is ONE linear map: a stride-2 transposed convolution with the 6x6 kernel
        G[d, e] = sum_{i,j} W[i, j] K[i + 1 - d, j + 1 - e],   d, e in [-2, 3]      (K = outer(k, k) / 16, k = [1,3,3,1])
        out[2U+a, 2V+b] = sum_{dy,dx in {-1,0,1}} z[U+dy, V+dx] G[a - 2 dy, b - 2 dx]
    i.e. four ordinary 3x3 / pad 1 convolutions over the low-resolution input, one per output parity (a, b), and — for the
    backward-to-input — either one 6x6 / stride 2 / pad 2 convolution over the high-resolution cotangent ('up_bwd') or the
    sum of the four parity convs' adjoints over the de-interleaved cotangent ('up_bwd{a}{b}': 3x3 / pad 1, the form the engine
    uses: 36 taps exceed the split-bf16 kernel's tap mask, 4 x 9 do not).
    w: [Cout, Cin, 3, 3] (already scaled).  Returns {'up{a}{b}': [Cout][9*Cin], 'up_bwd{a}{b}': [Cin][9*Cout], 'up_bwd': [Cin][36*Cout]}."""
    co, ci = w.shape[:2]
    k1 = torch.tensor([1.0, 3.0, 3.0, 1.0], dtype=torch.float64)
    K = torch.outer(k1, k1) / 16.0
    G = w.new_zeros(co, ci, 6, 6)                                   # index = d + 2
    for i in range(3):
        for j in range(3):
            for a in range(4):
                for b in range(4):
                    G[:, :, i + 1 - a + 2, j + 1 - b + 2] += w[:, :, i, j] * K[a, b]
    out = {}
    for a in (0, 1):
        for b in (0, 1):
            wp = w.new_zeros(co, ci, 3, 3)
            for ky in range(3):
                for kx in range(3):
                    wp[:, :, ky, kx] = G[:, :, a - 2 * (ky - 1) + 2, b - 2 * (kx - 1) + 2]
            out[f'up{a}{b}'] = f32(conv_fwd_layout(wp))
            out[f'up_bwd{a}{b}'] = f32(conv_bwd_layout(wp))        # adjoint of the parity conv: over the de-interleaved cotangent
    out['up_bwd'] = f32(conv_fwd_layout(G.permute(1, 0, 2, 3)))
    # all four parities as ONE 3x3 conv Cin -> 4*Cout (output channel block 2a+b = parity (a, b): the depth-to-space form of the
    # up-sampled map) and its adjoint 4*Cout -> Cin: the input window is gathered / split once for all parities
    wall = torch.cat([out[f'up{a}{b}'].double().view(co, 3, 3, ci).permute(0, 3, 1, 2) for a in (0, 1) for b in (0, 1)], dim=0)
    out['up_all'], out['up_all_bwd'] = f32(conv_fwd_layout(wall)), f32(conv_bwd_layout(wall))
    return out


def fold_mapping(sd: SD, n_mlp: int, lr_mul: float) -> dict:
    """Generator.style's EqualLinear stack (generator.py:85-92,306-317): weight * (lr_mul / sqrt(in)), bias * lr_mul"""
    out = {}
    for k in range(1, n_mlp + 1):
        w = sd[f'style.{k}.weight'].double()
        out[f'w{k}'] = f32(w * (lr_mul / w.shape[1] ** 0.5))
        out[f'b{k}'] = f32(sd[f'style.{k}.bias'].double() * lr_mul)
    return out


def fold_trans_layer(sd: SD, name: str) -> dict:
    """TransformerDecoderLayer (StyleGan_Trans/models/transformer.py:17-100): nn.MultiheadAttention's packed in_proj split into
    what each GEMM needs — self-attention projects q, k, v of the SAME tokens in one GEMM ([3d, d]); cross-attention projects the
    queries ([d, d]) and the memory tokens (k and v together, [2d, d]) separately — plus out_proj, the FFN and the LayerNorm
    affine parameters; every matrix also transposed for the backward-to-input GEMMs."""
    out = {}
    d = sd[f'{name}.self_attn.out_proj.weight'].shape[0]

    def lin(key, w, b):
        w = w.double()
        out[f'{key}_w'], out[f'{key}_w_bwd'], out[f'{key}_b'] = f32(w), f32(w.t().contiguous()), f32(b)
    lin('sa_qkv', sd[f'{name}.self_attn.in_proj_weight'], sd[f'{name}.self_attn.in_proj_bias'])
    lin('sa_out', sd[f'{name}.self_attn.out_proj.weight'], sd[f'{name}.self_attn.out_proj.bias'])
    wi, bi = sd[f'{name}.multihead_attn.in_proj_weight'], sd[f'{name}.multihead_attn.in_proj_bias']
    lin('ca_q', wi[:d], bi[:d])
    lin('ca_kv', wi[d:], bi[d:])
    lin('ca_out', sd[f'{name}.multihead_attn.out_proj.weight'], sd[f'{name}.multihead_attn.out_proj.bias'])
    lin('ff1', sd[f'{name}.linear1.weight'], sd[f'{name}.linear1.bias'])
    lin('ff2', sd[f'{name}.linear2.weight'], sd[f'{name}.linear2.bias'])
    for k in (1, 2, 3):
        out[f'ln{k}_g'], out[f'ln{k}_b'] = f32(sd[f'{name}.norm{k}.weight']), f32(sd[f'{name}.norm{k}.bias'])
    return out


def trans_queries(sd: SD, gsd: SD, n_mlp: int, lr_mul: float) -> torch.Tensor:
    """query = decoder.style(encoder.z) (style_transformer.py:61-66; src/defenses/ours/models.py:311-316): the mapping network
    (PixelNorm + n_mlp x EqualLinear(lr_mul, fused leaky ReLU); stylegan2/model.py) on the LEARNED z — a constant of the
    checkpoint, evaluated once at load time in float64.  Returns [n_query, d]."""
    z = sd['z'][0].double()
    h = z * torch.rsqrt(torch.mean(z ** 2, dim=-1, keepdim=True) + 1e-8)
    for k in range(1, n_mlp + 1):
        w = gsd[f'style.{k}.weight'].double()
        h = h @ (w * (lr_mul / w.shape[1] ** 0.5)).t() + gsd[f'style.{k}.bias'].double() * lr_mul
        h = torch.where(h > 0, h, 0.2 * h) * 2 ** 0.5
    return f32(h)
