"""
Host-side plan builder: turns (NVAE state dict, VGG state dict, row count) into two flat op lists — forward and
backward-to-input — over caller-visible device buffers, replayed by libga_ops' ga_plan_run.

What the plans compute (reference call chain):
  EoTWrapper.forward (src/defenses/wrappers.py:15-24)              image_io: repeat + noise + clamp, NCHW->NHWC
  MLVGMDefenseModel.__call__ (src/defenses/ours/abstract_models.py:161-193)
  NVAEDefenseModel.purify (src/defenses/ours/models.py:160-274)    conv / dwconv5 / SE / sampler / DML ops
  BaseClassificationModel.__call__ + Vgg (abstract_models.py:53-62; src/classifier/model.py:31-49)

PyTorch is used for device memory and the current stream only; every arithmetic op of the path is a HIP kernel.
"""
from __future__ import annotations

import ctypes as C
import json
import os
from typing import Dict, List, Optional, Sequence

import torch

from . import _lib as L
from . import folding as F
from .nvae_spec import DecCellSpec, EncCellSpec, NVAESpec, build_spec
from .e4e_spec import E4ESpec
from .resnet_spec import ResNetSpec
from .vgg_spec import VggSpec
from .engine_core import (IMG_LD, RES_SCALE, TUNE_FILE, WS_FLOATS, Act, WeightStore, _ptr, conv_key,  # noqa: F401
                          tune_cache)
from .engine_classifiers import ClassifierBuilder
from .engine_e4e import E4EBuilder
from .engine_avae import AvaeBuilder
from .engine_ndvae import NdvaeBuilder
from .engine_nvae import NvaeBuilder
from .engine_stylegan import StyleGanBuilder
from .engine_trans import TransBuilder


class Engine(NvaeBuilder, NdvaeBuilder, AvaeBuilder, ClassifierBuilder, E4EBuilder, StyleGanBuilder, TransBuilder):
    # decoder cells whose shape ga_dec_cell takes run as one fused launch per direction (GA_FUSE_DEC_CELL=0: the three
    # unfused launches, same numbers bit for bit — kept for A/B profiles and for the shapes the fused kernel refuses)
    fuse_dec_cells = os.environ.get('GA_FUSE_DEC_CELL', '1') != '0'
    # ... and only when the launch fills the chip: a fused workgroup walks the hidden width serially (115 us at 128 channels, 220 us at
    # 256, whatever the row count up to 256 workgroups), the three unfused launches scale with the rows: below ~160 workgroups they win
    # (the reference protocol of one image x EoT 32 would pay 10 ms of a 21 ms step).  GA_FUSE_DEC_CELL=force: always (tests).
    fuse_min_workgroups = 0 if os.environ.get('GA_FUSE_DEC_CELL') == 'force' else 160
    # ga_dec_cell_halo (8 x 16 tiles with a recomputed halo for the few-channel post-processing cells): parity-green, forward 0.63 vs
    # 0.95 ms and 0.32 vs 0.57 ms per 512-row chunk, but its backward needs a two-pixel-deeper halo — 2.9x the depthwise work, LDS-bound —
    # and measures 2.44 vs 1.22 ms / 1.29 vs 0.63 ms: OFF unless GA_FUSE_HALO_CELL=1
    fuse_halo_cells = os.environ.get('GA_FUSE_HALO_CELL', '0') == '1'
    # ga_dec_cell variant for the 128-channel cells: 0 = four waves per workgroup, 1 = eight (two per SIMD); same results
    dec_cell_variant = int(os.environ.get('GA_DEC_CELL_VARIANT', '1'))

    def __init__(self, nvae_sd, nvae_cfg: dict, resolution, vgg_sd, vgg_spec: VggSpec, rows: int, rep: int,
                 alphas: Sequence[float], temperature: float = 0.6, noise_eps: float = 0.0,
                 device: str = 'cuda:0', need_backward: bool = True, dry_run: bool = False,
                 store: Optional[WeightStore] = None, precision: str = 'bf16x3', blur: bool = False,
                 share_encoder: bool = False, cot_rep: int = 1):
        """cot_rep = K > 1: the backward plan takes K cotangents per forward row in ONE replay (the per-class
        backward loops of DeepFool / FAB, src/attacks/untargeted.py:526-560, :605-635).  `dlogits` is then [rows * K, classes]
        (cotangent k of defender row r at row r * K + k) and `dx` [images * K, 3, H, W] (gradient k of image b at row b * K + k);
        every saved activation is read by its K cotangent rows, nothing is recomputed or copied."""
        if rows % rep:
            raise ValueError('rows must be a multiple of the EoT repeat')
        self.device = torch.device(device)
        self.dry_run = dry_run
        if self.device.type != 'cuda' and not dry_run:
            raise RuntimeError('the HIP engine needs a GPU device; there is no CPU fallback '
                               '(dry_run=True only builds and validates the plans)')
        if precision not in ('fp32', 'bf16x3'):
            raise ValueError("precision must be 'fp32' (exact f32 MFMA) or 'bf16x3' (3 bf16 MFMAs per product)")
        self.precision = precision
        self.store = store if store is not None else WeightStore(self.device)
        self.has_nvae = nvae_sd is not None
        self.spec: Optional[NVAESpec] = build_spec(nvae_cfg, resolution) if self.has_nvae else None
        self.resolution = tuple(resolution)
        self.vspec = vgg_spec
        self.rows, self.rep = rows, rep
        self.alphas = [float(a) for a in alphas]
        if self.has_nvae and len(self.alphas) != len(self.spec.groups):
            raise ValueError(f'{len(self.spec.groups)} interpolation alphas expected, got {len(self.alphas)}')
        self.temperature = float(temperature)
        self.noise_eps = float(noise_eps)
        self.blur = bool(blur)
        # EoT replicas of one image are identical until randomness enters.  Without input noise the whole encoder
        # (pre-processing + encoder tower + encoder_0 + sampler_0:0, ~57 % of the FLOPs) sees `rep` identical rows per
        # image: with share_encoder it runs once per IMAGE and its feature maps are read by all replicas of the decoder
        # (gradients summed over the replicas).  Same numbers as the literal x.repeat(eot) path, row for row.
        self.share_encoder = bool(share_encoder) and rep > 1 and self.noise_eps == 0.0 and nvae_sd is not None
        self.enc_rows = rows // rep if self.share_encoder else rows
        self.need_backward = need_backward
        if cot_rep < 1 or (cot_rep > 1 and not (self.has_nvae or isinstance(vgg_spec, VggSpec))):
            raise ValueError('cot_rep > 1 (K-cotangent backward) is built for the NVAE + VGG defender and the VGG classifier')
        self.cot_rep = int(cot_rep)
        self.bytes = 0
        self.acts = {}                       # name -> Act (debugging / tests)
        self.version = 0                     # bumped by callers after each forward (stale-backward detection)
        self._sampler_descs = []             # (desc, latent index): alphas can be changed without rebuilding
        self._keep = []                      # weights etc.
        self._frag_ok = {}                   # id(ConvDesc) -> weight tensor of the convs that may run on tile 8
        self._thin_ok = {}                   # ... on tile 11
        self.fwd = L.Plan()
        self.bwd = L.Plan()
        self._bwd_steps = []                 # closures emitting backward ops, replayed in reverse
        self._build(nvae_sd, vgg_sd)

    @classmethod
    def bare(cls, rows: int, device='cuda:0', precision: str = 'bf16x3', store: Optional[WeightStore] = None,
             dry_run: bool = False, rep: int = 1, resolution=None, alphas: Sequence[float] = (), noise_eps: float = 0.0,
             need_backward: bool = True, blur: bool = False, share_encoder: bool = False) -> "Engine":
        """An engine with empty plans: building blocks that are not yet part of a full defender (the StyleGAN2 layers of
        engine_stylegan.py) are emitted into it by their builders and closed with `finish()`; forward() / backward() then
        replay the plans as for a full engine."""
        self = cls.__new__(cls)
        self.device, self.dry_run, self.precision = torch.device(device), dry_run, precision
        if self.device.type != 'cuda' and not dry_run:
            raise RuntimeError('the HIP engine needs a GPU device; there is no CPU fallback')
        self.store = store if store is not None else WeightStore(self.device)
        if rows % rep:
            raise ValueError('rows must be a multiple of the EoT repeat')
        self.has_nvae, self.spec, self.vspec, self.resolution = False, None, None, (tuple(resolution) if resolution else None)
        self.rows, self.rep, self.alphas, self.temperature = rows, rep, [float(a) for a in alphas], 1.0
        self.noise_eps, self.blur = float(noise_eps), bool(blur)
        # EoT replicas are identical until randomness enters: without input noise an encoder in front of the first random
        # draw runs once per image (see Engine.__init__); builders that support it read share_encoder / enc_rows
        self.share_encoder = bool(share_encoder) and rep > 1 and self.noise_eps == 0.0
        self.enc_rows = rows // rep if self.share_encoder else rows
        self.need_backward, self.image_s2d, self.cot_rep = need_backward, False, 1
        self.bytes, self.acts, self.version, self._sampler_descs, self._keep, self._frag_ok, self._thin_ok = 0, {}, 0, [], [], {}, {}
        self.fwd, self.bwd, self._bwd_steps, self._scratch = L.Plan(), L.Plan(), [], {}
        self.eps, self.purified, self.dpurified, self._purified_grad_nhwc = [], None, None, None
        return self

    def finish(self):
        self._finish(0)
        return self

    # ------------------------------------------------------------------------------------------------ memory
    def alloc(self, shape) -> torch.Tensor:
        t = torch.zeros(shape, dtype=torch.float32, device=self.device)
        self.bytes += t.numel() * 4
        self._keep.append(t)     # plans hold raw pointers: every buffer lives as long as the engine
        return t

    def dev(self, t: torch.Tensor) -> torch.Tensor:
        t = t.to(self.device, dtype=torch.float32).contiguous()
        self._keep.append(t)
        self.bytes += t.numel() * 4
        return t

    def devd(self, key: str, fn) -> dict:
        """folded weights by key, computed + uploaded once per model (WeightStore)."""
        return self.store.get(key, fn)

    # ------------------------------------------------------------------------------------------------ op emitters
    def conv(self, plan, name, x, w, y, *, cin=None, cout=None, bias=None, K=1, sn=1, sd=1, pad=0,
             x2=None, pro_scale=None, pro_shift=None, pro_act=0, pro_per_row=0,
             addend=None, addend_bcast=False, addend2=None, dact_x=None, dact_scale=None, dact_shift=None, dact_act=0,
             in_hw=None, out_hw=None, n=None, KH=None, KW=None, anchored=False, explicit_out=False,
             ldx=None, ldy=None, ldadd=None, flags=0, precise=False):
        """x, x2, y, addend*, dact_x are torch tensors [N,H,W,C] (or Act.t); shapes are taken from them.
        precise: run this contraction on the exact fp32 MFMA kernel even in 'bf16x3' mode (small GEMMs whose results feed an
        ill-conditioned step: the q / k projections in front of a softmax over thousands of keys)."""
        d = L.ConvDesc()
        N, Hi, Wi, Cx = x.shape
        d.x, d.ldx = _ptr(x), (ldx or Cx)
        d.C1 = cin if cin is not None else Cx
        if x2 is not None:
            d.x2, d.ldx2, d.C2 = _ptr(x2), x2.shape[3], x2.shape[3]
        d.w, d.bias = _ptr(w), _ptr(bias)
        if self.precision == 'bf16x3' and not self.dry_run and not precise:
            hi, lo = self.store.split(w)
            d.w_hi, d.w_lo = _ptr(hi), _ptr(lo)
            kh, kw = (KH or K), (KW or K)
            if (kh, kw, sn, sd, pad) == (3, 3, 1, 1, 1) and x2 is None and (cin or x.shape[3]) % 32 == 0 and 128 % x.shape[2] == 0 \
                    and w.dim() == 2 and w.shape[1] == 9 * (cin or x.shape[3]):
                # tile 8 reads its B fragments from a fragment-ordered copy of the weights (WeightStore.frag3).  The copy is built
                # LAZILY — by apply_tuning for the descs whose tuned tile is 8, by autotune for its candidates — not for every
                # eligible 3x3 weight (52 of 1594 tuned shapes select tile 8; an eager copy doubled the split-weight memory)
                self._frag_ok[id(d)] = w
            if (kh, kw, sn, sd, pad) == (3, 3, 1, 1, 1) and x2 is None and (cin or x.shape[3]) in (32, 64) and x.shape[1] % 8 == 0 \
                    and x.shape[2] % 16 == 0 and w.dim() == 2 and w.shape[1] == 9 * (cin or x.shape[3]):
                self._thin_ok[id(d)] = w                    # tile 11 (conv_thin3): persistent weights-resident kernel, its own fragment order
        d.pro_scale, d.pro_shift, d.pro_act, d.pro_per_row = _ptr(pro_scale), _ptr(pro_shift), pro_act, pro_per_row
        No, Ho, Wo, Cy = y.shape
        d.y, d.ldy = _ptr(y), (ldy or Cy)
        d.flags = flags
        d.Cout = cout if cout is not None else Cy
        if addend is not None:
            d.addend, d.ldadd, d.addend_bcast_n = _ptr(addend), (ldadd or addend.shape[-1]), int(addend_bcast)
        if addend2 is not None:
            d.addend2, d.ldadd2 = _ptr(addend2), addend2.shape[-1]
        if dact_x is not None:
            d.dact_x, d.lddact = _ptr(dact_x), dact_x.shape[-1]
            d.dact_scale, d.dact_shift, d.dact_act = _ptr(dact_scale), _ptr(dact_shift), dact_act
            if dact_x.shape[0] != x.shape[0]:           # K cotangent rows per saved activation row (K-cotangent backward plan)
                assert dact_x.shape[0] * self.cot_rep == x.shape[0], (name, tuple(dact_x.shape), tuple(x.shape), self.cot_rep)
                d.dact_rep = self.cot_rep
        d.N, d.Hi, d.Wi, d.Ho, d.Wo = N, Hi, Wi, Ho, Wo
        d.KH, d.KW = (KH or K), (KW or K)
        d.sn, d.sd, d.pad = sn, sd, pad
        assert No == N, (name, x.shape, y.shape)
        assert w.numel() == d.Cout * d.KH * d.KW * (d.C1 + d.C2), (name, tuple(w.shape), d.Cout, d.KH, d.KW, d.C1, d.C2)
        if anchored:            # sub-pixel kernels: window anchored at the output pixel, taps beyond the border masked
            assert (Ho, Wo) == (Hi, Wi) and pad == 0 and sn == 1 and sd == 1, name
        elif explicit_out:      # even-sized kernels (asymmetric padding): the output size is given, border taps are masked
            assert (Ho, Wo) == (Hi, Wi) and sn == 1 and sd == 1, name
        elif sd == 1:
            assert (Hi + 2 * pad - K) // sn + 1 == Ho, (name, Hi, Ho, K, sn, pad)
        else:
            assert Ho in (Hi * sd, Hi * sd - 1) or K == 1, (name, Hi, Ho)
        plan.add(d, name)
        return d

    def grad_conv(self, name, x, w, target: Act, *, primary=None, **kw):
        """Backward GEMM writing (or accumulating) into target.g; `primary` is an extra addend (identity skip)."""
        addend, addend2 = primary, None
        if target.g_written:
            if addend is None:
                addend = target.g
            else:
                addend2 = target.g
        d = self.conv(self.bwd, name, x, w, target.g, addend=addend, addend2=addend2, **kw)
        target.g_written = True
        return d

    def gconv(self, plan, name, x, w, y, cg, *, KH=3, KW=3, stride=1, pad=1, bias=None, pro_act=0, dact_x=None, dact_act=0):
        """grouped convolution with cg channels per group (ga_gconv): x, y NHWC tensors, w [C][KH*KW*cg]"""
        d = L.GconvDesc()
        n, hi, wi, c = x.shape
        _, ho, wo, cy = y.shape
        assert cy == c and c % cg == 0 and w.numel() == c * KH * KW * cg, (name, tuple(x.shape), tuple(y.shape), tuple(w.shape), cg)
        d.x, d.w, d.bias, d.dact_x, d.y = _ptr(x), _ptr(w), _ptr(bias), _ptr(dact_x), _ptr(y)
        d.N, d.Hi, d.Wi, d.Ho, d.Wo, d.C, d.cg = n, hi, wi, ho, wo, c, cg
        d.KH, d.KW, d.stride, d.pad, d.pro_act, d.dact_act = KH, KW, stride, pad, pro_act, dact_act
        plan.add(d, name)
        return d

    def grad_conv_up2(self, name, x, wts, key, target: Act, *, dact_x=None, dact_scale=None, dact_shift=None, dact_act=0, cg=0,
                      dact_prelu=False, ldx=None, **pro):
        """Backward-to-input of a stride-2 conv into target.g (twice the resolution of x) by sub-pixel decomposition:
        one stride-1 ga_conv2d per output parity that meets a tap (folding.subpixel_weights) into dense scratch planes,
        then ga_interleave2, which carries the epilogue (act', accumulation into an already written gradient).
        cg > 0: the conv is grouped with cg channels per group (ga_gconv, folding.grouped_subpixel_weights)."""
        n, h, w, _ = x.shape
        il = L.Interleave2Desc()
        for a in (0, 1):
            for b in (0, 1):
                wm = wts.get(f'{key}{a}{b}')
                if wm is None:
                    continue
                taps = wm.shape[1] // (cg or x.shape[3])            # 1, 2 or 4 taps: 1x1, 2x1 / 1x2, 2x2 windows
                assert taps in (1, 2, 4), (name, tuple(wm.shape), tuple(x.shape))
                kh, kw = 1 + (a if taps >= 2 else 0), 1 + (b if taps >= 2 else 0)
                plane = self.scratch((n, h, w, target.c), f'subpix{a}{b}')
                if cg:
                    self.gconv(self.bwd, f'{name}[{a}{b}]', x, wm, plane, cg, KH=kh, KW=kw, stride=1, pad=0)
                else:
                    self.conv(self.bwd, f'{name}[{a}{b}]', x, wm, plane, KH=kh, KW=kw, pad=0, anchored=True, ldx=ldx, **pro)
                il.s[2 * a + b] = _ptr(plane)
        il.y, il.N, il.H, il.W, il.C = _ptr(target.g), n, 2 * h, 2 * w, target.c
        il.dact_x, il.dact_scale, il.dact_shift, il.dact_act = _ptr(dact_x), _ptr(dact_scale), _ptr(dact_shift), dact_act
        il.dact_prelu = int(dact_prelu)
        if dact_x is not None and dact_x.shape[0] != n:
            assert dact_x.shape[0] * self.cot_rep == n, (name, tuple(dact_x.shape), n)
            il.dact_rep = self.cot_rep
        if target.g_written:
            il.addend = _ptr(target.g)
        self.bwd.add(il, name + '.interleave')
        target.g_written = True

    # rows up to this many elements get their SE merge (out = skip + res_scale * gate * t) from the squeeze / excite launch itself
    # (one workgroup per row re-reads its t while it is hot): 4x4x512 29.6 -> 22.5 us, 8x8x256 33.6 -> 26.8 us per cell; at 16x16x128
    # the two launches are already HBM-bound (50 -> 48 us) and larger images need the wide grid of ga_se_apply
    SE_FUSED_MERGE_MAX = 16384

    def se_forward(self, name, t: Act, wts, P, res_scale=None, merge=None):
        """squeeze + excite; returns (gate, hid) buffers.  merge = (skip tensor or None, out tensor): also emit the merge of
        ga_se_apply (skip_mode 0) from the same launch; the caller checks `se_merges(t, P)` first."""
        n, c = t.n, t.c
        hd = wts['se_w1'].shape[0]
        hid = self.alloc((n, hd))
        gate = self.alloc((n, c))
        if not hasattr(self, 'small_kinks'):
            self.small_kinks = []
        self.small_kinks.append(hid)                        # pre-ReLU hidden units, in call order (parity tests replay their decisions)
        e = L.SeExciteDesc()
        if merge is None and self._se_wide_rows(n, P, c):   # few rows of large images: squeeze over the whole chip first
            m = self.alloc((n, c))
            self._se_reduce(self.fwd, f'{name}.se_squeeze', t.t, None, m, n, P, c, 1.0 / P)
            e.m = _ptr(m)
        else:
            e.t = _ptr(t.t)                                 # fused squeeze + excite (one workgroup per row)
        e.w1, e.b1, e.w2, e.b2 = _ptr(wts['se_w1']), _ptr(wts['se_b1']), _ptr(wts['se_w2']), _ptr(wts['se_b2'])
        e.hid, e.gate, e.N, e.C, e.Hd, e.P, e.res_scale, e.backward = (_ptr(hid), _ptr(gate), n, c, hd, P,
                                                                        RES_SCALE if res_scale is None else res_scale, 0)
        if merge is not None:
            e.skip, e.out = _ptr(merge[0]), _ptr(merge[1])
        self.fwd.add(e, f'{name}.se_gate' + ('+merge' if merge is not None else ''))
        return gate, hid

    @staticmethod
    def _se_wide_rows(n, P, c) -> bool:
        """the fused squeeze of ga_se_excite is one workgroup per row: right for hundreds of small rows (the NVAE cells), 32 workgroups
        for the chip on the IR-SE50 body of the e4e / Style-Transformer defenders (32 - 64 rows of 128^2 x 64 ... 32^2 x 256: 1 - 4 MB per
        row, 190 us per backward launch).  Those rows are reduced by ga_rowchan_reduce (one workgroup per row and 64 channels; from 4096
        pixels its two-stage form, pixels split over workgroups) and the excite kernel takes the reduced vector (its `m` / `dgate` inputs)."""
        return n <= 128 and P >= 1024 and P * c >= 262144 and c % 4 == 0

    def _se_reduce(self, plan, name, a, b, out, n, P, c, scale):
        r = L.ReduceDesc()
        r.a, r.b, r.out, r.N, r.P, r.C, r.scale = _ptr(a), _ptr(b), _ptr(out), n, P, c, scale
        ws = self.scratch((256 * n * c,), 'se_reduce_ws')
        r.ws, r.ws_floats = _ptr(ws), ws.numel()
        plan.add(r, name)

    def se_merges(self, t: Act, P) -> bool:
        return P * t.c <= self.SE_FUSED_MERGE_MAX and t.c % 4 == 0

    def se_backward(self, name, dout: torch.Tensor, t: Act, wts, gate, hid, P, res_scale=None):
        """emits d(gate) reduction + excite backward; returns the per-row prologue (scale, shift) for the next GEMM."""
        n, c = dout.shape[0], t.c                           # cotangent rows (t.n * cot_rep)
        ps = self.scratch((n, c), f'ps{c}')
        pb = self.scratch((n, c), f'pb{c}')
        e = L.SeExciteDesc()
        rs = RES_SCALE if res_scale is None else res_scale
        if self.cot_rep == 1 and self._se_wide_rows(n, P, c):
            dg = self.scratch((n, c), f'se_dgate{c}')       # d(gate) = res_scale * sum_p dout * t, reduced over the whole chip
            self._se_reduce(self.bwd, f'{name}.se_dgate', t.t, dout, dg, n, P, c, rs)
            e.dgate = _ptr(dg)
        else:
            e.act_rep = self.cot_rep
            e.t, e.dout = _ptr(t.t), _ptr(dout)             # fused d(gate) reduction + excite backward
        e.w1, e.b1, e.w2, e.b2 = _ptr(wts['se_w1']), _ptr(wts['se_b1']), _ptr(wts['se_w2']), _ptr(wts['se_b2'])
        e.hid, e.gate, e.pro_scale, e.pro_shift = _ptr(hid), _ptr(gate), _ptr(ps), _ptr(pb)
        e.N, e.C, e.Hd, e.P, e.res_scale, e.backward = n, c, wts['se_w1'].shape[0], P, (RES_SCALE if res_scale is None else res_scale), 1
        self.bwd.add(e, f'{name}.se_gate_bwd')
        return ps, pb

    def scratch(self, shape, key) -> torch.Tensor:
        """Backward-only temporaries that die inside one cell: one buffer per (key, shape)."""
        k = (key, tuple(shape))
        if k not in self._scratch:
            self._scratch[k] = self.alloc(shape)
        return self._scratch[k]

    # ------------------------------------------------------------------------------------------------ build
    def _build(self, nvae_sd, vgg_sd):
        self.nvae_sd = nvae_sd
        self._scratch = {}
        self.image_s2d = (not self.has_nvae) and isinstance(self.vspec, ResNetSpec)
        x0 = self._build_input()
        img = self._build_nvae(x0) if self.has_nvae else x0

        # ---- classifier
        n_nvae_steps = len(self._bwd_steps)
        build = (self._build_resnet if isinstance(self.vspec, ResNetSpec) else
                 self._build_e4e if isinstance(self.vspec, E4ESpec) else self._build_vgg)
        self.logits = build(vgg_sd, img)

        self._finish(n_nvae_steps)

    def _build_input(self) -> Act:
        """caller-visible boundary buffers + the image_io op (EoT repeat, input noise, clamp, NCHW -> NHWC); returns the image Act"""
        R = self.rows
        H = self.resolution[1]

        # ---- boundary buffers (caller-visible)
        self.x_in = self.alloc((R // self.rep, 3, H, H))                    # NCHW images in [0,1]
        self.noise = self.alloc((R, 3, H, H)) if self.noise_eps != 0.0 else None
        self.noise_coef = self.alloc((R,)) if self.noise_eps != 0.0 else None
        self.dx = self.alloc((R // self.rep * self.cot_rep, 3, H, H))      # [image * K + k]: gradient k of image `image`
        self.eps, self.purified, self.dpurified, self._purified_grad_nhwc = [], None, None, None

        # optional Gaussian blur of the input (abstract_models.py:145-159): deterministic, so it is applied to the B
        # images before the EoT repeat; k = 2^(sqrt(H)//2) - 1 taps, sigma 1, reflect border (kornia semantics)
        x_src, dx_dst = self.x_in, self.dx
        if self.blur:
            import math
            k = int(2 ** (math.sqrt(H) // 2) - 1)
            xs = torch.arange(k, dtype=torch.float64) - (k - 1) / 2.0
            g = torch.exp(-xs.pow(2) / 2.0)
            taps = self.devd(f'blur_taps_{k}', lambda: {'g': (g / g.sum()).float()})['g']
            x_src = self.alloc((R // self.rep, 3, H, H))
            dx_dst = self.alloc((R // self.rep * self.cot_rep, 3, H, H))
            # sigma 1: taps beyond 12 pixels are < 1e-31 of the peak (the 255-tap kernel of 256-px images has 25 that matter);
            # planes above ~90 px do not fit LDS and run as two passes through `tmp`
            tmp = self.alloc((R // self.rep, 3, H, H)) if (2 * H * H + k) * 4 > 64 * 1024 else None
            bl = L.BlurDesc()
            bl.x, bl.y, bl.taps, bl.planes, bl.H, bl.W, bl.k, bl.backward = _ptr(self.x_in), _ptr(x_src), _ptr(taps), (R // self.rep) * 3, H, H, k, 0
            bl.radius, bl.tmp = min(12, k // 2), _ptr(tmp)
            self.fwd.add(bl, 'gauss_blur')

            def bwd_blur():
                b = L.BlurDesc()
                b.x, b.y, b.taps, b.planes, b.H, b.W, b.k, b.backward = (_ptr(dx_dst), _ptr(self.dx), _ptr(taps),
                                                                         (R // self.rep) * self.cot_rep * 3, H, H, k, 1)
                b.radius, b.tmp = min(12, k // 2), _ptr(tmp)
                self.bwd.add(b, 'gauss_blur^T')
            self._bwd_steps.append(bwd_blur)

        R0 = self.enc_rows                       # rows entering the network (images when the encoder is shared)
        rep0 = 1 if self.share_encoder else self.rep
        # the NHWC image is kept at a pitch of IMG_LD = 8 channels (3 real + zero pad): the first convolutions then take
        # 16-B loads and the split-bf16 matrix path like every other layer instead of a scalar 3-channel gather
        # (a classifier-only ResNet engine takes the image in space-to-depth form: its 7x7/2 stem is then a 4x4/1 conv)
        x0 = Act(self, R0, H // 2, H // 2, 4 * IMG_LD, 'x0') if self.image_s2d else Act(self, R0, H, H, IMG_LD, 'x0')
        io = L.ImageIoDesc()
        io.x_nchw, io.noise_nchw, io.noise_coef, io.y_nhwc = _ptr(x_src), _ptr(self.noise), _ptr(self.noise_coef), _ptr(x0.t)
        io.N, io.C, io.H, io.W, io.rep, io.backward, io.ld, io.s2d = R0, 3, H, H, rep0, 0, IMG_LD, int(self.image_s2d)
        self.fwd.add(io, 'image_in')

        def bwd_image():
            b = L.ImageIoDesc()
            b.x_nchw, b.noise_nchw, b.noise_coef = _ptr(x_src), _ptr(self.noise), _ptr(self.noise_coef)
            b.dy_nhwc, b.dx_nchw = _ptr(x0.g), _ptr(dx_dst)
            b.N, b.C, b.H, b.W, b.rep, b.backward, b.ld, b.s2d = R0 * self.cot_rep, 3, H, H, rep0, 1, IMG_LD, int(self.image_s2d)
            b.cot_rep = self.cot_rep
            self.bwd.add(b, 'image_in^T')
        self._bwd_steps.append(bwd_image)

        return x0

    def _finish(self, n_nvae_steps: int = 0):
        # ---- emit the backward plan: reverse registration order (classifier part first)
        self.bwd_split = 0
        if self.need_backward:
            for step in reversed(self._bwd_steps[n_nvae_steps:]):
                step()
            self.bwd_split = len(self.bwd)
            for step in reversed(self._bwd_steps[:n_nvae_steps]):
                step()
        self._bwd_steps = None
        # BPDA (Athalye et al. 2018; the attack BASELINE.json names beside PGD): the purifier's Jacobian is replaced by the
        # identity in the backward pass — the cotangent of the classifier's input image goes straight to the defender's input,
        # summed over the EoT replicas (the adjoint of x.repeat(eot)); blur / noise / purify are all skipped.
        self.bpda = None
        pg = self._purified_grad_nhwc
        if self.need_backward and pg is not None:
            s2d = int(pg.c == 4 * IMG_LD)
            hp = pg.h * (2 if s2d else 1)
            if hp == self.resolution[1]:                    # same size in and out (a face_pool to another size has no identity)
                b = L.ImageIoDesc()
                b.x_nchw, b.dy_nhwc, b.dx_nchw = _ptr(self.x_in), _ptr(pg.g), _ptr(self.dx)
                b.N, b.C, b.H, b.W, b.rep, b.backward, b.ld, b.s2d = self.rows * self.cot_rep, 3, hp, hp, self.rep, 1, IMG_LD, s2d
                b.cot_rep = self.cot_rep
                self.bpda = L.Plan()
                self.bpda.add(b, 'bpda_identity^T')
                self.bpda.finalize()
        self.ws = self.alloc((WS_FLOATS,)) if not self.dry_run else None
        self.apply_tuning(tune_cache())

    # ------------------------------------------------------------------------------------------------ tuning
    def _conv_descs(self):
        return [d for plan in (self.fwd, self.bwd) for d in plan.descs if isinstance(d, L.ConvDesc)]

    def apply_tuning(self, cache: dict):
        """set (tile, splits) of every conv from the cache; shapes not in the cache keep the library heuristic."""
        for d in self._conv_descs():
            ent = cache.get(conv_key(d), (0, 1, 1))
            tile, splits = ent[0], ent[1]
            use_bf3 = ent[2] if len(ent) > 2 else 1
            need = splits * d.N * d.Ho * d.Wo * d.Cout
            if splits > 1 and (self.ws is None or need > WS_FLOATS):
                tile, splits = 0, 1
            if tile == 8 and not self._want_frag(d):    # conv_key does not encode pad / Wo / weight layout: a desc that shares the key of
                tile = 5                                # a tile-8 entry without being eligible runs tile 5 (bitwise the same result)
            if tile == 11 and (splits > 1 or not self._want_thin(d)):
                tile, splits = 7, 1                     # (likewise: tile 7 gives tile 11's bits)
            d.tile, d.splits = int(tile), int(splits)
            d.ws, d.ws_floats = (_ptr(self.ws), WS_FLOATS) if splits > 1 else (None, 0)
            if not use_bf3:                      # this shape is faster on the exact fp32 kernel (small K or Cout)
                d.w_hi, d.w_lo = None, None
        self.fwd.finalize()
        self.bwd.finalize()

    def _want_frag(self, d) -> bool:
        """point d.w_frag at the tile-8 fragment copy (built on first use); False when this conv cannot run on tile 8"""
        w = self._frag_ok.get(id(d))
        if w is None or self.dry_run:
            return False
        d.w_frag = _ptr(self.store.frag3(w))
        return True

    def _want_thin(self, d) -> bool:
        """point d.w_frag at the tile-11 fragment copy; False when this conv cannot run on tile 11"""
        w = self._thin_ok.get(id(d))
        if w is None or self.dry_run:
            return False
        d.w_frag = _ptr(self.store.frag_thin(w))
        return True

    def autotune(self, cache: Optional[dict] = None, reps: int = 3, save: Optional[str] = None, verbose: bool = False) -> dict:
        """time every (tile, split-K) candidate of every distinct conv shape on this GPU and keep the fastest."""
        if self.dry_run:
            raise RuntimeError('autotune needs a GPU')
        cache = tune_cache() if cache is None else cache
        stream = self.stream()
        for d in self._conv_descs():
            key = conv_key(d)
            if key in cache:
                continue
            M = d.N * d.Ho * d.Wo
            T = d.KH * d.KW * ((d.C1 + d.C2 + 31) // 32)
            best = None
            modes = (1, 0) if d.w_hi else (0,)
            halo = (5, 6, 7) if (d.w_hi and d.KH == 3 and d.KW == 3 and d.sn == 1 and d.sd == 1 and d.C2 == 0) else ()
            if halo and self._want_frag(d):
                halo = halo + (8,)
            thin = (11,) if (d.w_hi and id(d) in self._thin_ok) else ()
            for use_bf3, tile in [(m_, t_) for m_ in modes for t_ in (1, 2, 3, 4) + ((halo + thin) if m_ else ())]:
                bm, bn = {1: (128, 128), 2: (128, 64), 3: (64, 64), 4: (128, 32), 5: (128, 128), 6: (128, 64), 7: (128, 32), 8: (128, 128),
                          11: (128, 32)}[tile]
                if bn >= 2 * max(32, d.Cout) and tile not in (4, 7):
                    continue
                blocks = -(-M // bm) * -(-d.Cout // bn)
                for splits in (1, 2, 4, 8, 16, 32):
                    if splits > 1 and (blocks * splits > 2048 or T < 2 * splits or splits * M * d.Cout > WS_FLOATS):
                        continue
                    if tile >= 5 and splits > d.C1 // 32:          # the halo kernel splits K over 32-channel chunks
                        continue
                    if tile == 11 and splits > 1:
                        continue
                    t = L.ConvDesc.from_buffer_copy(d)
                    t.tile, t.splits = tile, splits
                    if tile == 11:
                        t.w_frag = _ptr(self.store.frag_thin(self._thin_ok[id(d)]))
                    elif tile == 8:
                        t.w_frag = _ptr(self.store.frag3(self._frag_ok[id(d)]))
                    if not use_bf3:
                        t.w_hi, t.w_lo = None, None
                    t.ws, t.ws_floats = (_ptr(self.ws), WS_FLOATS) if splits > 1 else (None, 0)
                    try:
                        L.run(t, stream)
                    except L.GaError:                               # this kernel does not take the shape
                        continue
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(reps):
                        L.run(t, stream)
                    e1.record()
                    e1.synchronize()
                    ms = e0.elapsed_time(e1) / reps
                    if best is None or ms < best[0]:
                        best = (ms, tile, splits, use_bf3)
            cache[key] = (best[1], best[2], best[3])
            if verbose:
                print(f'tune {key}: tile {best[1]} splits {best[2]} {best[0] * 1e3:.1f} us bf3 {best[3]}', flush=True)
        self.apply_tuning(cache)
        if save:
            with open(save, 'w') as f:
                json.dump(cache, f, indent=0, sort_keys=True)
        return cache

    # ------------------------------------------------------------------------------------------------ run
    def set_alphas(self, alphas: Sequence[float]):
        """`interpolation_alphas` is mutable in the reference (alpha learning overwrites it,
        src/experiments/alpha_learning/common_utils.py:88): patch the sampler descriptors in place."""
        alphas = [float(a) for a in alphas]
        if len(alphas) != len(self.alphas):
            raise ValueError(f'{len(self.alphas)} interpolation alphas expected, got {len(alphas)}')
        if alphas == self.alphas:
            return
        self.alphas = alphas
        if getattr(self, 'alpha_dev', None) is not None:     # e4e defender: the alphas are device data, no descriptor changes
            self.alpha_dev.copy_(torch.tensor(alphas, dtype=torch.float32))
            return
        for d, idx in self._sampler_descs:
            d.alpha, d.one_minus_alpha = alphas[idx], 1.0 - alphas[idx]
        self.fwd.finalize()
        self.bwd.finalize()
        if getattr(self, '_g_fwd', None) is not None:
            self.enable_graphs()                            # descriptors changed: re-capture

    def stream(self) -> int:
        return torch.cuda.current_stream(self.device).cuda_stream

    def input_image_nchw(self) -> torch.Tensor:
        """the pre-processed image of the last forward (after blur / noise / clamp) as [rows, 3, H, W]"""
        t = self.acts['x0'].t
        if not self.image_s2d:
            return t[..., :3].permute(0, 3, 1, 2).contiguous()
        n, h2, w2, _ = t.shape
        v = t.view(n, h2, w2, 2, 2, IMG_LD)[..., :3]                     # [n, h/2, w/2, r_h, r_w, c]
        return v.permute(0, 5, 1, 3, 2, 4).reshape(n, 3, 2 * h2, 2 * w2).contiguous()

    # ---- HIP graphs: the ~700-launch plans become one graph launch each (host cost matters at EoT-32 row counts)
    def enable_graphs(self):
        """capture forward and backward on a side stream after one eager warm-up; invalidated by set_alphas/tuning"""
        if self.dry_run:
            raise RuntimeError('dry-run engine')
        self.disable_graphs()
        self._side = torch.cuda.Stream(device=self.device)
        cur = torch.cuda.current_stream(self.device)
        self.fwd.run(cur.cuda_stream)                       # eager warm-up (sets kernel attributes)
        if self.need_backward:
            self.bwd.run(cur.cuda_stream)
        self._side.wait_stream(cur)
        self._g_fwd = self.fwd.capture(self._side.cuda_stream)
        self._g_bwd = self.bwd.capture(self._side.cuda_stream) if self.need_backward else None
        cur.wait_stream(self._side)

    def disable_graphs(self):
        for h in (getattr(self, '_g_fwd', None), getattr(self, '_g_bwd', None)):
            if h is not None:
                L.Plan.destroy_graph(h)
        self._g_fwd = self._g_bwd = None

    def _launch_graph(self, handle):
        cur = torch.cuda.current_stream(self.device)
        self._side.wait_stream(cur)
        L.Plan.launch_graph(handle, self._side.cuda_stream)
        cur.wait_stream(self._side)

    def forward(self):
        if self.dry_run:
            raise RuntimeError('dry-run engine: plans were built for validation only')
        if getattr(self, '_g_fwd', None) is not None:
            self._launch_graph(self._g_fwd)
        else:
            self.fwd.run(self.stream())

    def backward(self, from_logits: bool = True, from_purified: bool = False, identity_purifier: bool = False):
        """Backward-to-input of the last forward.  Cotangents are read from `self.dlogits` (rows x classes) when
        from_logits and from `self.dpurified` (NCHW) when from_purified.  May be called repeatedly per forward.
        identity_purifier: BPDA — backward through the classifier only, the purifier (and the pre-processing in front of it)
        counted as the identity: dx[image] = sum over its EoT replicas of d loss / d purified."""
        if not self.need_backward:
            raise RuntimeError('engine was built without a backward plan')
        if self.dry_run:
            raise RuntimeError('dry-run engine: plans were built for validation only')
        if identity_purifier:
            if self.bpda is None:
                raise RuntimeError('BPDA needs a purifier whose output has the size of its input')
            if not from_logits or from_purified:
                raise ValueError('BPDA starts from the logits')
            self.bwd.run(self.stream(), start=0, end=self.bwd_split)
            self.bpda.run(self.stream())
            return
        if self.dpurified is not None and not from_purified:
            self.dpurified.zero_()
        if from_logits:
            if getattr(self, '_g_bwd', None) is not None:
                self._launch_graph(self._g_bwd)
            else:
                self.bwd.run(self.stream())
        else:
            if self._purified_grad_nhwc is None:
                raise RuntimeError('classifier-only engine: backward starts from the logits')
            self._purified_grad_nhwc.g.zero_()
            self.bwd.run(self.stream(), start=self.bwd_split)

