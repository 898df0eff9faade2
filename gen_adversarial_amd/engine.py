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

RES_SCALE = 0.1          # `0.1 * self.residual(x)` — architecture.py:133,183
IMG_LD = 8                  # channel pitch of the NHWC image tensors (3 channels + zero padding)
WS_FLOATS = 32 * 1024 * 1024     # split-K workspace shared by every conv of an engine (128 MB)
TUNE_FILE = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'conv_tune_gfx950.json')
_TUNE_CACHE: Optional[dict] = None


def tune_cache() -> dict:
    """(tile, splits) per conv shape, measured on an MI355X by Engine.autotune and kept in-tree."""
    global _TUNE_CACHE
    if _TUNE_CACHE is None:
        _TUNE_CACHE = {}
        if os.path.exists(TUNE_FILE):
            with open(TUNE_FILE) as f:
                _TUNE_CACHE = json.load(f)
    return _TUNE_CACHE


def conv_key(d) -> str:
    return ('b3_' if d.w_hi else '') + '_'.join(str(int(v)) for v in (
        d.N * d.Ho * d.Wo, d.Cout, d.C1, d.C2, d.KH, d.sn, d.sd, d.Hi, d.pro_act, bool(d.pro_scale), d.pro_per_row,
        bool(d.dact_x), d.dact_act, bool(d.addend), bool(d.addend2), d.addend_bcast_n)) + (
        f'_kw{d.KW}' if d.KW != d.KH else '')


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


class WeightStore:
    """Folded weights on the device, shared by every engine (row count) built for one model."""

    def __init__(self, device):
        self.device = torch.device(device)
        self.cache: Dict[str, dict] = {}
        self.splits: Dict[int, tuple] = {}
        self.bytes = 0

    def split(self, w: torch.Tensor):
        """bf16 (hi, lo) pair of a device weight tensor, hi = bf16(w), lo = bf16(w - hi); made once per tensor."""
        k = w.data_ptr()
        if k not in self.splits:
            hi = w.to(torch.bfloat16)
            lo = (w - hi.float()).to(torch.bfloat16)
            self.splits[k] = (hi.contiguous(), lo.contiguous(), w)
            self.bytes += 4 * w.numel()
        return self.splits[k][0], self.splits[k][1]

    def get(self, key: str, fn):
        if key not in self.cache:
            d = {k: v.to(self.device, dtype=torch.float32).contiguous() for k, v in fn().items()}
            self.bytes += sum(v.numel() * 4 for v in d.values())
            self.cache[key] = d
        return self.cache[key]


class Act:
    """An NHWC activation buffer plus its (lazily allocated) gradient buffer."""

    def __init__(self, eng: "Engine", n, h, w, c, name=''):
        self.eng, self.n, self.h, self.w, self.c, self.name = eng, n, h, w, c, name
        self.t = eng.alloc((n, h, w, c))
        self._g = None
        self.g_written = False
        eng.acts[name] = self

    @property
    def g(self) -> torch.Tensor:
        if self._g is None:
            self._g = self.eng.alloc((self.n, self.h, self.w, self.c))
        return self._g


class Engine:
    def __init__(self, nvae_sd, nvae_cfg: dict, resolution, vgg_sd, vgg_spec: VggSpec, rows: int, rep: int,
                 alphas: Sequence[float], temperature: float = 0.6, noise_eps: float = 0.0,
                 device: str = 'cuda:0', need_backward: bool = True, dry_run: bool = False,
                 store: Optional[WeightStore] = None, precision: str = 'bf16x3', blur: bool = False,
                 share_encoder: bool = False):
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
        self.bytes = 0
        self.acts = {}                       # name -> Act (debugging / tests)
        self.version = 0                     # bumped by callers after each forward (stale-backward detection)
        self._sampler_descs = []             # (desc, latent index): alphas can be changed without rebuilding
        self._keep = []                      # weights etc.
        self.fwd = L.Plan()
        self.bwd = L.Plan()
        self._bwd_steps = []                 # closures emitting backward ops, replayed in reverse
        self._build(nvae_sd, vgg_sd)

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
             ldx=None, ldy=None, ldadd=None, flags=0):
        """x, x2, y, addend*, dact_x are torch tensors [N,H,W,C] (or Act.t); shapes are taken from them."""
        d = L.ConvDesc()
        N, Hi, Wi, Cx = x.shape
        d.x, d.ldx = _ptr(x), (ldx or Cx)
        d.C1 = cin if cin is not None else Cx
        if x2 is not None:
            d.x2, d.ldx2, d.C2 = _ptr(x2), x2.shape[3], x2.shape[3]
        d.w, d.bias = _ptr(w), _ptr(bias)
        if self.precision == 'bf16x3' and not self.dry_run:
            hi, lo = self.store.split(w)
            d.w_hi, d.w_lo = _ptr(hi), _ptr(lo)
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
        if target.g_written:
            il.addend = _ptr(target.g)
        self.bwd.add(il, name + '.interleave')
        target.g_written = True

    def se_forward(self, name, t: Act, wts, P, res_scale=None):
        """squeeze + excite; returns (gate, hid) buffers."""
        n, c = t.n, t.c
        hd = wts['se_w1'].shape[0]
        hid = self.alloc((n, hd))
        gate = self.alloc((n, c))
        e = L.SeExciteDesc()
        e.t = _ptr(t.t)                                     # fused squeeze + excite (one workgroup per row)
        e.w1, e.b1, e.w2, e.b2 = _ptr(wts['se_w1']), _ptr(wts['se_b1']), _ptr(wts['se_w2']), _ptr(wts['se_b2'])
        e.hid, e.gate, e.N, e.C, e.Hd, e.P, e.res_scale, e.backward = (_ptr(hid), _ptr(gate), n, c, hd, P,
                                                                        RES_SCALE if res_scale is None else res_scale, 0)
        self.fwd.add(e, f'{name}.se_gate')
        return gate, hid

    def se_backward(self, name, dout: torch.Tensor, t: Act, wts, gate, hid, P, res_scale=None):
        """emits d(gate) reduction + excite backward; returns the per-row prologue (scale, shift) for the next GEMM."""
        n, c = t.n, t.c
        ps = self.scratch((n, c), f'ps{c}')
        pb = self.scratch((n, c), f'pb{c}')
        e = L.SeExciteDesc()
        e.t, e.dout = _ptr(t.t), _ptr(dout)                 # fused d(gate) reduction + excite backward
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

    # ------------------------------------------------------------------------------------------------ cells
    def enc_cell(self, cell: EncCellSpec, x: Act) -> Act:
        """ResidualCellEncoder (architecture.py:96-136): fwd ops now, bwd ops registered for later."""
        wts = self.devd(cell.prefix, lambda: F.fold_enc_cell(self.nvae_sd, cell))
        n, h, w = x.n, x.h, x.w
        st = 2 if cell.down else 1
        ho, wo = h // st, w // st
        t1 = Act(self, n, ho, wo, cell.cout, cell.prefix + '.t1')
        t2 = Act(self, n, ho, wo, cell.cout, cell.prefix + '.t2')
        out = Act(self, n, ho, wo, cell.cout, cell.prefix + '.out')
        p = cell.prefix
        self.conv(self.fwd, p + '.conv1', x.t, wts['w1'], t1.t, bias=wts['b1'], K=3, sn=st, pad=1,
                  pro_scale=wts['pro_scale'], pro_shift=wts['pro_shift'], pro_act=L.GA_ACT_SILU)
        self.conv(self.fwd, p + '.conv2', t1.t, wts['w2'], t2.t, bias=wts['b2'], K=3, pad=1, pro_act=L.GA_ACT_SILU)
        gate, hid = self.se_forward(p, t2, wts, ho * wo)
        if cell.down:
            sk = Act(self, n, ho, wo, cell.cout, p + '.skip')
            self.conv(self.fwd, p + '.skip', x.t, wts['ws'], sk.t, bias=wts['bs'], K=1, sn=2, pad=0, pro_act=L.GA_ACT_SILU)
            skip_t = sk.t
        else:
            skip_t = x.t
        a = L.SeApplyDesc()
        a.skip, a.t, a.gate, a.out = _ptr(skip_t), _ptr(t2.t), _ptr(gate), _ptr(out.t)
        a.N, a.H, a.W, a.C, a.skip_mode, a.res_scale = n, ho, wo, cell.cout, 0, RES_SCALE
        self.fwd.add(a, p + '.merge')

        def backward():
            ps, pb = self.se_backward(p, out.g, t2, wts, gate, hid, ho * wo)
            dt1 = self.scratch((n, ho, wo, cell.cout), 'enc_dt1')
            self.conv(self.bwd, p + '.conv2^T', out.g, wts['w2_bwd'], dt1, K=3, pad=1,
                      pro_scale=ps, pro_shift=pb, pro_per_row=1, dact_x=t1.t, dact_act=L.GA_ACT_SILU)
            if cell.down:       # both stride-2 transposes by sub-pixel decomposition (stride-1 convs on the matrix path)
                self.grad_conv_up2(p + '.conv1^T', dt1, wts, 'w1_sub', x, dact_x=x.t, dact_scale=wts['pro_scale'],
                                   dact_shift=wts['pro_shift'], dact_act=L.GA_ACT_SILU)
                self.grad_conv_up2(p + '.skip^T', out.g, wts, 'ws_sub', x, dact_x=x.t, dact_act=L.GA_ACT_SILU)
            else:
                self.grad_conv(p + '.conv1^T', dt1, wts['w1_bwd'], x, K=3, sn=1, sd=1, pad=1, primary=out.g,
                               dact_x=x.t, dact_scale=wts['pro_scale'], dact_shift=wts['pro_shift'], dact_act=L.GA_ACT_SILU)
        self._bwd_steps.append(backward)
        return out

    def dec_cell(self, cell: DecCellSpec, x: Act) -> Act:
        """ResidualCellDecoder (architecture.py:139-186) with nearest-up folded into the depthwise read and the
        SkipUp 1x1 applied before its bilinear interpolation."""
        wts = self.devd(cell.prefix, lambda: F.fold_dec_cell(self.nvae_sd, cell))
        n, h, w = x.n, x.h, x.w
        up = cell.up
        H, W = (2 * h, 2 * w) if up else (h, w)
        hid_c = cell.hidden
        p = cell.prefix
        t1 = Act(self, n, h, w, hid_c, p + '.t1')
        t2 = Act(self, n, H, W, hid_c, p + '.t2')
        t3 = Act(self, n, H, W, cell.cout, p + '.t3')
        out = Act(self, n, H, W, cell.cout, p + '.out')
        self.conv(self.fwd, p + '.pw1', x.t, wts['w1'], t1.t, bias=wts['b1'], K=1)
        d = L.DwDesc()
        d.x, d.w, d.bias, d.y = _ptr(t1.t), _ptr(wts['wd']), _ptr(wts['bd']), _ptr(t2.t)
        d.N, d.H, d.W, d.C, d.pro_act, d.up2 = n, H, W, hid_c, L.GA_ACT_SILU, int(up)
        self.fwd.add(d, p + '.dw5')
        self.conv(self.fwd, p + '.pw2', t2.t, wts['w2'], t3.t, bias=wts['b2'], K=1, pro_act=L.GA_ACT_SILU)
        gate, hid = self.se_forward(p, t3, wts, H * W)
        a = L.SeApplyDesc()
        if up:
            sl = Act(self, n, h, w, cell.cout, p + '.skip_low')
            self.conv(self.fwd, p + '.skip', x.t, wts['ws'], sl.t, bias=wts['bs'], K=1)
            a.skip, a.skip_mode = _ptr(sl.t), 1
        else:
            a.skip, a.skip_mode = _ptr(x.t), 0
        a.t, a.gate, a.out = _ptr(t3.t), _ptr(gate), _ptr(out.t)
        a.N, a.H, a.W, a.C, a.res_scale = n, H, W, cell.cout, RES_SCALE
        self.fwd.add(a, p + '.merge')

        def backward():
            ps, pb = self.se_backward(p, out.g, t3, wts, gate, hid, H * W)
            dt2 = self.scratch((n, H, W, hid_c), 'dec_dt2')
            self.conv(self.bwd, p + '.pw2^T', out.g, wts['w2_bwd'], dt2, K=1,
                      pro_scale=ps, pro_shift=pb, pro_per_row=1, dact_x=t2.t, dact_act=L.GA_ACT_SILU)
            dt1 = self.scratch((n, h, w, hid_c), 'dec_dt1')
            b = L.DwDesc()
            b.x, b.w, b.dact_x, b.y = _ptr(dt2), _ptr(wts['wd_bwd']), _ptr(t1.t), _ptr(dt1)
            b.N, b.H, b.W, b.C, b.dact_act, b.pool2 = n, H, W, hid_c, L.GA_ACT_SILU, int(up)
            self.bwd.add(b, p + '.dw5^T')
            self.grad_conv(p + '.pw1^T', dt1, wts['w1_bwd'], x, K=1, primary=None if up else out.g)
            if up:
                dsl = self.scratch((n, h, w, cell.cout), 'dec_dsl')
                bl = L.BilinearBwdDesc()
                bl.dhigh, bl.dlow, bl.N, bl.h, bl.w, bl.C, bl.accumulate = _ptr(out.g), _ptr(dsl), n, h, w, cell.cout, 0
                self.bwd.add(bl, p + '.bilinear^T')
                self.grad_conv(p + '.skip^T', dsl, wts['ws_bwd'], x, K=1)
        self._bwd_steps.append(backward)
        return out

    # ------------------------------------------------------------------------------------------------ build
    def _build(self, nvae_sd, vgg_sd):
        self.nvae_sd = nvae_sd
        self._scratch = {}
        R = self.rows
        H = self.resolution[1]

        # ---- boundary buffers (caller-visible)
        self.x_in = self.alloc((R // self.rep, 3, H, H))                    # NCHW images in [0,1]
        self.noise = self.alloc((R, 3, H, H)) if self.noise_eps != 0.0 else None
        self.noise_coef = self.alloc((R,)) if self.noise_eps != 0.0 else None
        self.dx = self.alloc((R // self.rep, 3, H, H))
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
            dx_dst = self.alloc((R // self.rep, 3, H, H))
            bl = L.BlurDesc()
            bl.x, bl.y, bl.taps, bl.planes, bl.H, bl.W, bl.k, bl.backward = _ptr(self.x_in), _ptr(x_src), _ptr(taps), (R // self.rep) * 3, H, H, k, 0
            self.fwd.add(bl, 'gauss_blur')

            def bwd_blur():
                b = L.BlurDesc()
                b.x, b.y, b.taps, b.planes, b.H, b.W, b.k, b.backward = _ptr(dx_dst), _ptr(self.dx), _ptr(taps), (R // self.rep) * 3, H, H, k, 1
                self.bwd.add(b, 'gauss_blur^T')
            self._bwd_steps.append(bwd_blur)

        R0 = self.enc_rows                       # rows entering the network (images when the encoder is shared)
        rep0 = 1 if self.share_encoder else self.rep
        # the NHWC image is kept at a pitch of IMG_LD = 8 channels (3 real + zero pad): the first convolutions then take
        # 16-B loads and the split-bf16 matrix path like every other layer instead of a scalar 3-channel gather
        # (a classifier-only ResNet engine takes the image in space-to-depth form: its 7x7/2 stem is then a 4x4/1 conv)
        self.image_s2d = (not self.has_nvae) and isinstance(self.vspec, ResNetSpec)
        x0 = Act(self, R0, H // 2, H // 2, 4 * IMG_LD, 'x0') if self.image_s2d else Act(self, R0, H, H, IMG_LD, 'x0')
        io = L.ImageIoDesc()
        io.x_nchw, io.noise_nchw, io.noise_coef, io.y_nhwc = _ptr(x_src), _ptr(self.noise), _ptr(self.noise_coef), _ptr(x0.t)
        io.N, io.C, io.H, io.W, io.rep, io.backward, io.ld, io.s2d = R0, 3, H, H, rep0, 0, IMG_LD, int(self.image_s2d)
        self.fwd.add(io, 'image_in')

        def bwd_image():
            b = L.ImageIoDesc()
            b.x_nchw, b.noise_nchw, b.noise_coef = _ptr(x_src), _ptr(self.noise), _ptr(self.noise_coef)
            b.dy_nhwc, b.dx_nchw = _ptr(x0.g), _ptr(dx_dst)
            b.N, b.C, b.H, b.W, b.rep, b.backward, b.ld, b.s2d = R0, 3, H, H, rep0, 1, IMG_LD, int(self.image_s2d)
            self.bwd.add(b, 'image_in^T')
        self._bwd_steps.append(bwd_image)

        img = self._build_nvae(x0) if self.has_nvae else x0

        # ---- classifier
        n_nvae_steps = len(self._bwd_steps)
        build = (self._build_resnet if isinstance(self.vspec, ResNetSpec) else
                 self._build_e4e if isinstance(self.vspec, E4ESpec) else self._build_vgg)
        self.logits = build(vgg_sd, img)

        # ---- emit the backward plan: reverse registration order (classifier part first)
        self.bwd_split = 0
        if self.need_backward:
            for step in reversed(self._bwd_steps[n_nvae_steps:]):
                step()
            self.bwd_split = len(self.bwd)
            for step in reversed(self._bwd_steps[:n_nvae_steps]):
                step()
        self._bwd_steps = None
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
            d.tile, d.splits = int(tile), int(splits)
            d.ws, d.ws_floats = (_ptr(self.ws), WS_FLOATS) if splits > 1 else (None, 0)
            if not use_bf3:                      # this shape is faster on the exact fp32 kernel (small K or Cout)
                d.w_hi, d.w_lo = None, None
        self.fwd.finalize()
        self.bwd.finalize()

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
            halo = (5, 6) if (d.w_hi and d.KH == 3 and d.KW == 3 and d.sn == 1 and d.sd == 1 and d.C2 == 0) else ()
            for use_bf3, tile in [(m_, t_) for m_ in modes for t_ in (1, 2, 3, 4) + (halo if m_ else ())]:
                bm, bn = {1: (128, 128), 2: (128, 64), 3: (64, 64), 4: (128, 32), 5: (128, 128), 6: (128, 64)}[tile]
                if bn >= 2 * max(32, d.Cout) and tile != 4:
                    continue
                blocks = -(-M // bm) * -(-d.Cout // bn)
                for splits in (1, 2, 4, 8, 16, 32):
                    if splits > 1 and (blocks * splits > 2048 or T < 2 * splits or splits * M * d.Cout > WS_FLOATS):
                        continue
                    if tile >= 5 and splits > d.C1 // 32:          # the halo kernel splits K over 32-channel chunks
                        continue
                    t = L.ConvDesc.from_buffer_copy(d)
                    t.tile, t.splits = tile, splits
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

    def _build_nvae(self, x0: Act) -> Act:
        """NVAEDefenseModel.purify (models.py:160-274) on the NHWC image x0; returns the purified NHWC image."""
        nvae_sd = self.nvae_sd
        spec, R = self.spec, self.rows
        H = spec.resolution
        NL = spec.num_latent
        self.eps = [self.alloc((R, NL, gs.res, gs.res)) for gs in spec.groups]   # NCHW like the reference draws them
        self.purified = self.alloc((R, 3, H, H))                            # NCHW

        # ---- stem: normalisation (x-0.5)/0.5 as prologue affine, then weight-normed 3x3 (model.py:106-107)
        stem = self.devd('stem', lambda: F.pad_image_conv(F.fold_wn_conv(nvae_sd, 'preprocessing_block.init_conv'), 3, IMG_LD))
        norm = self.devd('norm05', lambda: {'two': torch.full((IMG_LD,), 2.0), 'mone': torch.full((IMG_LD,), -1.0)})
        two, mone = norm['two'], norm['mone']
        RE = self.enc_rows                        # encoder rows: R, or R/rep when the encoder is shared by the replicas
        erep = self.rep if self.share_encoder else 1
        x = Act(self, RE, H, H, spec.base_channels, 'stem')
        self.conv(self.fwd, 'stem', x0.t, stem['w'], x.t, bias=stem['b'], K=3, pad=1, pro_scale=two, pro_shift=mone)
        stem_out = x

        def bwd_stem():
            self.grad_conv('stem^T', stem_out.g, stem['w_bwd'], x0, K=3, pad=1,
                           dact_x=x0.t, dact_scale=two, dact_shift=mone, dact_act=L.GA_ACT_NONE)
        self._bwd_steps.append(bwd_stem)

        for cell in spec.pre_cells:
            x = self.enc_cell(cell, x)

        stash: Dict[str, Act] = {}
        for kind, payload in spec.enc_program:
            if kind == 'stash':
                stash[payload] = x
            else:
                x = self.enc_cell(payload, x)
        x_top = x

        # ---- encoder_0: ELU -> 1x1 -> ELU (model.py:184-187) and sampler_0:0 (3x3, mu half only: purify uses
        #      dist_enc.mu alone, models.py:199-206)
        C0 = spec.enc0_channels
        g0 = spec.groups[0]
        enc0 = self.devd('encoder_0', lambda: F.fold_wn_conv(nvae_sd, 'encoder_0.1'))
        e0 = Act(self, RE, g0.res, g0.res, C0, 'enc0')
        self.conv(self.fwd, 'encoder_0', x_top.t, enc0['w'], e0.t, bias=enc0['b'], K=1, pro_act=L.GA_ACT_ELU)
        s00 = self.devd('enc_sampler_0:0', lambda: F.fold_wn_conv(nvae_sd, 'enc_sampler.sampler_0:0', out_slice=slice(0, NL)))
        muq0 = Act(self, RE, g0.res, g0.res, NL, 'mu_q0')
        self.conv(self.fwd, 'enc_sampler_0:0', e0.t, s00['w'], muq0.t, bias=s00['b'], K=3, pad=1, pro_act=L.GA_ACT_ELU)
        z = Act(self, R, g0.res, g0.res, NL, 'z0')
        self._sampler_fwd('sample_0:0', muq0, None, self.eps[0], z, self.alphas[0], q_rep=erep)

        # ---- combiner_0:0 on cat[const_prior, z0]: the prior half is row-independent -> folded into a broadcast addend
        def fold_comb0():
            wfull = F.wn_weight64(nvae_sd, 'decoder_combiners.combiner_0:0.conv')[:, :, 0, 0]   # [C0, C0+NL]
            prior = nvae_sd['const_prior'].double()[0]                                           # [C0,h,w]
            pc = torch.einsum('oc,chw->hwo', wfull[:, :C0], prior) + \
                nvae_sd['decoder_combiners.combiner_0:0.conv.bias'].double()
            if spec.num_nf_cells:            # flow of this group = z - c (folding.nf_constant_shift): fold W_z c into the addend
                pc = pc - wfull[:, C0:] @ F.nf_constant_shift(nvae_sd, '0:0', spec.num_nf_cells, NL)
            return {'pc': pc.float().unsqueeze(0), 'wz': wfull[:, C0:].float(), 'wz_bwd': wfull[:, C0:].t().float()}
        c0w = self.devd('combiner_0:0', fold_comb0)
        pc, wz, wz_bwd = c0w['pc'], c0w['wz'], c0w['wz_bwd']                                     # pc: [1,h,w,C0]
        x = Act(self, R, g0.res, g0.res, C0, 'comb_0:0')
        self.conv(self.fwd, 'combiner_0:0', z.t, wz, x.t, K=1, addend=pc, addend_bcast=True)
        comb0_out, z0 = x, z

        def bwd_group0():
            self.grad_conv('combiner_0:0^T', comb0_out.g, wz_bwd, z0, K=1)
            self._sampler_bwd('sample_0:0^T', muq0, None, self.eps[0], z0, self.alphas[0], None, q_rep=erep)
            self.grad_conv('enc_sampler_0:0^T', muq0.g, s00['w_bwd'], e0, K=3, pad=1, dact_x=e0.t, dact_act=L.GA_ACT_ELU)
            self.grad_conv('encoder_0^T', e0.g, enc0['w_bwd'], x_top, K=1, dact_x=x_top.t, dact_act=L.GA_ACT_ELU)
        group0_bwd = bwd_group0     # must run after every decoder-side use of the encoder features: registered below

        dec_bwd_steps_start = len(self._bwd_steps)
        # decoder-side backward steps are registered AFTER encoder ones so that they replay first (reverse order)
        self._bwd_steps.append(group0_bwd)

        for gs in spec.groups:
            if gs.dec_cells:
                for cell in gs.dec_cells:
                    x = self.dec_cell(cell, x)
                x = self._latent_group(gs, x, stash[f'{gs.s}:{gs.g}'], erep)
            if gs.g == spec.groups_per_scale[gs.s] - 1 and gs.s in spec.dec_up_cells:
                x = self.dec_cell(spec.dec_up_cells[gs.s], x)

        for cell in spec.post_cells:
            x = self.dec_cell(cell, x)

        # ---- to_logits (ELU -> 3x3, model.py:310-313) + DiscMixLogistic.mean + denormalise
        tl = self.devd('to_logits', lambda: F.fold_wn_conv(nvae_sd, 'to_logits.1'))
        logits = Act(self, R, H, H, spec.logits_out, 'mix_logits')
        post_out = x
        self.conv(self.fwd, 'to_logits', x.t, tl['w'], logits.t, bias=tl['b'], K=3, pad=1, pro_act=L.GA_ACT_ELU)
        img = Act(self, R, H, H, IMG_LD, 'purified_nhwc')
        dm = L.DmlDesc()
        dm.logits, dm.ld, dm.nmix, dm.img_nchw, dm.img_nhwc = _ptr(logits.t), spec.logits_out, spec.num_mixtures, _ptr(self.purified), _ptr(img.t)
        dm.N, dm.H, dm.W, dm.backward, dm.ld_img = R, H, H, 0, IMG_LD
        self.fwd.add(dm, 'dml_mean')
        self.dpurified = self.alloc((R, 3, H, H))    # optional external gradient on the purified image (NCHW)
        purified_img = img

        def bwd_dml():
            b = L.DmlDesc()
            b.logits, b.ld, b.nmix, b.dimg_nhwc, b.dlogits = _ptr(logits.t), spec.logits_out, spec.num_mixtures, _ptr(img.g), _ptr(logits.g)
            b.dimg_nchw = _ptr(self.dpurified)
            b.N, b.H, b.W, b.backward, b.ld_img = R, H, H, 1, IMG_LD
            self.bwd.add(b, 'dml_mean^T')
            self.grad_conv('to_logits^T', logits.g, tl['w_bwd'], post_out, K=3, pad=1, dact_x=post_out.t, dact_act=L.GA_ACT_ELU)
        self._bwd_steps.append(bwd_dml)

        self._purified_grad_nhwc = purified_img
        return img

    # ------------------------------------------------------------------------------------------------ latents
    def _sampler_fwd(self, name, muq: Act, p: Optional[Act], eps, z: Act, alpha: float, q_rep: int = 1):
        d = L.SamplerDesc()
        d.mu_q, d.ldq = _ptr(muq.t), muq.c
        if p is not None:
            d.p, d.ldp = _ptr(p.t), p.c
        d.eps, d.eps_nchw, d.z = _ptr(eps), 1, _ptr(z.t)
        d.N, d.h, d.w, d.NL = z.n, z.h, z.w, z.c
        d.alpha, d.one_minus_alpha, d.temp, d.backward = alpha, 1.0 - alpha, self.temperature, 0
        d.q_rep = q_rep
        self._sampler_descs.append((d, [i for i, e in enumerate(self.eps) if e is eps][0]))
        self.fwd.add(d, name)

    def _sampler_bwd(self, name, muq: Act, p: Optional[Act], eps, z: Act, alpha: float, dp: Optional[Act], q_rep: int = 1):
        d = L.SamplerDesc()
        d.mu_q, d.ldq = _ptr(muq.t), muq.c
        if p is not None:
            d.p, d.ldp, d.dp = _ptr(p.t), p.c, _ptr(p.g)
            p.g_written = True
        d.eps, d.eps_nchw, d.dz = _ptr(eps), 1, _ptr(z.g)
        d.q_rep = q_rep
        rows_grad = None
        if q_rep > 1:
            rows_grad = self.scratch((z.n, z.h, z.w, z.c), 'dmu_q_rows')
            d.dmu_q_rows = _ptr(rows_grad)
        else:
            d.dmu_q = _ptr(muq.g)
        d.N, d.h, d.w, d.NL = z.n, z.h, z.w, z.c
        d.alpha, d.one_minus_alpha, d.temp, d.backward = alpha, 1.0 - alpha, self.temperature, 1
        self._sampler_descs.append((d, [i for i, e in enumerate(self.eps) if e is eps][0]))
        self.bwd.add(d, name)
        if q_rep > 1:
            self.rep_sum(name + '.rep_sum', rows_grad, muq, q_rep)
        muq.g_written = True

    def rep_sum(self, name, x_rows: torch.Tensor, target: Act, rep: int):
        """target.g (+)= sum over the `rep` replicas of x_rows (gradient of a tensor shared by the EoT replicas)"""
        r = L.RepSumDesc()
        r.x, r.y, r.rows, r.inner, r.rep = _ptr(x_rows), _ptr(target.g), x_rows.shape[0], x_rows[0].numel(), rep
        r.accumulate = int(target.g_written)
        self.bwd.add(r, name)
        target.g_written = True

    def _latent_group(self, gs, x: Act, enc_feat: Act, enc_rep: int = 1) -> Act:
        """models.py:236-257 for one latent group: encoder/decoder parameters, interpolation, combiner."""
        sd, R, NL, C, r = self.nvae_sd, self.rows, self.spec.num_latent, gs.channels, gs.res
        key = f'{gs.s}:{gs.g}'
        ec_w = self.devd(f'enc_combiner_{key}', lambda: F.fold_wn_conv(sd, f'encoder_combiners.combiner_{key}.conv'))
        es_w = self.devd(f'enc_sampler_{key}', lambda: F.fold_wn_conv(sd, f'enc_sampler.sampler_{key}', out_slice=slice(0, NL)))
        ds_w = self.devd(f'dec_sampler_{key}', lambda: F.fold_wn_conv(sd, f'dec_sampler.sampler_{key}.1'))

        def fold_comb():
            cb = F.wn_weight64(sd, f'decoder_combiners.combiner_{key}.conv')[:, :, 0, 0]         # [C, C+NL]
            bias = sd[f'decoder_combiners.combiner_{key}.conv.bias'].double()
            if self.spec.num_nf_cells:       # the group's flow is z - c: combiner(cat[x, z - c]) = ... - W_z c
                bias = bias - cb[:, C:] @ F.nf_constant_shift(sd, key, self.spec.num_nf_cells, NL)
            return {'w': cb.float(), 'b': bias.float(),
                    'x_bwd': cb[:, :C].t().float(), 'z_bwd': cb[:, C:].t().float()}
        cbw = self.devd(f'combiner_{key}', fold_comb)
        cb_w, cb_b, cbx_bwd, cbz_bwd = cbw['w'], cbw['b'], cbw['x_bwd'], cbw['z_bwd']
        alpha = self.alphas[gs.latent_idx]
        eps = self.eps[gs.latent_idx]

        ec = Act(self, R, r, r, C, f'ec_{key}')
        d_ec = self.conv(self.fwd, f'enc_combiner_{key}', x.t, ec_w['w'], ec.t, bias=ec_w['b'], K=1, addend=enc_feat.t)
        d_ec.addend_rep = enc_rep
        muq = Act(self, R, r, r, NL, f'mu_q_{key}')
        self.conv(self.fwd, f'enc_sampler_{key}', ec.t, es_w['w'], muq.t, bias=es_w['b'], K=3, pad=1)
        pp = Act(self, R, r, r, 2 * NL, f'p_{key}')
        self.conv(self.fwd, f'dec_sampler_{key}', x.t, ds_w['w'], pp.t, bias=ds_w['b'], K=1, pro_act=L.GA_ACT_ELU)
        z = Act(self, R, r, r, NL, f'z_{key}')
        self._sampler_fwd(f'sample_{key}', muq, pp, eps, z, alpha)
        out = Act(self, R, r, r, C, f'comb_{key}')
        self.conv(self.fwd, f'combiner_{key}', x.t, cb_w, out.t, bias=cb_b, K=1, x2=z.t)

        def backward():
            self.grad_conv(f'combiner_{key}^T.z', out.g, cbz_bwd, z, K=1)
            self.grad_conv(f'combiner_{key}^T.x', out.g, cbx_bwd, x, K=1)
            self._sampler_bwd(f'sample_{key}^T', muq, pp, eps, z, alpha, pp)
            self.grad_conv(f'dec_sampler_{key}^T', pp.g, ds_w['w_bwd'], x, K=1, dact_x=x.t, dact_act=L.GA_ACT_ELU)
            self.grad_conv(f'enc_sampler_{key}^T', muq.g, es_w['w_bwd'], ec, K=3, pad=1)
            self.grad_conv(f'enc_combiner_{key}^T', ec.g, ec_w['w_bwd'], x, K=1)
            # the additive encoder feature receives d(ec) unchanged (summed over the replicas that share it)
            if enc_rep > 1:
                self.rep_sum(f'enc_feat_{key}.grad', ec.g, enc_feat, enc_rep)
                return
            if enc_feat.g_written:
                a = L.AxpbyDesc()
                a.x, a.y, a.n, a.alpha, a.beta = _ptr(ec.g), _ptr(enc_feat.g), ec.g.numel(), 1.0, 1.0
            else:
                a = L.AxpbyDesc()
                a.x, a.y, a.n, a.alpha, a.beta = _ptr(ec.g), _ptr(enc_feat.g), ec.g.numel(), 1.0, 0.0
                enc_feat.g_written = True
            self.bwd.add(a, f'enc_feat_{key}.grad')
        self._bwd_steps.append(backward)
        return out

    # ------------------------------------------------------------------------------------------------ classifier
    def _build_vgg(self, vsd, img: Act) -> torch.Tensor:
        """Vgg.forward on the purified image (abstract_models.py:188 -> :53-62): normalise (0.5,0.5) as prologue affine,
        conv+BN folded, ReLU as the next op's prologue, max-pool on pre-activations."""
        vs, R = self.vspec, self.rows
        norm = self.devd('norm05', lambda: {'two': torch.full((IMG_LD,), 2.0), 'mone': torch.full((IMG_LD,), -1.0)})
        two, mone = norm['two'], norm['mone']
        cur, first = img, True
        pending_pool = None
        for op in vs.program:
            if op[0] == 'conv':
                _, i, cin, cout = op
                wts = self.devd(f'vgg.conv{i}', lambda i=i, first=first: F.pad_image_conv(F.fold_vgg_conv(vsd, i), 3, IMG_LD)
                                if first else F.fold_vgg_conv(vsd, i))
                t = Act(self, R, cur.h, cur.w, cout, f'vgg.conv{i}')
                src = cur
                if first:
                    self.conv(self.fwd, f'vgg.conv{i}', src.t, wts['w'], t.t, bias=wts['b'], K=3, pad=1, pro_scale=two, pro_shift=mone)

                    def bwd(src=src, t=t, wts=wts, i=i):
                        self.grad_conv(f'vgg.conv{i}^T', t.g, wts['w_bwd'], src, K=3, pad=1,
                                       dact_x=src.t, dact_scale=two, dact_shift=mone, dact_act=L.GA_ACT_NONE)
                else:
                    self.conv(self.fwd, f'vgg.conv{i}', src.t, wts['w'], t.t, bias=wts['b'], K=3, pad=1, pro_act=L.GA_ACT_RELU)

                    def bwd(src=src, t=t, wts=wts, i=i):
                        self.grad_conv(f'vgg.conv{i}^T', t.g, wts['w_bwd'], src, K=3, pad=1, dact_x=src.t, dact_act=L.GA_ACT_RELU)
                self._bwd_steps.append(bwd)
                cur, first = t, False
            else:
                src = cur
                pl = Act(self, R, src.h // 2, src.w // 2, src.c, src.name + '.pool')
                m = L.MaxpoolDesc()
                m.x, m.y, m.N, m.H, m.W, m.C, m.backward = _ptr(src.t), _ptr(pl.t), R, src.h, src.w, src.c, 0
                self.fwd.add(m, pl.name)

                def bwd(src=src, pl=pl):
                    b = L.MaxpoolDesc()
                    b.x, b.dy, b.dx, b.N, b.H, b.W, b.C, b.backward = _ptr(src.t), _ptr(pl.g), _ptr(src.g), R, src.h, src.w, src.c, 1
                    self.bwd.add(b, pl.name + '^T')
                    src.g_written = True
                self._bwd_steps.append(bwd)
                cur = pl
        # head
        f = cur.h
        head = self.devd(f'vgg.head.f{f}', lambda: F.fold_vgg_head(vsd, vs.feat_channels, f))
        d = vs.head_dim
        feat = cur
        feat_flat = feat.t.view(R, 1, 1, f * f * feat.c)
        h1 = Act(self, R, 1, 1, d, 'vgg.head1')
        self.conv(self.fwd, 'vgg.head1', feat_flat, head['w_head'], h1.t, bias=head['b_head'], K=1, pro_act=L.GA_ACT_RELU)
        out = Act(self, R, 1, 1, vs.n_classes, 'vgg.logits')
        self.conv(self.fwd, 'vgg.head2', h1.t, head['w_out'], out.t, bias=head['b_out'], K=1, pro_act=L.GA_ACT_RELU)
        self.dlogits = out.g
        out.g_written = True

        def bwd_head():
            self.grad_conv('vgg.head2^T', out.g, head['w_out_bwd'], h1, K=1, dact_x=h1.t, dact_act=L.GA_ACT_RELU)
            gflat = feat.g.view(R, 1, 1, f * f * feat.c)
            assert not feat.g_written
            self.conv(self.bwd, 'vgg.head1^T', h1.g, head['w_head_bwd'], gflat, K=1, dact_x=feat_flat, dact_act=L.GA_ACT_RELU)
            feat.g_written = True
        self._bwd_steps.append(bwd_head)
        return out.t.view(R, vs.n_classes)

    # ------------------------------------------------------------------------------------------------ ResNet-50
    def _build_resnet(self, rsd, img: Act) -> torch.Tensor:
        """ResNet.forward (src/classifier/model.py:10-28; torchvision resnet50, resnet_spec.py) on the NHWC image:
        normalisation as the stem's prologue affine, every conv with its BatchNorm folded, residual sums stored
        PRE-activation (ReLU is the consumers' prologue, the identity branch adds relu(sum) through
        GA_CONV_ADDEND_RELU, and its cotangent passes the same relu' as the conv branch, GA_CONV_ADDEND_PRE_DACT)."""
        rs, R = self.vspec, self.rows
        if not self.image_s2d:
            raise NotImplementedError('ResNet behind a purifier: the purified image must be produced in space-to-depth form (next row)')
        norm = self.devd('norm05_s2d', lambda: {'two': torch.full((4 * IMG_LD,), 2.0), 'mone': torch.full((4 * IMG_LD,), -1.0)})
        two, mone = norm['two'], norm['mone']
        stem = self.devd('resnet.stem', lambda: F.fold_resnet_stem(rsd, IMG_LD))
        # 7x7/2 pad 3 == 4x4/1 over the space-to-depth image, window anchored two phase-pixels before the output pixel
        c1 = Act(self, R, img.h, img.w, rs.stem_channels, 'resnet.conv1')
        self.conv(self.fwd, 'resnet.conv1', img.t, stem['w'], c1.t, bias=stem['b'], K=4, pad=2, explicit_out=True,
                  pro_scale=two, pro_shift=mone)
        p1 = Act(self, R, c1.h // 2, c1.w // 2, rs.stem_channels, 'resnet.pool')
        m = L.Maxpool3s2Desc()
        m.x, m.y, m.N, m.H, m.W, m.C, m.backward = _ptr(c1.t), _ptr(p1.t), R, c1.h, c1.w, c1.c, 0
        self.fwd.add(m, 'resnet.maxpool')

        def bwd_stem():
            b = L.Maxpool3s2Desc()
            b.x, b.dy, b.dx, b.N, b.H, b.W, b.C, b.backward = _ptr(c1.t), _ptr(p1.g), _ptr(c1.g), R, c1.h, c1.w, c1.c, 1
            self.bwd.add(b, 'resnet.maxpool^T')
            c1.g_written = True
            self.grad_conv('resnet.conv1^T', c1.g, stem['w_bwd'], img, K=4, pad=1, explicit_out=True,
                           dact_x=img.t, dact_scale=two, dact_shift=mone, dact_act=L.GA_ACT_NONE)
        self._bwd_steps.append(bwd_stem)

        cur = p1
        for blk in rs.blocks:
            cur = self._resnet_block(rsd, blk, cur)

        head = self.devd('resnet.head', lambda: F.fold_resnet_head(rsd))
        last = cur
        pooled = Act(self, R, 1, 1, last.c, 'resnet.avgpool')
        a = L.AvgpoolActDesc()
        a.x, a.y, a.N, a.P, a.C, a.act, a.backward = _ptr(last.t), _ptr(pooled.t), R, last.h * last.w, last.c, L.GA_ACT_RELU, 0
        self.fwd.add(a, 'resnet.avgpool')
        h1 = Act(self, R, 1, 1, last.c, 'resnet.fc0')
        self.conv(self.fwd, 'resnet.fc0', pooled.t, head['w_h'], h1.t, bias=head['b_h'], K=1)
        out = Act(self, R, 1, 1, rs.n_classes, 'resnet.logits')
        self.conv(self.fwd, 'resnet.fc3', h1.t, head['w_o'], out.t, bias=head['b_o'], K=1, pro_act=L.GA_ACT_RELU)
        self.dlogits = out.g
        out.g_written = True

        def bwd_head():
            self.grad_conv('resnet.fc3^T', out.g, head['w_o_bwd'], h1, K=1, dact_x=h1.t, dact_act=L.GA_ACT_RELU)
            self.grad_conv('resnet.fc0^T', h1.g, head['w_h_bwd'], pooled, K=1)
            b = L.AvgpoolActDesc()
            b.x, b.dy, b.dx, b.N, b.P, b.C, b.act, b.backward = (_ptr(last.t), _ptr(pooled.g), _ptr(last.g), R, last.h * last.w,
                                                                 last.c, L.GA_ACT_RELU, 1)
            assert not last.g_written
            self.bwd.add(b, 'resnet.avgpool^T')
            last.g_written = True
        self._bwd_steps.append(bwd_head)
        return out.t.view(R, rs.n_classes)

    def _resnet_block(self, rsd, blk, s_in: Act) -> Act:
        """torchvision Bottleneck (1x1 -> 3x3 (stride) -> 1x1, + identity or 1x1-strided shortcut, ReLU after the sum)"""
        p, R = blk.prefix.replace('model.', 'resnet.'), self.rows
        wts = self.devd(p, lambda: F.fold_resnet_block(rsd, blk))
        h, w, st = s_in.h, s_in.w, blk.stride
        t1 = Act(self, R, h, w, blk.width, p + '.t1')
        t2 = Act(self, R, h // st, w // st, blk.width, p + '.t2')
        s_out = Act(self, R, h // st, w // st, blk.cout, p + '.sum')
        self.conv(self.fwd, p + '.conv1', s_in.t, wts['w1'], t1.t, bias=wts['b1'], K=1, pro_act=L.GA_ACT_RELU)
        cg = blk.width // blk.groups if blk.groups > 1 else 0                      # ResNeXt: grouped 3x3
        if cg:
            self.gconv(self.fwd, p + '.conv2', t1.t, wts['w2'], t2.t, cg, stride=st, pad=1, bias=wts['b2'], pro_act=L.GA_ACT_RELU)
        else:
            self.conv(self.fwd, p + '.conv2', t1.t, wts['w2'], t2.t, bias=wts['b2'], K=3, sn=st, pad=1, pro_act=L.GA_ACT_RELU)
        if blk.downsample:
            ds = Act(self, R, h // st, w // st, blk.cout, p + '.shortcut')
            self.conv(self.fwd, p + '.downsample', s_in.t, wts['wd'], ds.t, bias=wts['bd'], K=1, sn=st, pad=0, pro_act=L.GA_ACT_RELU)
            self.conv(self.fwd, p + '.conv3', t2.t, wts['w3'], s_out.t, bias=wts['b3'], K=1, pro_act=L.GA_ACT_RELU, addend=ds.t)
        else:
            d = self.conv(self.fwd, p + '.conv3', t2.t, wts['w3'], s_out.t, bias=wts['b3'], K=1, pro_act=L.GA_ACT_RELU, addend=s_in.t)
            d.flags = L.GA_CONV_ADDEND_RELU

        def backward():
            self.grad_conv(p + '.conv3^T', s_out.g, wts['w3_bwd'], t2, K=1, dact_x=t2.t, dact_act=L.GA_ACT_RELU)
            if st == 1 and cg:
                assert not t1.g_written
                self.gconv(self.bwd, p + '.conv2^T', t2.g, wts['w2_bwd'], t1.g, cg, stride=1, pad=1, dact_x=t1.t, dact_act=L.GA_ACT_RELU)
                t1.g_written = True
            elif st == 1:
                self.grad_conv(p + '.conv2^T', t2.g, wts['w2_bwd'], t1, K=3, pad=1, dact_x=t1.t, dact_act=L.GA_ACT_RELU)
            else:
                self.grad_conv_up2(p + '.conv2^T', t2.g, wts, 'w2_sub', t1, dact_x=t1.t, dact_act=L.GA_ACT_RELU, cg=cg)
            if blk.downsample:
                self.grad_conv(p + '.conv1^T', t1.g, wts['w1_bwd'], s_in, K=1, dact_x=s_in.t, dact_act=L.GA_ACT_RELU)
                if st == 1:
                    self.grad_conv(p + '.downsample^T', s_out.g, wts['wd_bwd'], s_in, K=1, dact_x=s_in.t, dact_act=L.GA_ACT_RELU)
                else:
                    self.grad_conv_up2(p + '.downsample^T', s_out.g, wts, 'wd_sub', s_in, dact_x=s_in.t, dact_act=L.GA_ACT_RELU)
            else:       # identity shortcut: (W1^T dt1 + d s_out) * relu'(s_in)
                assert not s_in.g_written
                d = self.grad_conv(p + '.conv1^T', t1.g, wts['w1_bwd'], s_in, K=1, primary=s_out.g,
                                   dact_x=s_in.t, dact_act=L.GA_ACT_RELU)
                d.flags = L.GA_CONV_ADDEND_PRE_DACT
        self._bwd_steps.append(backward)
        return s_out

    # ------------------------------------------------------------------------------------------------ e4e encoder
    def _build_e4e(self, esd, img: Act) -> torch.Tensor:
        """Encoder4Editing.forward (encoding/encoder.py:108-140; e4e_spec.py) on the NHWC image (taken as is: the caller's
        normalisation, if any, is part of its input).  Returns the latents as [rows, style_count * 512]; `dlogits` is
        their cotangent."""
        es, R = self.vspec, self.rows
        if self.image_s2d:
            raise NotImplementedError
        inp = self.devd('e4e.input', lambda: F.fold_e4e_input(esd, IMG_LD))
        t0 = Act(self, R, img.h, img.w, es.base, 'e4e.input.conv')
        self.conv(self.fwd, 'e4e.input.conv', img.t, inp['w'], t0.t, bias=inp['b'], K=3, pad=1)
        x = Act(self, R, img.h, img.w, es.base, 'e4e.input')
        pr = L.PreluDesc()
        pr.x, pr.slope, pr.y, pr.rows, pr.C, pr.backward = _ptr(t0.t), _ptr(inp['slope']), _ptr(x.t), R * img.h * img.w, es.base, 0
        self.fwd.add(pr, 'e4e.input.prelu')
        x_in = x

        def bwd_input():
            b = L.PreluDesc()
            b.x, b.slope, b.dy, b.dx, b.rows, b.C, b.backward = (_ptr(t0.t), _ptr(inp['slope']), _ptr(x_in.g), _ptr(t0.g),
                                                                 R * img.h * img.w, es.base, 1)
            self.bwd.add(b, 'e4e.input.prelu^T')
            t0.g_written = True
            self.grad_conv('e4e.input.conv^T', t0.g, inp['w_bwd'], img, K=3, pad=1)
        self._bwd_steps.append(bwd_input)

        feats = {}
        for i, u in enumerate(es.units):
            x = self._ir_se_unit(esd, u, x)
            if i in es.taps:
                feats[es.taps.index(i)] = x
        c1, c2, c3 = feats[0], feats[1], feats[2]
        p2 = self._fpn_level(esd, 'latlayer1', c3, c2)
        p1 = self._fpn_level(esd, 'latlayer2', p2, c1)
        src = (c3, p2, p1)

        D, cnt = es.style_dim, es.style_count
        out = self.alloc((R, cnt * D))
        dout = self.alloc((R, cnt * D))
        self.dlogits = dout
        g0 = self.alloc((R, 1, 1, D))                  # cotangent of w0 = sum over the heads (w[:, j] = w0 + delta_j)

        def bwd_w0():                                  # emitted first in the backward plan (registered last)
            r = L.ReduceDesc()
            r.a, r.out, r.N, r.P, r.C, r.scale = _ptr(dout), _ptr(g0), R, cnt, D, 1.0
            self.bwd.add(r, 'e4e.w0.grad')
        for j in range(cnt):
            self._style_head(esd, j, src[es.style_src[j]], out, dout, g0)
        self._bwd_steps.append(bwd_w0)
        return out

    def _ir_se_unit(self, esd, u, x: Act) -> Act:
        """bottleneck_IR_SE (encoding/helpers.py:97-119): shortcut(x) + SE(BN(conv3x3_s(PReLU(conv3x3(BN(x))))))"""
        p, R = 'e4e.' + u.prefix, self.rows
        wts = self.devd(p, lambda: F.fold_ir_se_unit(esd, u))
        h, w, st = x.h, x.w, u.stride
        ho, wo = h // st, w // st
        t1 = Act(self, R, h, w, u.depth, p + '.t1')
        t2 = Act(self, R, ho, wo, u.depth, p + '.t2')
        out = Act(self, R, ho, wo, u.depth, p + '.out')
        self.conv(self.fwd, p + '.conv1', x.t, wts['w1'], t1.t, K=3, pad=1, pro_scale=wts['pro_scale'], pro_shift=wts['pro_shift'])
        self.conv(self.fwd, p + '.conv2', t1.t, wts['w2'], t2.t, bias=wts['b2'], K=3, sn=st, pad=1,
                  pro_scale=wts['slope'], pro_shift=wts['slope'], flags=L.GA_CONV_PRO_PRELU)
        gate, hid = self.se_forward(p, t2, wts, ho * wo, res_scale=1.0)
        conv_shortcut = u.cin != u.depth
        if conv_shortcut:
            sk = Act(self, R, ho, wo, u.depth, p + '.shortcut')
            self.conv(self.fwd, p + '.shortcut', x.t, wts['ws'], sk.t, bias=wts['bs'], K=1, sn=st, pad=0)
        a = L.SeApplyDesc()
        a.skip, a.t, a.gate, a.out = _ptr(sk.t if conv_shortcut else x.t), _ptr(t2.t), _ptr(gate), _ptr(out.t)
        a.N, a.H, a.W, a.C, a.res_scale = R, ho, wo, u.depth, 1.0
        a.skip_mode = 0 if (conv_shortcut or st == 1) else 2             # MaxPool2d(1, 2): the even pixels of x
        self.fwd.add(a, p + '.merge')

        def backward():
            ps, pb = self.se_backward(p, out.g, t2, wts, gate, hid, ho * wo, res_scale=1.0)
            pro = dict(pro_scale=ps, pro_shift=pb, pro_per_row=1)
            if st == 1:
                self.conv(self.bwd, p + '.conv2^T', out.g, wts['w2_bwd'], t1.g, K=3, pad=1, dact_x=t1.t, dact_scale=wts['slope'],
                          dact_shift=wts['slope'], flags=L.GA_CONV_DACT_PRELU, **pro)
                t1.g_written = True
            else:
                self.grad_conv_up2(p + '.conv2^T', out.g, wts, 'w2_sub', t1, dact_x=t1.t, dact_scale=wts['slope'],
                                   dact_shift=wts['slope'], dact_prelu=True, **pro)
            # conv1^T: (W1^T dt1) * s0, plus the shortcut's share of d out
            identity = (not conv_shortcut) and st == 1
            self.grad_conv(p + '.conv1^T', t1.g, wts['w1_bwd'], x, K=3, pad=1, primary=out.g if identity else None,
                           dact_x=x.t, dact_scale=wts['pro_scale'], dact_shift=wts['pro_shift'], dact_act=L.GA_ACT_NONE)
            if conv_shortcut:
                if st == 1:
                    self.grad_conv(p + '.shortcut^T', out.g, wts['ws_bwd'], x, K=1)
                else:
                    self.grad_conv_up2(p + '.shortcut^T', out.g, wts, 'ws_sub', x)
            elif st == 2:                               # sub-sampled shortcut: d out lands on the even pixels of x
                il = L.Interleave2Desc()
                il.s[0] = _ptr(out.g)
                il.y, il.addend, il.N, il.H, il.W, il.C = _ptr(x.g), _ptr(x.g), R, h, w, u.depth
                self.bwd.add(il, p + '.shortcut^T')
        self._bwd_steps.append(backward)
        return out

    def _fpn_level(self, esd, name, top: Act, lat_src: Act) -> Act:
        """_upsample_add (helpers.py:122-139): bilinear x2 (align_corners=True) of `top` + 1x1 lateral conv of `lat_src`"""
        R = self.rows
        assert (lat_src.h, lat_src.w) == (2 * top.h, 2 * top.w), (name, top.h, lat_src.h)
        wts = self.devd('e4e.' + name, lambda: F.fold_e4e_lateral(esd, name))
        lat = Act(self, R, lat_src.h, lat_src.w, top.c, 'e4e.' + name)
        self.conv(self.fwd, 'e4e.' + name, lat_src.t, wts['w'], lat.t, bias=wts['b'], K=1)
        ones = self.devd(f'e4e.ones.{R}.{top.c}', lambda: {'g': torch.ones(R, top.c)})['g']
        out = Act(self, R, lat_src.h, lat_src.w, top.c, 'e4e.' + name + '.sum')
        a = L.SeApplyDesc()
        a.skip, a.t, a.gate, a.out = _ptr(top.t), _ptr(lat.t), _ptr(ones), _ptr(out.t)
        a.N, a.H, a.W, a.C, a.skip_mode, a.res_scale = R, lat_src.h, lat_src.w, top.c, 1, 1.0
        self.fwd.add(a, 'e4e.' + name + '.upsample_add')

        def backward():
            self.grad_conv('e4e.' + name + '^T', out.g, wts['w_bwd'], lat_src, K=1)
            b = L.BilinearBwdDesc()
            b.dhigh, b.dlow, b.N, b.h, b.w, b.C, b.accumulate = _ptr(out.g), _ptr(top.g), R, top.h, top.w, top.c, int(top.g_written)
            self.bwd.add(b, 'e4e.' + name + '.upsample^T')
            top.g_written = True
        self._bwd_steps.append(backward)
        return out

    def _style_head(self, esd, j, feat: Act, out: torch.Tensor, dout: torch.Tensor, g0: torch.Tensor):
        """GradualStyleBlock j (encoder.py:33-54) on `feat`; its latent goes to out[:, j] (+ out[:, 0] for j >= 1)"""
        es, R = self.vspec, self.rows
        D, cnt = es.style_dim, es.style_count
        p = f'e4e.styles.{j}'
        wts = self.devd(p, lambda: F.fold_e4e_style(esd, j, es.style_pools[j]))
        acts, cur = [], feat
        for k in range(es.style_pools[j]):
            act_in = L.GA_ACT_NONE if k == 0 else L.GA_ACT_LRELU
            if cur.h == 1:                              # a 3x3 / 2 conv on a 1x1 map is its centre tap
                nxt = Act(self, R, 1, 1, D, f'{p}.h{k}')
                self.conv(self.fwd, f'{p}.conv{k}', cur.t, wts[f'w{k}_c'], nxt.t, bias=wts[f'b{k}'], K=1, pro_act=act_in)
            else:
                nxt = Act(self, R, cur.h // 2, cur.w // 2, D, f'{p}.h{k}')
                self.conv(self.fwd, f'{p}.conv{k}', cur.t, wts[f'w{k}'], nxt.t, bias=wts[f'b{k}'], K=3, sn=2, pad=1, pro_act=act_in)
            acts.append((cur, nxt, act_in))
            cur = nxt
        assert cur.h == 1 and cur.w == 1, (p, cur.h)
        last = cur
        y = out.view(R, 1, 1, cnt * D)[..., j * D:(j + 1) * D]
        self.conv(self.fwd, f'{p}.linear', last.t, wts['wl'], y, bias=wts['bl'], K=1, pro_act=L.GA_ACT_LRELU, ldy=cnt * D,
                  addend=(out.view(R, 1, 1, cnt * D)[..., :D] if j else None), ldadd=cnt * D)

        def backward():
            dy = g0 if j == 0 else dout.view(R, 1, 1, cnt * D)[..., j * D:(j + 1) * D]
            self.grad_conv(f'{p}.linear^T', dy, wts['wl_bwd'], last, K=1, dact_x=last.t, dact_act=L.GA_ACT_LRELU,
                           ldx=(D if j == 0 else cnt * D))
            for k in reversed(range(len(acts))):
                src, dst, act_in = acts[k]
                dact = dict(dact_x=src.t, dact_act=act_in) if act_in else {}
                if src.h == 1:
                    self.grad_conv(f'{p}.conv{k}^T', dst.g, wts[f'w{k}_c_bwd'], src, K=1, **dact)
                else:
                    self.grad_conv_up2(f'{p}.conv{k}^T', dst.g, wts, f'w{k}_sub', src, **dact)
        self._bwd_steps.append(backward)

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

    def backward(self, from_logits: bool = True, from_purified: bool = False):
        """Backward-to-input of the last forward.  Cotangents are read from `self.dlogits` (rows x classes) when
        from_logits and from `self.dpurified` (NCHW) when from_purified.  May be called repeatedly per forward."""
        if not self.need_backward:
            raise RuntimeError('engine was built without a backward plan')
        if self.dry_run:
            raise RuntimeError('dry-run engine: plans were built for validation only')
        if self.dpurified is not None and not from_purified:
            self.dpurified.zero_()
        if from_logits:
            if getattr(self, '_g_bwd', None) is not None:
                self._launch_graph(self._g_bwd)
            else:
                self.bwd.run(self.stream())
        else:
            if not self.has_nvae:
                raise RuntimeError('classifier-only engine: backward starts from the logits')
            self._purified_grad_nhwc.g.zero_()
            self.bwd.run(self.stream(), start=self.bwd_split)
