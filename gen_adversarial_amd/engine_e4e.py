"""e4e encoder plans (Encoder4Editing, encoding/encoder.py:57-140).  Mixin of engine.Engine."""
from __future__ import annotations

from typing import Optional

import torch

from . import _lib as L
from . import folding as F
from .engine_core import IMG_LD, RES_SCALE, Act, _ptr


class E4EBuilder:
    # ------------------------------------------------------------------------------------------------ e4e encoder
    def _build_e4e(self, esd, img: Act, normalize: bool = False) -> torch.Tensor:
        """Encoder4Editing.forward (encoding/encoder.py:108-140; e4e_spec.py) on the NHWC image (taken as is: the caller's
        normalisation, if any, is part of its input — or, with normalize, Normalize(0.5, 0.5) of an image in [0, 1] as the
        input conv's prologue affine, abstract_models.py:177-178).  Returns the latents as [rows, style_count * 512]; `dlogits` is
        their cotangent."""
        es, R = self.vspec, self.rows
        if self.image_s2d:
            raise NotImplementedError
        x = self._e4e_input_layer(esd, es, img, normalize)
        src = self._e4e_body_fpn(esd, es, x)

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

    def _e4e_body_fpn(self, esd, es, x: Act):
        """the 24 IR-SE units and the FPN (encoder.py:113-128; the Style-Transformer's GradualStyleEncoder shares them,
        style_transformer_encoders.py:59-73): returns (c3, p2, p1)"""
        feats = {}
        for i, u in enumerate(es.units):
            x = self._ir_se_unit(esd, u, x)
            if i in es.taps:
                feats[es.taps.index(i)] = x
        c1, c2, c3 = feats[0], feats[1], feats[2]
        p2 = self._fpn_level(esd, 'latlayer1', c3, c2)
        p1 = self._fpn_level(esd, 'latlayer2', p2, c1)
        return c3, p2, p1

    def _e4e_input_layer(self, esd, es, img: Act, normalize: bool) -> Act:
        """input conv + BN + PReLU (encoder.py:70-74), optionally with Normalize(0.5, 0.5) as the conv's prologue affine"""
        R = self.rows
        inp = self.devd('e4e.input', lambda: F.fold_e4e_input(esd, IMG_LD))
        t0 = Act(self, R, img.h, img.w, es.base, 'e4e.input.conv')
        nrm, nrm_b = {}, {}
        if normalize:
            c = self.devd('norm05', lambda: {'two': torch.full((IMG_LD,), 2.0), 'mone': torch.full((IMG_LD,), -1.0)})
            nrm = dict(pro_scale=c['two'], pro_shift=c['mone'])
            nrm_b = dict(dact_x=img.t, dact_scale=c['two'], dact_shift=c['mone'], dact_act=L.GA_ACT_NONE)
        self.conv(self.fwd, 'e4e.input.conv', img.t, inp['w'], t0.t, bias=inp['b'], K=3, pad=1, **nrm)
        x = Act(self, R, img.h, img.w, es.base, 'e4e.input')
        pr = L.PreluDesc()
        pr.x, pr.slope, pr.y, pr.rows, pr.C, pr.backward = _ptr(t0.t), _ptr(inp['slope']), _ptr(x.t), R * img.h * img.w, es.base, 0
        self.fwd.add(pr, 'e4e.input.prelu')

        def bwd_input():
            b = L.PreluDesc()
            b.x, b.slope, b.dy, b.dx, b.rows, b.C, b.backward = (_ptr(t0.t), _ptr(inp['slope']), _ptr(x.g), _ptr(t0.g),
                                                                 R * img.h * img.w, es.base, 1)
            self.bwd.add(b, 'e4e.input.prelu^T')
            t0.g_written = True
            self.grad_conv('e4e.input.conv^T', t0.g, inp['w_bwd'], img, K=3, pad=1, **nrm_b)
        self._bwd_steps.append(bwd_input)
        return x

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

