"""
Plan builder for the StyleGAN2 synthesis layers (StyleGan_E4E/stylegan2/generator.py): this round the modulated
convolution without resampling — StyledConv(upsample=False) and ToRGB's convolution.

The reference builds per-sample weights [N*Cout, Cin, k, k] and runs a grouped convolution (generator.py:166-203); here
the weights stay shared: the style scales the INPUT channels in the conv's per-row prologue and the demodulation scales
the OUTPUT channels in the tail pass (folding.fold_styled_conv).  Forward / backward ops per layer:

  s      = modulation(w_latent)                 ga_conv2d 1x1 on [N,1,1,D]
  t      = conv(W, x * s)                       ga_conv2d, pro_per_row scale
  demod  = rsqrt(W2 s^2 + 1e-8)                 ga_unary(square), ga_conv2d 1x1, ga_unary(rsqrt)
  out    = act(demod * t + add)                 ga_modout
  ---- backward (dout given)
  dt     = dout * act'(u) * demod               ga_modout (u recomputed)
  d(W2 s^2) = -1/2 demod^2 sum_p dt t           ga_rowchan_reduce, ga_unary
  ds     = 2 s W2^T d(W2 s^2) + sum_p dxm x     ga_conv2d 1x1, ga_unary, ga_rowchan_reduce, ga_axpby
  dxm    = conv^T(W, dt);  dx = dxm * s         ga_conv2d, ga_se_apply (row scale, accumulating into x.g)
  dw_latent += modulation^T ds                  ga_conv2d 1x1
"""
from __future__ import annotations

import torch

from . import _lib as L
from . import folding as F
from .engine_core import Act, _ptr
from .stylegan_spec import StyledConvSpec


class StyleGanBuilder:
    def styled_conv(self, sd, spec: StyledConvSpec, x: Act, w_latent: Act, noise: torch.Tensor = None) -> Act:
        """emit one modulated convolution: x [N,res,res,Cin] (post-activation), w_latent [N,1,1,D] -> Act [N,res,res,Cout']
        (Cout' = Cout rounded up to 4 lanes; padded lanes hold zeros).  Gradients flow to x.g and w_latent.g."""
        R, p = self.rows, spec.prefix
        assert (x.n, x.h, x.w, x.c) == (R, spec.res, spec.res, spec.cin) and (w_latent.n, w_latent.c) == (R, spec.style_dim)
        co = -(-spec.cout // 4) * 4
        nkey = 'none' if noise is None else f'{noise.data_ptr():x}'
        wts = self.devd(f'sg.{p}.{spec.res}.{nkey}', lambda: F.fold_styled_conv(sd, spec, noise, cout_pad=co))
        P, k = spec.res * spec.res, spec.kernel
        act = L.GA_ACT_FLRELU if spec.activate else L.GA_ACT_NONE

        s = Act(self, R, 1, 1, spec.cin, f'{p}.style')
        self.conv(self.fwd, f'{p}.modulation', w_latent.t, wts['wm'], s.t, bias=wts['bm'], K=1)
        zeros = self.devd(f'sg.zeros.{R}.{spec.cin}', lambda: {'z': torch.zeros(R, spec.cin)})['z']
        t = Act(self, R, spec.res, spec.res, co, f'{p}.t')
        self.conv(self.fwd, f'{p}.conv', x.t, wts['w'], t.t, K=k, pad=k // 2, pro_scale=s.t, pro_shift=zeros, pro_per_row=1)
        demod = None
        if spec.demodulate:
            s2 = self.alloc((R, 1, 1, spec.cin))
            q = self.alloc((R, 1, 1, co))
            demod = self.alloc((R, 1, 1, co))
            self._unary(self.fwd, f'{p}.style^2', 0, s.t, None, s2)
            self.conv(self.fwd, f'{p}.demod_sum', s2, wts['w2'], q, K=1)
            self._unary(self.fwd, f'{p}.demod', 2, q, None, demod, eps=1e-8)
        out = Act(self, R, spec.res, spec.res, co, f'{p}.out')
        m = L.ModoutDesc()
        m.t, m.scale, m.add, m.out = _ptr(t.t), _ptr(demod), _ptr(wts['add']), _ptr(out.t)
        m.N, m.P, m.C, m.act, m.backward = R, P, co, act, 0
        self.fwd.add(m, f'{p}.tail')

        def backward():
            b = L.ModoutDesc()
            b.t, b.scale, b.add, b.dout, b.dt = _ptr(t.t), _ptr(demod), _ptr(wts['add']), _ptr(out.g), _ptr(t.g)
            b.N, b.P, b.C, b.act, b.backward = R, P, co, act, 1
            self.bwd.add(b, f'{p}.tail^T')
            ds = self.scratch((R, 1, 1, spec.cin), 'sg.ds')
            dxm = self.scratch((R, spec.res, spec.res, spec.cin), 'sg.dxm')
            self.conv(self.bwd, f'{p}.conv^T', t.g, wts['w_bwd'], dxm, K=k, pad=k // 2)
            self._reduce(f'{p}.dstyle_conv', dxm, x.t, ds, R, P, spec.cin)
            if spec.demodulate:
                gq = self.scratch((R, 1, 1, co), 'sg.gq')
                ds2 = self.scratch((R, 1, 1, spec.cin), 'sg.ds2')
                self._reduce(f'{p}.ddemod', t.g, t.t, gq, R, P, co)
                self._unary(self.bwd, f'{p}.demod^T', 3, demod, gq, gq)
                self.conv(self.bwd, f'{p}.demod_sum^T', gq, wts['w2_bwd'], ds2, K=1)
                self._unary(self.bwd, f'{p}.style^2^T', 1, s.t, ds2, ds2)
                a = L.AxpbyDesc()
                a.x, a.y, a.n, a.alpha, a.beta = _ptr(ds2), _ptr(ds), R * spec.cin, 1.0, 1.0
                self.bwd.add(a, f'{p}.dstyle_sum')
            self.grad_conv(f'{p}.modulation^T', ds, wts['wm_bwd'], w_latent, K=1)
            ap = L.SeApplyDesc()                                 # dx = dxm * s (+ an already written x.g)
            ap.skip = _ptr(x.g) if x.g_written else None
            ap.t, ap.gate, ap.out = _ptr(dxm), _ptr(s.t), _ptr(x.g)
            ap.N, ap.H, ap.W, ap.C, ap.skip_mode, ap.res_scale = R, spec.res, spec.res, spec.cin, 0, 1.0
            self.bwd.add(ap, f'{p}.dx')
            x.g_written = True
        self._bwd_steps.append(backward)
        return out

    def _unary(self, plan, name, mode, x, g, y, eps=0.0):
        u = L.UnaryDesc()
        u.x, u.g, u.y, u.n, u.mode, u.eps = _ptr(x), _ptr(g), _ptr(y), y.numel(), mode, eps
        plan.add(u, name)

    def _reduce(self, name, a, b, out, n, p, c):
        r = L.ReduceDesc()
        r.a, r.b, r.out, r.N, r.P, r.C, r.scale = _ptr(a), _ptr(b), _ptr(out), n, p, c, 1.0
        self.bwd.add(r, name)
