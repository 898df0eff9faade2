"""
Plan builder for the StyleGAN2 synthesis network (StyleGan_E4E/stylegan2/generator.py): StyledConv with and without
up-sampling, ToRGB with its up-sampled skip, and Generator.forward's wiring for latent input and fixed noise buffers.

The reference builds per-sample weights [N*Cout, Cin, k, k] and runs a grouped convolution (generator.py:166-203); here
the weights stay shared: the style scales the INPUT channels in the conv's per-row prologue and the demodulation scales
the OUTPUT channels in the tail pass (folding.fold_styled_conv).  Forward / backward ops per layer:

  s      = modulation(w_latent)                 ga_conv2d 1x1 on [N,1,1,D]
  t      = conv(W, x * s)                       ga_conv2d, pro_per_row scale
  demod  = rsqrt(W2 s^2 + 1e-8)                 ga_unary(square), ga_conv2d 1x1, ga_unary(rsqrt)
  out    = act(demod * t + add)                 ga_modout
  ---- backward (dout given)
  dt     = dout * act'(u) * demod               ga_modout (u recomputed; sum_p dt t reduced in the same pass)
  d(W2 s^2) = -1/2 demod^2 sum_p dt t           ga_unary
  ds     = 2 s W2^T d(W2 s^2) + sum_p dxm x     ga_conv2d 1x1, ga_unary, ga_rowchan_reduce, ga_axpby
  dxm    = conv^T(W, dt);  dx = dxm * s         ga_conv2d; the row scale (accumulating into x.g) rides on the ga_rowchan_reduce pass
                                                that forms sum_p dxm x: one read of dxm for both
  dw_latent += modulation^T ds                  ga_conv2d 1x1
Up-sampling layer: transposed conv + blur = one 6x6 / stride-2 transposed conv (folding.upsample_conv_weights) = four 3x3
parity convs, run as ONE 3x3 conv Cin -> 4*Cout into the depth-to-space form, which the tail reads directly (ga_modout_desc.t_planes:
no interleave pass, no interleaved t); backward one 3x3 conv 4*Cout -> Cin over the cotangent that ga_modout writes in the same
depth-to-space form.  ToRGB skip: ga_up2_blur.
"""
from __future__ import annotations

import torch

from . import _lib as L
from . import folding as F
from .engine_core import IMG_LD, Act, _ptr
from .stylegan_spec import LR_MLP, N_MLP, StyledConvSpec


class LatentSlice:
    """latent[:, j] of a [N, n_latent, D] code tensor held in one Act [N,1,1,n_latent*D]: strided views + its own
    written-flag (several layers read the same index: generator.py:452-461)"""

    def __init__(self, latent: Act, j: int, dim: int):
        ld = getattr(latent, 'ld', latent.c)             # pitch of the code tensor (wider than latent.c for a latent prefix)
        self.n, self.c, self.ld = latent.n, dim, ld
        self.t = latent.t.view(latent.n, 1, 1, ld)[..., j * dim:(j + 1) * dim]
        self.g = latent.g.view(latent.n, 1, 1, ld)[..., j * dim:(j + 1) * dim]
        self.g_written = False


class StyleGanBuilder:
    def styled_conv(self, sd, spec: StyledConvSpec, x: Act, w_latent, noise: torch.Tensor = None, skip: Act = None,
                    need_dx: bool = True) -> Act:
        """emit one modulated convolution: x [N,r,r,Cin] (post-activation; r = res/2 for the up-sampling layer), w_latent an
        Act [N,1,1,D] or a LatentSlice -> Act [N,res,res,Cout'] (Cout' = Cout rounded up to 4 lanes; padded lanes hold zeros).
        skip: ToRGB's previous image at half the resolution (up-sampled and added).  Gradients flow to x.g, w_latent.g, skip.g."""
        R, p = self.rows, spec.prefix
        rin = spec.res // 2 if spec.upsample else spec.res
        assert (x.n, x.h, x.w, x.c) == (R, rin, rin, spec.cin) and (w_latent.n, w_latent.c) == (R, spec.style_dim), p
        lat_ld = getattr(w_latent, 'ld', w_latent.c)
        co = -(-spec.cout // 4) * 4
        nkey = 'none' if noise is None else f'{noise.data_ptr():x}'

        def fold():
            f = F.fold_styled_conv(sd, spec, noise, cout_pad=co)
            if spec.upsample:
                f.update(F.upsample_conv_weights(f.pop('w64')))
                f.pop('w'), f.pop('w_bwd'), f.pop('up_bwd')
                for a_ in (0, 1):
                    for b_ in (0, 1):
                        f.pop(f'up{a_}{b_}'), f.pop(f'up_bwd{a_}{b_}')
            f.pop('w64', None)
            return f
        wts = self.devd(f'sg.{p}.{spec.res}.{nkey}', fold)
        P, Pin, k = spec.res * spec.res, rin * rin, spec.kernel
        act = L.GA_ACT_FLRELU if spec.activate else L.GA_ACT_NONE

        s = Act(self, R, 1, 1, spec.cin, f'{p}.style')
        self.conv(self.fwd, f'{p}.modulation', w_latent.t, wts['wm'], s.t, bias=wts['bm'], K=1, ldx=lat_ld)
        zeros = self.devd(f'sg.zeros.{R}.{spec.cin}', lambda: {'z': torch.zeros(R, spec.cin)})['z']
        pro = dict(pro_scale=s.t, pro_shift=zeros, pro_per_row=1)
        t = s2d = None
        # ToRGB (no demodulation, no activation, no noise): the tail is `+ bias` — the conv's own bias; no tail pass in either direction,
        # the cotangent of t IS the cotangent of out
        plain = (not spec.demodulate) and (not spec.activate) and noise is None and not spec.upsample
        out = Act(self, R, spec.res, spec.res, co, f'{p}.out')
        if plain:
            t = out
            self.conv(self.fwd, f'{p}.conv', x.t, wts['w'], out.t, bias=wts['add'][0], K=k, pad=k // 2, **pro)
        elif spec.upsample:                # all four parities as one 3x3 conv Cin -> 4*Cout over the low-resolution input; t STAYS in
            # that depth-to-space form [R, rin, rin, 4*Cout] (channel block i = parity plane i): the tail and its adjoint read it through
            # ga_modout_desc.t_planes — no interleave pass, no interleaved copy of t or of its gradient (round 4)
            s2d = self.alloc((R, rin, rin, 4 * co))
            self.conv(self.fwd, f'{p}.conv[parities]', x.t, wts['up_all'], s2d, K=3, pad=1, **pro)
        else:
            t = Act(self, R, spec.res, spec.res, co, f'{p}.t')
            self.conv(self.fwd, f'{p}.conv', x.t, wts['w'], t.t, K=k, pad=k // 2, **pro)

        def tail_t(m):                   # where the tail finds t
            if s2d is None:
                m.t = _ptr(t.t)
                return
            m.W, m.ld_planes = spec.res, 4 * co
            for i in range(4):
                m.t_planes[i] = _ptr(s2d) + 4 * i * co
        demod = None
        if spec.demodulate:
            s2 = self.alloc((R, 1, 1, spec.cin))
            q = self.alloc((R, 1, 1, co))
            demod = self.alloc((R, 1, 1, co))
            self._unary(self.fwd, f'{p}.style^2', 0, s.t, None, s2)
            self.conv(self.fwd, f'{p}.demod_sum', s2, wts['w2'], q, K=1)
            self._unary(self.fwd, f'{p}.demod', 2, q, None, demod, eps=1e-8)
        if not plain:
            m = L.ModoutDesc()
            m.scale, m.add, m.out = _ptr(demod), _ptr(wts['add']), _ptr(out.t)
            m.N, m.P, m.C, m.act, m.backward = R, P, co, act, 0
            tail_t(m)
            self.fwd.add(m, f'{p}.tail')
        if skip is not None:
            assert (skip.n, skip.h, skip.w, skip.c) == (R, spec.res // 2, spec.res // 2, co), p
            u = L.Up2BlurDesc()
            u.lo_in, u.hi, u.N, u.H, u.W, u.C, u.backward = _ptr(skip.t), _ptr(out.t), R, skip.h, skip.w, co, 0
            self.fwd.add(u, f'{p}.skip_upsample')

        def backward():
            if skip is not None:
                assert not skip.g_written
                ub = L.Up2BlurDesc()
                ub.hi_in, ub.lo, ub.N, ub.H, ub.W, ub.C, ub.backward = _ptr(out.g), _ptr(skip.g), R, skip.h, skip.w, co, 1
                self.bwd.add(ub, f'{p}.skip_upsample^T')
                skip.g_written = True
            b = L.ModoutDesc()
            b.scale, b.add, b.dout = _ptr(demod), _ptr(wts['add']), _ptr(out.g)
            b.N, b.P, b.C, b.act, b.backward = R, P, co, act, 1
            if not plain:
                tail_t(b)
            s2d_g = None
            if plain:
                pass
            elif spec.upsample:                                    # dt in depth-to-space form only: the operand of the parity adjoint
                s2d_g = self.scratch((R, rin, rin, 4 * co), 'sg.up_s2d_g')
                for i in range(4):
                    b.dt_planes[i] = _ptr(s2d_g) + 4 * i * co
            else:
                b.dt = _ptr(t.g)
            gq = None
            if spec.demodulate:                                  # sum_p dt * t rides on the tail's adjoint (no second read of dt, t)
                gq = self.scratch((R, 1, 1, co), 'sg.gq')
                ws = self.scratch((256 * R * co,), 'sg.tail_ws')
                b.red, b.ws, b.ws_floats = _ptr(gq), _ptr(ws), ws.numel()
            if not plain:
                self.bwd.add(b, f'{p}.tail^T')
            ds = self.scratch((R, 1, 1, spec.cin), 'sg.ds')
            # ToRGB (1x1, 3 -> 4 output lanes): d(x*s) = W^T dt is formed inside the reduction pass from the 4-lane cotangent — the
            # Cin-wide tensor (4.3 GB at 1024 x 1024 x 32 channels x 32 rows) is neither written nor read (ga_rowchan_reduce a_src / a_w)
            torgb = k == 1 and not spec.upsample and co == 4 and tuple(wts['w_bwd'].shape) == (spec.cin, 4)
            dxm = None if torgb else self.scratch((R, rin, rin, spec.cin), 'sg.dxm')
            if spec.upsample:                                    # adjoint of (transposed conv + blur): one 3x3 conv 4*Cout -> Cin
                self.conv(self.bwd, f'{p}.conv^T[parities]', s2d_g, wts['up_all_bwd'], dxm, K=3, pad=1)
            elif not torgb:
                self.conv(self.bwd, f'{p}.conv^T', t.g, wts['w_bwd'], dxm, K=k, pad=k // 2)
            # sum_p dxm * x (style gradient) and d x = dxm * s (+ an already written x.g) from ONE read of dxm (round 4)
            self._reduce(f'{p}.' + ('conv^T+' if torgb else '') + 'dstyle_conv' + ('+dx' if need_dx else ''), dxm, x.t, ds, R, Pin, spec.cin,
                         scaled=(x.g if need_dx else None), gate=s.t, skip=(x.g if need_dx and x.g_written else None),
                         a_src=(t.g if torgb else None), a_w=(wts['w_bwd'] if torgb else None))
            if need_dx:
                x.g_written = True
            if spec.demodulate:
                ds2 = self.scratch((R, 1, 1, spec.cin), 'sg.ds2')
                self._unary(self.bwd, f'{p}.demod^T', 3, demod, gq, gq)
                self.conv(self.bwd, f'{p}.demod_sum^T', gq, wts['w2_bwd'], ds2, K=1)
                self._unary(self.bwd, f'{p}.style^2^T', 1, s.t, ds2, ds2)
                a = L.AxpbyDesc()
                a.x, a.y, a.n, a.alpha, a.beta = _ptr(ds2), _ptr(ds), R * spec.cin, 1.0, 1.0
                self.bwd.add(a, f'{p}.dstyle_sum')
            self.conv(self.bwd, f'{p}.modulation^T', ds, wts['wm_bwd'], w_latent.g, K=1, ldy=lat_ld,
                      addend=(w_latent.g if w_latent.g_written else None), ldadd=lat_ld)
            w_latent.g_written = True
        self._bwd_steps.append(backward)
        return out

    def build_stylegan(self, sd, spec, latent: Act) -> Act:
        """The synthesis network for styles=[latent], input_is_latent=True, randomize_noise=False (generator.py:399-470):
        latent Act [N,1,1,n_latent*D] -> image Act [N,size,size,4] (lane 3 is zero).  latent.g receives the gradient."""
        R, D = self.rows, spec.style_dim
        assert latent.c == spec.n_latent * D and latent.n == R
        lat = [LatentSlice(latent, j, D) for j in range(spec.n_latent)]
        noise = [sd[f'noises.noise_{i}'][0, 0] for i in range(1 + len(spec.convs))]
        const = Act(self, R, 4, 4, spec.const_channels, 'sg.const')          # ConstantInput (generator.py:214-224)
        const.t.copy_(sd['input.input'].permute(0, 2, 3, 1).expand(R, -1, -1, -1))
        out = self.styled_conv(sd, spec.conv1, const, lat[0], noise[0], need_dx=False)
        img = self.styled_conv(sd, spec.to_rgb1, out, lat[1])
        i = 1
        for k in range(len(spec.to_rgbs)):
            out = self.styled_conv(sd, spec.convs[2 * k], out, lat[i], noise[1 + 2 * k])
            out = self.styled_conv(sd, spec.convs[2 * k + 1], out, lat[i + 1], noise[2 + 2 * k])
            img = self.styled_conv(sd, spec.to_rgbs[k], out, lat[i + 2], skip=img)
            i += 2
        latent.g_written = True
        return img

    def build_mapping(self, gsd, z: torch.Tensor) -> torch.Tensor:
        """Generator.style (generator.py:306-317) on z [rows, D]: PixelNorm + 8 x (EqualLinear, fused leaky ReLU).  Forward only:
        on the defender's path z is fresh noise (src/defenses/ours/models.py:118-120)."""
        rows, D = z.shape
        wts = self.devd('sg.mapping', lambda: F.fold_mapping(gsd, N_MLP, LR_MLP))
        h = self.alloc((rows, 1, 1, D))
        pn = L.PixelnormDesc()
        pn.x, pn.y, pn.rows, pn.C = _ptr(z), _ptr(h), rows, D
        self.fwd.add(pn, 'sg.mapping.pixelnorm')
        for k in range(1, N_MLP + 1):
            nxt = self.alloc((rows, 1, 1, D))
            self.conv(self.fwd, f'sg.mapping.fc{k}', h, wts[f'w{k}'], nxt, bias=wts[f'b{k}'], K=1)
            m = L.ModoutDesc()
            m.t, m.out, m.N, m.P, m.C, m.act, m.backward = _ptr(nxt), _ptr(nxt), 1, rows, D, L.GA_ACT_FLRELU, 0
            self.fwd.add(m, f'sg.mapping.act{k}')
            h = nxt
        return h

    def build_e4e_defense(self, esd, espec, gsd, gspec, latent_avg, csd, cspec, pool_to: int):
        """E4EStyleGanDefenseModel.purify + classifier (src/defenses/ours/models.py:80-132; abstract_models.py:161-193) as one
        forward / backward plan pair:  image_io -> Normalize -> e4e encoder (+ latent_avg) -> mix with mapping(noise) ->
        StyleGAN2 synthesis -> face_pool + de-normalise -> ResNet classifier.  Caller-visible: x_in, eps[0] (the N(0,1) noise
        [rows, n_latent, D]), logits / dlogits, dx, purified_s2d; the mixing alphas live in a device buffer (set_alphas)."""
        R, J, D = self.rows, espec.style_count, espec.style_dim
        assert (gspec.n_latent, gspec.style_dim) == (J, D), 'encoder and generator disagree on the latent layout'
        assert gspec.size % pool_to == 0 and pool_to % 2 == 0, 'face_pool is built as a k x k mean'
        self.image_s2d = False
        x0 = self._build_input()                                               # enc_rows rows (one per image when shared)
        self.vspec = espec
        # With share_encoder (no input noise) the EoT replicas of an image are the same encoder input: the encoder runs once
        # per image, latent_mix reads code row r / rep and its adjoint sums the replicas' cotangents.  Same numbers as the
        # literal x.repeat(eot) path (wrappers.py:20), row for row.
        self.rows = self.enc_rows
        try:
            codes = self._build_e4e(esd, x0, normalize=True)                   # [enc_rows, J*D]; cotangent buffer = self.dlogits
        finally:
            self.rows = R
        dcodes = self.dlogits
        self.eps = [self.alloc((R, J, D))]
        styles = self.build_mapping(gsd, self.eps[0].view(R * J, D))
        avg = self.devd('sg.latent_avg', lambda: {'a': latent_avg.reshape(J, D)})['a'] if latent_avg is not None else None
        self.alpha_dev = self.alloc((J,))
        self.alpha_dev.copy_(torch.tensor(self.alphas, dtype=torch.float32))
        latent = Act(self, R, 1, 1, J * D, 'sg.latent')
        mx = L.LatentMixDesc()
        mx.codes, mx.avg, mx.styles, mx.alpha, mx.out = _ptr(codes), _ptr(avg), _ptr(styles), _ptr(self.alpha_dev), _ptr(latent.t)
        mx.R, mx.J, mx.D, mx.backward, mx.rep = R, J, D, 0, R // self.enc_rows
        self.fwd.add(mx, 'latent_mix')

        def bwd_mix():
            b = L.LatentMixDesc()
            b.alpha, b.dout, b.dcodes, b.R, b.J, b.D, b.backward = _ptr(self.alpha_dev), _ptr(latent.g), _ptr(dcodes), R, J, D, 1
            b.rep = R // self.enc_rows
            self.bwd.add(b, 'latent_mix^T')
        self._bwd_steps.append(bwd_mix)

        img = self.build_stylegan(gsd, gspec, latent)
        k = gspec.size // pool_to
        pooled = Act(self, R, pool_to // 2, pool_to // 2, 4 * IMG_LD, 'purified_s2d')
        pd = L.PoolDenormDesc()
        pd.x, pd.y, pd.N, pd.H, pd.W, pd.k, pd.ld, pd.backward = _ptr(img.t), _ptr(pooled.t), R, pool_to, pool_to, k, IMG_LD, 0
        self.fwd.add(pd, 'face_pool_denorm')

        self.purified = None                                     # API output comes from purified_nchw()
        self.dpurified = self.alloc((R, 3, pool_to, pool_to)) if self.need_backward else None   # cotangent on the returned image

        def bwd_pool():
            b = L.PoolDenormDesc()
            b.dy, b.dx, b.N, b.H, b.W, b.k, b.ld, b.backward = _ptr(pooled.g), _ptr(img.g), R, pool_to, pool_to, k, IMG_LD, 1
            b.dy_nchw = _ptr(self.dpurified)
            self.bwd.add(b, 'face_pool_denorm^T')
            img.g_written = True
        self._bwd_steps.append(bwd_pool)

        self.vspec, self.image_s2d = cspec, True
        n_purifier_steps = len(self._bwd_steps)                  # backward steps up to face_pool: replayed alone by
        self.logits = self._build_resnet(csd, pooled)            # backward(from_logits=False) (gradient from the purified image)
        self.image_s2d = False                                   # x0 (the defender's input) is a plain NHWC image
        self.purified_s2d = pooled
        self._purified_grad_nhwc = pooled
        self._finish(n_purifier_steps)
        return self

    def purified_nchw(self) -> torch.Tensor:
        """the purified image of the last forward, [rows, 3, H, W] in [0, 1] (classifier input before its normalisation)"""
        t = self.purified_s2d.t
        n, h2, w2, _ = t.shape
        v = t.view(n, h2, w2, 2, 2, IMG_LD)[..., :3]
        return v.permute(0, 5, 1, 3, 2, 4).reshape(n, 3, 2 * h2, 2 * w2).contiguous()

    def _unary(self, plan, name, mode, x, g, y, eps=0.0):
        u = L.UnaryDesc()
        u.x, u.g, u.y, u.n, u.mode, u.eps = _ptr(x), _ptr(g), _ptr(y), y.numel(), mode, eps
        plan.add(u, name)

    def _reduce(self, name, a, b, out, n, p, c, scaled=None, gate=None, skip=None, a_src=None, a_w=None):
        r = L.ReduceDesc()
        r.a, r.b, r.out, r.N, r.P, r.C, r.scale = _ptr(a), _ptr(b), _ptr(out), n, p, c, 1.0
        if a is None:
            r.a_src, r.a_w = _ptr(a_src), _ptr(a_w)
        if scaled is not None:
            r.scaled, r.gate, r.skip = _ptr(scaled), _ptr(gate), _ptr(skip)
        if p >= 4096:                                    # long rows: split the pixels over workgroups (two-stage reduction)
            ws = self.scratch((256 * n * c,), 'sg.reduce_ws')
            r.ws, r.ws_floats = _ptr(ws), ws.numel()
        self.bwd.add(r, name)
