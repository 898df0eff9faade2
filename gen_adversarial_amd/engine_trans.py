"""
Plan builder for the Style-Transformer defender: GradualStyleEncoder
(src/mlvgms_autoencoders/StyleGan_Trans/models/encoders/style_transformer_encoders.py:10-85) = the IR-SE50 trunk + FPN of the
e4e encoder (engine_e4e) followed by three DETR post-norm decoder layers (models/transformer.py:17-100) over 16 style queries,
and TransStyleGanDefenseModel.purify around it (src/defenses/ours/models.py:299-353).  Mixin of engine.Engine.

Tokens are NHWC tensors [rows, T, 1, C]: every linear layer of the transformer (in_proj, out_proj, FFN) is a 1x1 ga_conv2d over
the token axis (M = rows x T on the matrix cores); the feature maps c3 / p2 / p1 ARE their own token lists ([rows, h*w, C]), so
the memory needs no flatten / permute copy.  Per decoder layer (post-norm, dropout inactive, no positional encodings):
    qkv = in_proj(tgt); a = out_proj(attn(q, k, v));                          t1 = LN1(tgt + a)
    q = Wq t1; kv = Wkv mem; c = out_proj(attn(q, k, v));                      t2 = LN2(t1 + c)
    f = W2 relu(W1 t2);                                                        t3 = LN3(t2 + f)
The attention core and LayerNorm are ga_attn / ga_layernorm (csrc/attention.hip).
"""
from __future__ import annotations

import math

import torch

from . import _lib as L
from . import folding as F
from .engine_core import IMG_LD, Act, _ptr
from .stylegan_spec import LR_MLP, N_MLP
from .trans_spec import LAYERS


class TokenView:
    """a feature map Act [N,h,w,C] seen as its token list [N, h*w, 1, C] (same memory); gradient bookkeeping stays with the Act"""

    def __init__(self, act: Act):
        self.act, self.n, self.h, self.w, self.c = act, act.n, act.h * act.w, 1, act.c
        self.t = act.t.view(act.n, act.h * act.w, 1, act.c)

    @property
    def g(self):
        return self.act.g.view(self.n, self.h, 1, self.c)

    @property
    def g_written(self):
        return self.act.g_written

    @g_written.setter
    def g_written(self, v):
        self.act.g_written = v


class TransBuilder:
    def _linear(self, name, x, w, b, cout, pro_act=0, precise=False) -> Act:
        y = Act(self, x.n, x.h, 1, cout, name)
        self.conv(self.fwd, name, x.t, w, y.t, bias=b, K=1, pro_act=pro_act, precise=precise)
        return y

    def _attention(self, name, q_t, k_t, v_t, ldq, ldk, Tk, d, nhead, dq_t, dk_t, dv_t) -> Act:
        """softmax(q k^T / sqrt(dh)) v per head; q_t / k_t / v_t are (views of) projected token tensors.  The backward op WRITES
        dq_t / dk_t / dv_t (views of the gradients of those projections).  Returns the head-concatenated output Act."""
        R, Tq = self.rows, 16
        out = Act(self, R, Tq, 1, d, name + '.heads')
        p = self.alloc((R, nhead, Tq, Tk))
        a = L.AttnDesc()
        a.q, a.k, a.v, a.out, a.p = _ptr(q_t), _ptr(k_t), _ptr(v_t), _ptr(out.t), _ptr(p)
        a.ldq, a.ldk, a.ldv, a.ldo = ldq, ldk, ldk, d
        a.N, a.Tq, a.Tk, a.heads, a.dh, a.scale, a.backward = R, Tq, Tk, nhead, d // nhead, 1.0 / math.sqrt(d // nhead), 0
        self.fwd.add(a, name)

        def backward():
            b = L.AttnDesc()
            ds = self.scratch((R, nhead, Tq, Tk), 'attn.ds')
            b.q, b.k, b.v, b.p, b.dout, b.ds = _ptr(q_t), _ptr(k_t), _ptr(v_t), _ptr(p), _ptr(out.g), _ptr(ds)
            b.dq, b.dk, b.dv = _ptr(dq_t()), _ptr(dk_t()), _ptr(dv_t())
            b.ldq, b.ldk, b.ldv, b.ldo, b.lddq, b.lddk, b.lddv = ldq, ldk, ldk, d, ldq, ldk, ldk
            b.N, b.Tq, b.Tk, b.heads, b.dh, b.scale, b.backward = R, Tq, Tk, nhead, d // nhead, 1.0 / math.sqrt(d // nhead), 1
            self.bwd.add(b, name + '^T')
        self._bwd_steps.append(backward)
        return out

    def _add_norm(self, name, res, sub: Act, gamma, beta, eps, res_needs_grad=True) -> Act:
        """LayerNorm(res + sub): the cotangent of the sum goes to sub.g (written) and is added into res.g"""
        R, T, C = sub.n, sub.h, sub.c
        y = Act(self, R, T, 1, C, name)
        stats = self.alloc((R * T, 2))
        d = L.LayernormDesc()
        d.a, d.b, d.gamma, d.beta, d.y, d.stats = _ptr(res.t), _ptr(sub.t), _ptr(gamma), _ptr(beta), _ptr(y.t), _ptr(stats)
        d.rows, d.C, d.eps, d.backward = R * T, C, eps, 0
        self.fwd.add(d, name)

        def backward():
            b = L.LayernormDesc()
            b.a, b.b, b.gamma, b.stats, b.dy, b.dx = _ptr(res.t), _ptr(sub.t), _ptr(gamma), _ptr(stats), _ptr(y.g), _ptr(sub.g)
            b.rows, b.C, b.eps, b.backward, b.accumulate = R * T, C, eps, 1, 0
            self.bwd.add(b, name + '^T')
            sub.g_written = True
            if res_needs_grad:
                ax = L.AxpbyDesc()
                ax.x, ax.y, ax.n, ax.alpha, ax.beta = _ptr(sub.g), _ptr(res.g), R * T * C, 1.0, 1.0 if res.g_written else 0.0
                self.bwd.add(ax, name + '.residual^T')
                res.g_written = True
        self._bwd_steps.append(backward)
        return y

    def _decoder_layer(self, tsd, spec, name, tgt: Act, mem, tgt_needs_grad: bool) -> Act:
        """TransformerDecoderLayer.forward_post (transformer.py:42-64) on 16 queries `tgt` [R,16,1,d] against the tokens of `mem`"""
        w = self.devd('trans.' + name, lambda: F.fold_trans_layer(tsd, name))
        d, nh, R = spec.d_model, spec.nhead, self.rows
        p = 'trans.' + name
        steps = self._bwd_steps

        def grab(n_expected, fn, *a, **kw):
            """run a forward emitter and take the backward closures it registered OUT of the shared list: this layer orders
            its backward steps itself (below); a helper that starts registering more (or fewer) steps fails here, loudly"""
            n0 = len(steps)
            r = fn(*a, **kw)
            new = steps[n0:]
            del steps[n0:]
            assert len(new) == n_expected, (name, getattr(fn, '__name__', fn), len(new), n_expected)
            return r, new

        # ---- self attention: one GEMM projects q, k and v of the same 16 tokens
        # the projections in front of a softmax run on the exact fp32 kernel (a few GFLOP per row): logits of tens (random or
        # sharp trained heads) turn the split-bf16 kernels' 1e-5 relative error into 1e-3 on the gradient
        qkv, _ = grab(0, self._linear, p + '.sa.in_proj', tgt, w['sa_qkv_w'], w['sa_qkv_b'], 3 * d, precise=True)

        def third(i):
            return lambda: qkv.g.view(R, 16, 1, 3 * d)[..., i * d:(i + 1) * d]
        v3 = qkv.t.view(R, 16, 1, 3 * d)
        heads, (sa_attn,) = grab(1, self._attention, p + '.sa.attn', v3[..., :d], v3[..., d:2 * d], v3[..., 2 * d:], 3 * d, 3 * d, 16, d,
                                 nh, third(0), third(1), third(2))
        a, _ = grab(0, self._linear, p + '.sa.out_proj', heads, w['sa_out_w'], w['sa_out_b'], d)

        def sa_out():
            self.grad_conv(p + '.sa.out_proj^T', a.g, w['sa_out_w_bwd'], heads, K=1)
        t1, (n1,) = grab(1, self._add_norm, p + '.norm1', tgt, a, w['ln1_g'], w['ln1_b'], spec.eps, res_needs_grad=tgt_needs_grad)
        # ---- cross attention to the memory tokens
        q2, _ = grab(0, self._linear, p + '.ca.q_proj', t1, w['ca_q_w'], w['ca_q_b'], d, precise=True)
        Tm = mem.h
        kv = Act(self, R, Tm, 1, 2 * d, p + '.ca.kv_proj')
        kvv = kv.t.view(R, Tm, 1, 2 * d)
        self.conv(self.fwd, p + '.ca.k_proj', mem.t, w['ca_kv_w'][:d], kvv[..., :d], bias=w['ca_kv_b'][:d], K=1, ldy=2 * d, precise=True)
        self.conv(self.fwd, p + '.ca.v_proj', mem.t, w['ca_kv_w'][d:], kvv[..., d:], bias=w['ca_kv_b'][d:], K=1, ldy=2 * d)
        heads2, (ca_attn,) = grab(1, self._attention, p + '.ca.attn', q2.t, kvv[..., :d], kvv[..., d:], d, 2 * d, Tm, d, nh,
                                  lambda: q2.g, lambda: kv.g.view(R, Tm, 1, 2 * d)[..., :d], lambda: kv.g.view(R, Tm, 1, 2 * d)[..., d:])
        c, _ = grab(0, self._linear, p + '.ca.out_proj', heads2, w['ca_out_w'], w['ca_out_b'], d)

        def ca_out():
            self.grad_conv(p + '.ca.out_proj^T', c.g, w['ca_out_w_bwd'], heads2, K=1)
        t2, (n2,) = grab(1, self._add_norm, p + '.norm2', t1, c, w['ln2_g'], w['ln2_b'], spec.eps)
        # ---- feed forward
        h, _ = grab(0, self._linear, p + '.ff.linear1', t2, w['ff1_w'], w['ff1_b'], spec.dff)
        f, _ = grab(0, self._linear, p + '.ff.linear2', h, w['ff2_w'], w['ff2_b'], d, pro_act=L.GA_ACT_RELU)
        t3, (n3,) = grab(1, self._add_norm, p + '.norm3', t2, f, w['ln3_g'], w['ln3_b'], spec.eps)

        def bwd_ff():
            self.grad_conv(p + '.ff.linear2^T', f.g, w['ff2_w_bwd'], h, K=1, dact_x=h.t, dact_act=L.GA_ACT_RELU)
            self.grad_conv(p + '.ff.linear1^T', h.g, w['ff1_w_bwd'], t2, K=1)

        def bwd_ca_in():
            q2.g_written = kv.g_written = True            # written by the attention backward
            self.grad_conv(p + '.ca.kv_proj^T', kv.g, w['ca_kv_w_bwd'], mem, K=1)
            self.grad_conv(p + '.ca.q_proj^T', q2.g, w['ca_q_w_bwd'], t1, K=1)

        def bwd_sa_in():
            qkv.g_written = True
            if tgt_needs_grad:
                self.grad_conv(p + '.sa.in_proj^T', qkv.g, w['sa_qkv_w_bwd'], tgt, K=1)
        # this layer's backward steps in registration (= forward) order; the reversed list replays
        # [norm3^T, ff, norm2^T, ca.out^T, ca.attn^T, ca_in, norm1^T, sa.out^T, sa.attn^T, sa_in]
        steps.extend([bwd_sa_in, sa_attn, sa_out, n1, bwd_ca_in, ca_attn, ca_out, n2, bwd_ff, n3])
        return t3

    def _build_trans_encoder(self, tsd, spec, gsd, img: Act):
        """GradualStyleEncoder.forward(x, style(z)) on the normalised-by-prologue image Act; returns the codes Act [R,16,1,d]"""
        es, R = spec.trunk, self.rows
        x = self._e4e_input_layer(tsd, es, img, normalize=True)
        c3, p2, p1 = self._e4e_body_fpn(tsd, es, x)
        qconst = self.devd('trans.queries', lambda: {'q': F.trans_queries(tsd, gsd, N_MLP, LR_MLP)})['q']        # [16, d]
        tgt = Act(self, R, spec.n_query, 1, spec.d_model, 'trans.queries')
        tgt.t.copy_(qconst.view(1, spec.n_query, 1, spec.d_model).expand(R, -1, -1, -1))
        for i, (name, mem) in enumerate(zip(LAYERS, (c3, p2, p1))):
            tgt = self._decoder_layer(tsd, spec, name, tgt, TokenView(mem), tgt_needs_grad=i > 0)
        return tgt

    def build_trans_defense(self, tsd, tspec, gsd, gspec, latent_avg, csd, cspec, pool_to: int, mid: int = 256, crop: int = 32,
                            noise_std: float = 0.8):
        """TransStyleGanDefenseModel.purify + classifier (src/defenses/ours/models.py:299-353; abstract_models.py:161-193) as one
        forward / backward plan pair:  image_io -> resize x2 + crop (kornia resize to `mid`, rows crop:-crop) -> Normalize ->
        GradualStyleEncoder (+ latent_avg) -> mix with mapping(N(0, noise_std)) -> StyleGAN2 synthesis -> face_pool + the -1 band
        + resize to the input size (one k x k mean: both resizes are exact 2 x 2 means) + de-normalise -> ResNeXt classifier.
        The caller fills eps[0] with N(0, 1) draws; `eps_std` tells the API to scale them (models.py:331 draws N(0, 0.8))."""
        R, J, D = self.rows, tspec.n_query, tspec.d_model
        H = self.resolution[1]
        assert mid == 2 * H, 'the resize in front of the encoder is built as a bilinear x2'
        assert gspec.style_dim == D and gspec.n_latent <= J, 'encoder and generator disagree on the latent layout'
        # face_pool (AdaptiveAvgPool to 256) + resize to the input size = ONE k x k mean when the generator is at least as large as
        # the input image (512 -> 128: k = 4); reduced test generators smaller than the input keep their own resolution (the
        # reference's pooling then only replicates pixels), as the e4e defender does
        assert pool_to == min(H, gspec.size) and gspec.size % pool_to == 0 and pool_to % 2 == 0, 'face_pool + resize are built as one k x k mean'
        self.image_s2d = False
        x0 = self._build_input()                                               # enc_rows rows (one per image when shared)
        self.rows = self.enc_rows
        try:
            up = Act(self, self.rows, mid - 2 * crop, mid, IMG_LD, 'trans.resized')
            rc = L.Resize2CropDesc()
            rc.x, rc.y, rc.N, rc.H, rc.W, rc.C, rc.crop, rc.backward = _ptr(x0.t), _ptr(up.t), self.rows, H, H, IMG_LD, crop, 0
            self.fwd.add(rc, 'trans.resize_crop')
            n_enc = self.rows

            def bwd_resize():
                b = L.Resize2CropDesc()
                b.dy, b.dx, b.N, b.H, b.W, b.C, b.crop, b.backward, b.accumulate = _ptr(up.g), _ptr(x0.g), n_enc, H, H, IMG_LD, crop, 1, 0
                self.bwd.add(b, 'trans.resize_crop^T')
                x0.g_written = True
            self._bwd_steps.append(bwd_resize)
            codes_act = self._build_trans_encoder(tsd, tspec, gsd, up)          # [enc_rows, 16, 1, D]
        finally:
            self.rows = R
        self.eps = [self.alloc((R, J, D))]
        self.eps_std = float(noise_std)
        styles = self.build_mapping(gsd, self.eps[0].view(R * J, D))
        avg = self.devd('sg.latent_avg', lambda: {'a': latent_avg.reshape(J, D)})['a'] if latent_avg is not None else None
        self.alpha_dev = self.alloc((J,))
        self.alpha_dev.copy_(torch.tensor(self.alphas, dtype=torch.float32))
        latent = Act(self, R, 1, 1, J * D, 'sg.latent')
        mx = L.LatentMixDesc()
        mx.codes, mx.avg, mx.styles, mx.alpha, mx.out = _ptr(codes_act.t), _ptr(avg), _ptr(styles), _ptr(self.alpha_dev), _ptr(latent.t)
        mx.R, mx.J, mx.D, mx.backward, mx.rep = R, J, D, 0, R // self.enc_rows
        self.fwd.add(mx, 'latent_mix')

        def bwd_mix():
            b = L.LatentMixDesc()
            b.alpha, b.dout, b.dcodes, b.R, b.J, b.D, b.backward = _ptr(self.alpha_dev), _ptr(latent.g), _ptr(codes_act.g), R, J, D, 1
            b.rep = R // self.enc_rows
            self.bwd.add(b, 'latent_mix^T')
            codes_act.g_written = True
        self._bwd_steps.append(bwd_mix)

        # the generator reads its first n_latent indices (16 for the 512-px model; fewer in reduced test models)
        if gspec.n_latent == J:
            glat = latent
        else:
            glat = _LatentPrefix(latent, gspec.n_latent * D)
        img = self.build_stylegan(gsd, gspec, glat)
        k = gspec.size // pool_to
        band = crop * pool_to // mid
        pooled = Act(self, R, pool_to // 2, pool_to // 2, 4 * IMG_LD, 'purified_s2d')
        pd = L.PoolDenormDesc()
        pd.x, pd.y, pd.N, pd.H, pd.W, pd.k, pd.ld, pd.backward, pd.band = _ptr(img.t), _ptr(pooled.t), R, pool_to, pool_to, k, IMG_LD, 0, band
        self.fwd.add(pd, 'face_pool_band_resize_denorm')
        self.purified = None
        self.dpurified = self.alloc((R, 3, pool_to, pool_to)) if self.need_backward else None

        def bwd_pool():
            b = L.PoolDenormDesc()
            b.dy, b.dx, b.N, b.H, b.W, b.k, b.ld, b.backward, b.band = _ptr(pooled.g), _ptr(img.g), R, pool_to, pool_to, k, IMG_LD, 1, band
            b.dy_nchw = _ptr(self.dpurified)
            self.bwd.add(b, 'face_pool_band_resize_denorm^T')
            img.g_written = True
        self._bwd_steps.append(bwd_pool)

        self.vspec, self.image_s2d = cspec, True
        n_purifier_steps = len(self._bwd_steps)
        self.logits = self._build_resnet(csd, pooled)
        self.image_s2d = False
        self.purified_s2d = pooled
        self._purified_grad_nhwc = pooled
        self._finish(n_purifier_steps)
        return self


class _LatentPrefix:
    """a latent Act [N,1,1,J*D] of which the generator reads the first `c` channels (reduced test generators have fewer latent
    indices than the encoder's 16 queries): same memory and pitch, narrower logical width"""

    def __init__(self, latent: Act, c: int):
        self.base, self.n, self.c, self.ld = latent, latent.n, c, latent.c
        self.t = latent.t

    @property
    def g(self):
        return self.base.g

    @property
    def g_written(self):
        return self.base.g_written

    @g_written.setter
    def g_written(self, v):
        self.base.g_written = v
