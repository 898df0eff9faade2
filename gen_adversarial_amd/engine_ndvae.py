"""ND-VAE competitor defender plans (`NDVaeDefenseModel` over `Defence_NVAE`):
src/defenses/competitors/nd_vae/purification_model.py:18-31, src/defenses/competitors/nd_vae/modules/models/NVAE.py:639-720.
Mixin of engine.Engine; the cells are the NVAE builders' (engine_nvae.enc_cell / dec_cell) with the competitor's folding."""
from __future__ import annotations

from typing import List

import torch

from . import _lib as L
from . import folding as F
from .engine_core import IMG_LD, Act, _ptr
from .ndvae_spec import NdvaeSpec
from .resnet_spec import ResNetSpec


def _fold_conv(sd, prefix: str) -> dict:
    w = sd[f'{prefix}.weight'].double()
    return {'w': F.f32(F.conv_fwd_layout(w)), 'w_bwd': F.f32(F.conv_bwd_layout(w)), 'b': F.f32(sd[f'{prefix}.bias'].double())}


class NdvaeBuilder:
    def _nd_sampler(self, idx: int, x: Act, eps: torch.Tensor) -> Act:
        """Sampler.forward (NVAE.py:608-634): prior = 1x1(ELU(x)), posterior parameters = 3x3(x), z = mu + sigma * eps with the
        two distributions' parameters summed (ga_sampler_mix mode 1)."""
        sd, p, ch = self.nvae_sd, f'decoder.samplers.{idx}', x.c
        wq = self.devd(f'nd.{p}.cell', lambda: _fold_conv(sd, f'{p}.cell'))
        wp = self.devd(f'nd.{p}.prior', lambda: _fold_conv(sd, f'{p}.prior_cell.1'))
        q = Act(self, x.n, x.h, x.w, 2 * ch, f'{p}.q')
        pr = Act(self, x.n, x.h, x.w, 2 * ch, f'{p}.p')
        self.conv(self.fwd, f'{p}.cell', x.t, wq['w'], q.t, bias=wq['b'], K=3, pad=1)
        self.conv(self.fwd, f'{p}.prior', x.t, wp['w'], pr.t, bias=wp['b'], K=1, pro_act=L.GA_ACT_ELU)
        z = Act(self, x.n, x.h, x.w, ch, f'{p}.z')

        def desc(backward: int) -> L.SamplerDesc:
            d = L.SamplerDesc()
            d.mu_q, d.ldq, d.p, d.ldp, d.eps, d.eps_nchw = _ptr(q.t), 2 * ch, _ptr(pr.t), 2 * ch, _ptr(eps), 1
            d.N, d.h, d.w, d.NL, d.ldz, d.mode, d.backward = x.n * (self.cot_rep if backward else 1), x.h, x.w, ch, ch, 1, backward
            d.alpha, d.one_minus_alpha, d.temp = 0.0, 1.0, 1.0
            return d
        f = desc(0)
        f.z = _ptr(z.t)
        self.fwd.add(f, f'{p}.sample')

        def backward():
            b = desc(1)
            b.dz, b.dmu_q, b.dp, b.act_rep = _ptr(z.g), _ptr(q.g), _ptr(pr.g), self.cot_rep
            self.bwd.add(b, f'{p}.sample^T')
            q.g_written = pr.g_written = True
            self.grad_conv(f'{p}.cell^T', q.g, wq['w_bwd'], x, K=3, pad=1)
            self.grad_conv(f'{p}.prior^T', pr.g, wp['w_bwd'], x, K=1, dact_x=x.t, dact_act=L.GA_ACT_ELU)
        self._bwd_steps.append(backward)
        return z

    def _nd_cat_conv(self, name: str, a: Act, b: Act) -> Act:
        """DecCombinerCell (NVAE.py:243-251): conv1x1(cat[a, b]) as a two-pointer K loop; backward = the two halves of W^T"""
        sd = self.nvae_sd

        def fold():
            w = sd[f'{name}.weight'].double()[:, :, 0, 0]                        # [C, Ca + Cb]
            return {'w': F.f32(w), 'b': F.f32(sd[f'{name}.bias'].double()), 'a_bwd': F.f32(w[:, :a.c].t()), 'b_bwd': F.f32(w[:, a.c:].t())}
        wts = self.devd('nd.' + name, fold)
        out = Act(self, a.n, a.h, a.w, wts['w'].shape[0], name)
        self.conv(self.fwd, name, a.t, wts['w'], out.t, bias=wts['b'], K=1, x2=b.t)

        def backward():
            self.grad_conv(name + '^T.b', out.g, wts['b_bwd'], b, K=1)
            self.grad_conv(name + '^T.a', out.g, wts['a_bwd'], a, K=1)
        self._bwd_steps.append(backward)
        return out

    def build_ndvae_defense(self, nsd, nspec: NdvaeSpec, h: torch.Tensor, csd, cspec):
        """NDVaeDefenseModel.forward (purification_model.py:28-31) as one forward / backward plan pair: image_io (x + noise * std,
        clamp) -> Defence_NVAE -> DiscMixLogistic mean -> classifier.  Caller-visible: x_in, noise / noise_coef (the API fills
        noise_coef with the constant noise_std: purification_model.py:21 scales the draw, it does not normalise it), eps (one
        N(0,1) tensor per sampler, NCHW like the reference draws them), logits / dlogits, dx, purified / dpurified."""
        R, D = self.rows, nspec.input_dim
        assert self.resolution[1] == D, (self.resolution, D)
        self.nvae_sd, self.ndspec = nsd, nspec
        self.noise_is_std = True                            # noise_coef = noise_std for every row (the API fills it)
        self.image_s2d = False
        self.share_encoder, self.enc_rows = False, R        # every run draws input noise: nothing in front of it to share
        x0 = self._build_input()
        self.eps = [self.alloc((R, c, r, r)) for c, r in nspec.latent_shapes]
        self.purified = self.alloc((R, 3, D, D))

        # ---- stem: (clamp is the image boundary's) x * 2 - 1 as the prologue affine of the 3x3 stem (NVAE.py:690-693)
        stem = self.devd('nd.stem', lambda: F.pad_image_conv(_fold_conv(nsd, 'stem'), 3, IMG_LD))
        norm = self.devd('norm05', lambda: {'two': torch.full((IMG_LD,), 2.0), 'mone': torch.full((IMG_LD,), -1.0)})
        two, mone = norm['two'], norm['mone']
        x = Act(self, R, D, D, nspec.base, 'nd.stem')
        self.conv(self.fwd, 'nd.stem', x0.t, stem['w'], x.t, bias=stem['b'], K=3, pad=1, pro_scale=two, pro_shift=mone)
        stem_out = x

        def bwd_stem():
            self.grad_conv('nd.stem^T', stem_out.g, stem['w_bwd'], x0, K=3, pad=1,
                           dact_x=x0.t, dact_scale=two, dact_shift=mone, dact_act=L.GA_ACT_NONE)
        self._bwd_steps.append(bwd_stem)

        for cell in nspec.pre_cells:
            x = self.enc_cell(cell, x)
        outs: List[Act] = [x]
        for cells in nspec.enc_scales:
            for cell in cells:
                x = self.enc_cell(cell, x)
            outs.append(x)
        latent = outs[::-1]

        # ---- decoder (Decoder_tower.forward, NVAE.py:547-575)
        z = self._nd_sampler(0, latent[0], self.eps[0])

        def fold_comb0():           # conv1x1(cat[z, h]): the h half is the same for every row -> a broadcast addend
            w = nsd['decoder.combiner_cells.0.conv.weight'].double()[:, :, 0, 0]
            cz = z.c
            ph = torch.einsum('oc,chw->hwo', w[:, cz:], h.double()) + nsd['decoder.combiner_cells.0.conv.bias'].double()
            return {'ph': ph.float().unsqueeze(0), 'wz': F.f32(w[:, :cz]), 'wz_bwd': F.f32(w[:, :cz].t())}
        c0 = self.devd('nd.combiner_0', fold_comb0)
        out = Act(self, R, z.h, z.w, c0['wz'].shape[0], 'nd.comb0')
        self.conv(self.fwd, 'nd.combiner_0', z.t, c0['wz'], out.t, K=1, addend=c0['ph'], addend_bcast=True)
        comb0_out, z0 = out, z

        def bwd_comb0():
            self.grad_conv('nd.combiner_0^T', comb0_out.g, c0['wz_bwd'], z0, K=1)
        self._bwd_steps.append(bwd_comb0)

        for s, sc in enumerate(nspec.dec_scales):
            y = out
            for grp in sc.groups:
                t = y
                for cell in grp.cells:
                    t = self.dec_cell(cell, t)
                y = self._nd_cat_conv(f'{grp.prefix}.combiner.conv', y, t)
            if sc.up is not None:
                y = self.dec_cell(sc.up, y)
            # EncCombinerCell (NVAE.py:231-240): latent + conv1x1(y)
            ecn = f'encoder.combiner_cells.{s}.conv'
            ecw = self.devd('nd.' + ecn, lambda ecn=ecn: {k: v for k, v in (
                ('w', F.f32(nsd[f'{ecn}.weight'].double()[:, :, 0, 0])), ('b', F.f32(nsd[f'{ecn}.bias'].double())),
                ('w_bwd', F.f32(nsd[f'{ecn}.weight'].double()[:, :, 0, 0].t())))})
            feat = latent[s + 1]
            ec = Act(self, R, y.h, y.w, feat.c, f'nd.ec{s}')
            self.conv(self.fwd, ecn, y.t, ecw['w'], ec.t, bias=ecw['b'], K=1, addend=feat.t)

            def bwd_ec(ec=ec, y=y, feat=feat, ecw=ecw, ecn=ecn):
                self.grad_conv(ecn + '^T', ec.g, ecw['w_bwd'], y, K=1)
                a = L.AxpbyDesc()                           # the additive encoder feature receives d(ec) unchanged
                a.x, a.y, a.n, a.alpha, a.beta = _ptr(ec.g), _ptr(feat.g), ec.g.numel(), 1.0, 1.0 if feat.g_written else 0.0
                feat.g_written = True
                self.bwd.add(a, f'nd.enc_feat{s}.grad')
            self._bwd_steps.append(bwd_ec)
            z = self._nd_sampler(s + 1, ec, self.eps[s + 1])
            out = self._nd_cat_conv(f'decoder.combiner_cells.{s + 1}.conv', z, y)

        x = out
        for cell in nspec.post_cells:
            x = self.dec_cell(cell, x)

        # ---- image_conditional (ELU -> 3x3, NVAE.py:667) + DiscMixLogistic.mean (NVAE_utils.py:224-248)
        LO = (nspec.logits_out + 7) // 8 * 8
        tl = self.devd('nd.image_conditional', lambda: F.pad_conv_out(_fold_conv(nsd, 'image_conditional.1'), nspec.logits_out, LO))
        logits = Act(self, R, D, D, LO, 'nd.mix_logits')
        post_out = x
        self.conv(self.fwd, 'nd.image_conditional', x.t, tl['w'], logits.t, bias=tl['b'], K=3, pad=1, pro_act=L.GA_ACT_ELU)
        img = Act(self, R, D, D, IMG_LD, 'purified_nhwc')
        dm = L.DmlDesc()
        dm.logits, dm.ld, dm.nmix, dm.img_nchw, dm.img_nhwc = _ptr(logits.t), LO, nspec.num_mixtures, _ptr(self.purified), _ptr(img.t)
        dm.N, dm.H, dm.W, dm.backward, dm.ld_img = R, D, D, 0, IMG_LD
        self.fwd.add(dm, 'nd.dml_mean')
        self.dpurified = self.alloc((R * self.cot_rep, 3, D, D))

        def bwd_dml():
            b = L.DmlDesc()
            b.logits, b.ld, b.nmix, b.dimg_nhwc, b.dlogits = _ptr(logits.t), LO, nspec.num_mixtures, _ptr(img.g), _ptr(logits.g)
            b.dimg_nchw = _ptr(self.dpurified)
            b.N, b.H, b.W, b.backward, b.ld_img, b.act_rep = R * self.cot_rep, D, D, 1, IMG_LD, self.cot_rep
            self.bwd.add(b, 'nd.dml_mean^T')
            self.grad_conv('nd.image_conditional^T', logits.g, tl['w_bwd'], post_out, K=3, pad=1, dact_x=post_out.t, dact_act=L.GA_ACT_ELU)
        self._bwd_steps.append(bwd_dml)
        self._purified_grad_nhwc = img
        self._finish_with_classifier(csd, cspec, img)
        return self

    def _finish_with_classifier(self, csd, cspec, img: Act):
        """classifier behind a purifier that hands over `img` (NHWC, pitch IMG_LD, values in [0, 1]; self.purified holds the same
        image as NCHW): the VGG reads it directly; a ResNet / ResNeXt wants its input in space-to-depth form, produced from the NCHW
        copy by a second image boundary op (clamp to [0, 1] is the identity there) whose adjoint feeds ga_dml_mean's NCHW cotangent."""
        n_purifier_steps = len(self._bwd_steps)
        self.vspec = cspec
        if isinstance(cspec, ResNetSpec):
            if self.cot_rep != 1:
                raise NotImplementedError('K-cotangent plans are built for the VGG classifier')
            R, H = self.rows, img.h
            xs = Act(self, R, H // 2, H // 2, 4 * IMG_LD, 'purified_s2d')
            io = L.ImageIoDesc()
            io.x_nchw, io.y_nhwc, io.N, io.C, io.H, io.W, io.rep, io.backward, io.ld, io.s2d = _ptr(self.purified), _ptr(xs.t), R, 3, H, H, 1, 0, IMG_LD, 1
            self.fwd.add(io, 'purified_to_s2d')
            dpur_cls = self.alloc((R, 3, H, H))                # d loss / d purified coming from the classifier (NCHW)

            def bwd_io():
                b = L.ImageIoDesc()
                b.x_nchw, b.dy_nhwc, b.dx_nchw = _ptr(self.purified), _ptr(xs.g), _ptr(dpur_cls)
                b.N, b.C, b.H, b.W, b.rep, b.backward, b.ld, b.s2d = R, 3, H, H, 1, 1, IMG_LD, 1
                self.bwd.add(b, 'purified_to_s2d^T')
                a = L.AxpbyDesc()                               # into the NCHW cotangent ga_dml_mean^T reads beside the NHWC one
                a.x, a.y, a.n, a.alpha, a.beta = _ptr(dpur_cls), _ptr(self.dpurified), dpur_cls.numel(), 1.0, 1.0
                self.bwd.add(a, 'purified_to_s2d^T.add')
            self._bwd_steps.append(bwd_io)
            self.image_s2d = True
            self.logits = self._build_resnet(csd, xs)
            self.image_s2d = False
            self._purified_grad_nhwc = xs                      # BPDA starts from the classifier's own input cotangent
        else:
            self.logits = self._build_vgg(csd, img)
        self._finish(n_purifier_steps)
