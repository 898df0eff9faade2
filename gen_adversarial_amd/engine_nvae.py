"""NVAE purifier plans: ResidualCellEncoder / ResidualCellDecoder cells, latent groups, pre/post-processing
(NVAEDefenseModel.purify, src/defenses/ours/models.py:160-274).  Mixin of engine.Engine."""
from __future__ import annotations

from typing import Dict, Optional

import torch

from . import _lib as L
from . import folding as F
from .engine_core import IMG_LD, RES_SCALE, Act, _ptr
from .ndvae_spec import NdGenCell, NdResCell
from .nvae_spec import DecCellSpec, EncCellSpec


class NvaeBuilder:
    # ------------------------------------------------------------------------------------------------ cells
    def enc_cell(self, cell: EncCellSpec, x: Act) -> Act:
        """ResidualCellEncoder (architecture.py:96-136): fwd ops now, bwd ops registered for later.  Also the ND-VAE competitor's
        Residual_Cell_NVAE (competitors/nd_vae/modules/models/NVAE.py:255-297: the same chain with an unscaled residual and a
        2x2 / stride-2 skip = FactorizedReduce), told apart by the cell spec."""
        nd = isinstance(cell, NdResCell)
        wts = self.devd(('nd.' if nd else '') + cell.prefix, lambda: (F.fold_nd_res_cell if nd else F.fold_enc_cell)(self.nvae_sd, cell))
        rs = getattr(cell, 'res_scale', RES_SCALE)
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
        if cell.down:
            sk = Act(self, n, ho, wo, cell.cout, p + '.skip')
            ks = 2 if wts['ws'].shape[1] == 4 * x.c else 1          # SkipDown: 1x1 / 2; FactorizedReduce: 2x2 / 2 (one tap per quarter)
            self.conv(self.fwd, p + '.skip', x.t, wts['ws'], sk.t, bias=wts['bs'], K=ks, sn=2, pad=0, pro_act=L.GA_ACT_SILU)
            skip_t = sk.t
        else:
            skip_t = x.t
        if self.se_merges(t2, ho * wo):
            gate, hid = self.se_forward(p, t2, wts, ho * wo, res_scale=rs, merge=(skip_t, out.t))
        else:
            gate, hid = self.se_forward(p, t2, wts, ho * wo, res_scale=rs)
            a = L.SeApplyDesc()
            a.skip, a.t, a.gate, a.out = _ptr(skip_t), _ptr(t2.t), _ptr(gate), _ptr(out.t)
            a.N, a.H, a.W, a.C, a.skip_mode, a.res_scale = n, ho, wo, cell.cout, 0, rs
            self.fwd.add(a, p + '.merge')

        def backward():
            ps, pb = self.se_backward(p, out.g, t2, wts, gate, hid, ho * wo, res_scale=rs)
            dt1 = self.scratch((out.g.shape[0], ho, wo, cell.cout), 'enc_dt1')
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
        SkipUp 1x1 applied before its bilinear interpolation.  Also the ND-VAE competitor's Generative_Cell_NVAE
        (competitors/nd_vae/modules/models/NVAE.py:156-228): the same chain with one more 1x1 conv between the depthwise conv and
        its BatchNorm (`wp`) and an unscaled residual."""
        nd = isinstance(cell, NdGenCell)
        wts = self.devd(('nd.' if nd else '') + cell.prefix, lambda: (F.fold_nd_gen_cell if nd else F.fold_dec_cell)(self.nvae_sd, cell))
        rs = getattr(cell, 'res_scale', RES_SCALE)
        n, h, w = x.n, x.h, x.w
        up = cell.up
        H, W = (2 * h, 2 * w) if up else (h, w)
        hid_c = cell.hidden
        p = cell.prefix
        # whole-image tiles at 128 / 256 channels: ONE launch per direction, the two hid_c-wide tensors never reach HBM
        fused = (not up and not nd and self.precision == 'bf16x3' and self.fuse_dec_cells and cell.cout == x.c
                 and L.lib.ga_dec_cell_supported(n, H, W, x.c, hid_c) == 1
                 and n * H * W // (256 if x.c == 128 else 128) >= self.fuse_min_workgroups)
        # few-channel cells on images larger than a workgroup (post-processing): 8 x 16 tiles with a recomputed halo, and d x from
        # the same launch (ga_dec_cell_halo); one cotangent per forward row only
        # ... selected in FORWARD-ONLY plans (need_backward=False: clean predictions, get_purified, the alpha objective — forward 0.63 vs
        # 0.95 ms and 0.32 vs 0.57 ms per 512-row chunk, same bits), and with GA_FUSE_HALO_CELL=1 everywhere (its backward is slower)
        halo = (not fused and not up and not nd and self.precision == 'bf16x3' and self.fuse_dec_cells
                and (self.fuse_halo_cells or not self.need_backward) and cell.cout == x.c
                and self.cot_rep == 1 and L.lib.ga_dec_cell_halo_supported(n, H, W, x.c, hid_c) == 1
                and n * H * W // 128 >= self.fuse_min_workgroups)
        t3 = Act(self, n, H, W, cell.cout, p + '.t3')
        out = Act(self, n, H, W, cell.cout, p + '.out')

        def halo_desc(backward: int) -> L.DecCellHaloDesc:
            f = L.DecCellHaloDesc()
            f.x, f.b1, f.wd, f.wd_bwd, f.bd, f.b2 = (_ptr(x.t), _ptr(wts['b1']), _ptr(wts['wd']), _ptr(wts['wd_bwd']),
                                                    _ptr(wts['bd']), _ptr(wts['b2']))
            if not self.dry_run:
                (h1, l1), (h2, l2) = self.store.split(wts['w1']), self.store.split(wts['w2_bwd' if backward else 'w2'])
                f.w1_hi, f.w1_lo, f.w2_hi, f.w2_lo = _ptr(h1), _ptr(l1), _ptr(h2), _ptr(l2)
                if backward:
                    h3, l3 = self.store.split(wts['w1_bwd'])
                    f.w1t_hi, f.w1t_lo = _ptr(h3), _ptr(l3)
            f.N, f.H, f.W, f.Cin, f.Cout, f.Hd, f.backward, f.up = n, H, W, x.c, cell.cout, hid_c, backward, 0
            return f

        def fused_desc(backward: int) -> L.DecCellDesc:
            f = L.DecCellDesc()
            f.x, f.b1, f.wd, f.wd_bwd, f.bd, f.b2 = (_ptr(x.t), _ptr(wts['b1']), _ptr(wts['wd']), _ptr(wts['wd_bwd']),
                                                    _ptr(wts['bd']), _ptr(wts['b2']))
            if not self.dry_run:
                (h1, l1), (h2, l2) = self.store.split(wts['w1']), self.store.split(wts['w2_bwd' if backward else 'w2'])
                f.w1_hi, f.w1_lo, f.w2_hi, f.w2_lo = _ptr(h1), _ptr(l1), _ptr(h2), _ptr(l2)
            f.N, f.H, f.W, f.C, f.Hd, f.backward = n, H, W, x.c, hid_c, backward
            f.variant = self.dec_cell_variant if x.c == 128 else 0
            return f

        if fused:
            f = fused_desc(0)
            f.y = _ptr(t3.t)
            self.fwd.add(f, p + '.cell')
        elif halo:
            f = halo_desc(0)
            f.y = _ptr(t3.t)
            self.fwd.add(f, p + '.cell')
        else:
            t1 = Act(self, n, h, w, hid_c, p + '.t1')
            t2 = Act(self, n, H, W, hid_c, p + '.t2')
            self.conv(self.fwd, p + '.pw1', x.t, wts['w1'], t1.t, bias=wts['b1'], K=1)
            d = L.DwDesc()
            d.x, d.w, d.bias, d.y = _ptr(t1.t), _ptr(wts['wd']), _ptr(wts['bd']), _ptr(t2.t)
            d.N, d.H, d.W, d.C, d.pro_act, d.up2 = n, H, W, hid_c, L.GA_ACT_SILU, int(up)
            self.fwd.add(d, p + '.dw5')
            if nd:          # depthwise_separable_conv: depthwise (above, bias only) then pointwise with the BatchNorm behind it folded in
                u = t2
                t2 = Act(self, n, H, W, hid_c, p + '.t2p')
                self.conv(self.fwd, p + '.pw', u.t, wts['wp'], t2.t, bias=wts['bp'], K=1)
            self.conv(self.fwd, p + '.pw2', t2.t, wts['w2'], t3.t, bias=wts['b2'], K=1, pro_act=L.GA_ACT_SILU)
        if not up and self.se_merges(t3, H * W):
            gate, hid = self.se_forward(p, t3, wts, H * W, res_scale=rs, merge=(x.t, out.t))
        else:
            a = L.SeApplyDesc()
            if up:
                sl = Act(self, n, h, w, cell.cout, p + '.skip_low')
                self.conv(self.fwd, p + '.skip', x.t, wts['ws'], sl.t, bias=wts['bs'], K=1)
                a.skip, a.skip_mode = _ptr(sl.t), 1
            else:
                a.skip, a.skip_mode = _ptr(x.t), 0
            gate, hid = self.se_forward(p, t3, wts, H * W, res_scale=rs)
            a.t, a.gate, a.out = _ptr(t3.t), _ptr(gate), _ptr(out.t)
            a.N, a.H, a.W, a.C, a.res_scale = n, H, W, cell.cout, rs
            self.fwd.add(a, p + '.merge')

        def backward():
            nc = out.g.shape[0]                                 # cotangent rows: n * cot_rep
            ps, pb = self.se_backward(p, out.g, t3, wts, gate, hid, H * W, res_scale=rs)
            if halo:            # d x = out.g (identity skip) [+ what is already in x.g] + W1^T dt1, dt1 never stored
                b = halo_desc(1)
                b.dout, b.pro_scale, b.pro_shift, b.addend, b.y = _ptr(out.g), _ptr(ps), _ptr(pb), _ptr(out.g), _ptr(x.g)
                if x.g_written:
                    b.addend2 = _ptr(x.g)
                self.bwd.add(b, p + '.cell^T')
                x.g_written = True
                return
            dt1 = self.scratch((nc, h, w, hid_c), 'dec_dt1')
            if fused:
                b = fused_desc(1)
                b.dout, b.pro_scale, b.pro_shift, b.y = _ptr(out.g), _ptr(ps), _ptr(pb), _ptr(dt1)
                b.N, b.act_rep = nc, self.cot_rep
                self.bwd.add(b, p + '.cell^T')
            else:
                dt2 = self.scratch((nc, H, W, hid_c), 'dec_dt2')
                self.conv(self.bwd, p + '.pw2^T', out.g, wts['w2_bwd'], dt2, K=1,
                          pro_scale=ps, pro_shift=pb, pro_per_row=1, dact_x=t2.t, dact_act=L.GA_ACT_SILU)
                if nd:      # the pointwise conv's transpose (linear: no act' between it and the depthwise conv)
                    du = self.scratch((nc, H, W, hid_c), 'dec_du')
                    self.conv(self.bwd, p + '.pw^T', dt2, wts['wp_bwd'], du, K=1)
                    dt2 = du
                b = L.DwDesc()
                b.x, b.w, b.dact_x, b.y = _ptr(dt2), _ptr(wts['wd_bwd']), _ptr(t1.t), _ptr(dt1)
                b.N, b.H, b.W, b.C, b.dact_act, b.pool2, b.act_rep = nc, H, W, hid_c, L.GA_ACT_SILU, int(up), self.cot_rep
                self.bwd.add(b, p + '.dw5^T')
            self.grad_conv(p + '.pw1^T', dt1, wts['w1_bwd'], x, K=1, primary=None if up else out.g)
            if up:
                dsl = self.scratch((nc, h, w, cell.cout), 'dec_dsl')
                bl = L.BilinearBwdDesc()
                bl.dhigh, bl.dlow, bl.N, bl.h, bl.w, bl.C, bl.accumulate = _ptr(out.g), _ptr(dsl), nc, h, w, cell.cout, 0
                self.bwd.add(bl, p + '.bilinear^T')
                self.grad_conv(p + '.skip^T', dsl, wts['ws_bwd'], x, K=1)
        self._bwd_steps.append(backward)
        return out

    def _build_nvae(self, x0: Act) -> Act:
        """NVAEDefenseModel.purify (models.py:160-274) on the NHWC image x0; returns the purified NHWC image."""
        nvae_sd = self.nvae_sd
        spec, R = self.spec, self.rows
        H = spec.resolution
        NL = spec.num_latent
        self.latent_pitch = (NL + 7) // 8 * 8
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
        # latent tensors (mu_q, z and their cotangents) carry NLP = NL rounded up to 8 channels, the extra ones exact zeros meeting
        # zero weights: a conv with 20 input channels runs on the exact-fp32 kernel, with 24 on the split-bf16 ones
        NLP = self.latent_pitch
        s00 = self.devd('enc_sampler_0:0', lambda: F.pad_conv_out(
            F.fold_wn_conv(nvae_sd, 'enc_sampler.sampler_0:0', out_slice=slice(0, NL)), NL, NLP))
        muq0 = Act(self, RE, g0.res, g0.res, NLP, 'mu_q0')
        self.conv(self.fwd, 'enc_sampler_0:0', e0.t, s00['w'], muq0.t, bias=s00['b'], K=3, pad=1, pro_act=L.GA_ACT_ELU)
        z = Act(self, R, g0.res, g0.res, NLP, 'z0')
        self._sampler_fwd('sample_0:0', muq0, None, self.eps[0], z, self.alphas[0], q_rep=erep)

        # ---- combiner_0:0 on cat[const_prior, z0]: the prior half is row-independent -> folded into a broadcast addend
        def fold_comb0():
            wfull = F.wn_weight64(nvae_sd, 'decoder_combiners.combiner_0:0.conv')[:, :, 0, 0]   # [C0, C0+NL]
            prior = nvae_sd['const_prior'].double()[0]                                           # [C0,h,w]
            pc = torch.einsum('oc,chw->hwo', wfull[:, :C0], prior) + \
                nvae_sd['decoder_combiners.combiner_0:0.conv.bias'].double()
            if spec.num_nf_cells:            # flow of this group = z - c (folding.nf_constant_shift): fold W_z c into the addend
                pc = pc - wfull[:, C0:] @ F.nf_constant_shift(nvae_sd, '0:0', spec.num_nf_cells, NL)
            return {'pc': pc.float().unsqueeze(0), 'wz': F.pad_cols(wfull[:, C0:].float(), 0, NL, NLP),
                    'wz_bwd': F.pad_rows(wfull[:, C0:].t().float().contiguous(), NLP)}
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
        LO = (spec.logits_out + 7) // 8 * 8          # channel pitch of the mixture logits: the transposed conv then takes the bf16 path
        tl = self.devd('to_logits', lambda: F.pad_conv_out(F.fold_wn_conv(nvae_sd, 'to_logits.1'), spec.logits_out, LO))
        logits = Act(self, R, H, H, LO, 'mix_logits')
        post_out = x
        self.conv(self.fwd, 'to_logits', x.t, tl['w'], logits.t, bias=tl['b'], K=3, pad=1, pro_act=L.GA_ACT_ELU)
        img = Act(self, R, H, H, IMG_LD, 'purified_nhwc')
        dm = L.DmlDesc()
        dm.logits, dm.ld, dm.nmix, dm.img_nchw, dm.img_nhwc = _ptr(logits.t), LO, spec.num_mixtures, _ptr(self.purified), _ptr(img.t)
        dm.N, dm.H, dm.W, dm.backward, dm.ld_img = R, H, H, 0, IMG_LD
        self.fwd.add(dm, 'dml_mean')
        self.dpurified = self.alloc((R * self.cot_rep, 3, H, H))    # optional external gradient on the purified image (NCHW)
        purified_img = img

        def bwd_dml():
            b = L.DmlDesc()
            b.logits, b.ld, b.nmix, b.dimg_nhwc, b.dlogits = _ptr(logits.t), LO, spec.num_mixtures, _ptr(img.g), _ptr(logits.g)
            b.dimg_nchw = _ptr(self.dpurified)
            b.N, b.H, b.W, b.backward, b.ld_img, b.act_rep = R * self.cot_rep, H, H, 1, IMG_LD, self.cot_rep
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
        d.N, d.h, d.w, d.NL, d.ldz = z.n, z.h, z.w, self.spec.num_latent, z.c
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
        nc = z.n * self.cot_rep                                 # cotangent rows
        if q_rep > 1:
            rows_grad = self.scratch((nc, z.h, z.w, z.c), 'dmu_q_rows')
            d.dmu_q_rows = _ptr(rows_grad)
        else:
            d.dmu_q = _ptr(muq.g)
        d.N, d.h, d.w, d.NL, d.ldz, d.act_rep = nc, z.h, z.w, self.spec.num_latent, z.c, self.cot_rep
        d.alpha, d.one_minus_alpha, d.temp, d.backward = alpha, 1.0 - alpha, self.temperature, 1
        self._sampler_descs.append((d, [i for i, e in enumerate(self.eps) if e is eps][0]))
        self.bwd.add(d, name)
        if q_rep > 1:
            self.rep_sum(name + '.rep_sum', rows_grad, muq, q_rep)
        muq.g_written = True

    def rep_sum(self, name, x_rows: torch.Tensor, target: Act, rep: int):
        """target.g (+)= sum over the `rep` replicas of x_rows (gradient of a tensor shared by the EoT replicas)"""
        r = L.RepSumDesc()
        # x_rows is [forward rows][K cotangents][...]: summing the `rep` replicas of an image keeps (image, k) apart when the K
        # cotangents count as part of the row
        K = self.cot_rep
        r.x, r.y, r.rows, r.inner, r.rep = _ptr(x_rows), _ptr(target.g), x_rows.shape[0] // K, x_rows[0].numel() * K, rep
        r.accumulate = int(target.g_written)
        self.bwd.add(r, name)
        target.g_written = True

    def _latent_group(self, gs, x: Act, enc_feat: Act, enc_rep: int = 1) -> Act:
        """models.py:236-257 for one latent group: encoder/decoder parameters, interpolation, combiner."""
        sd, R, NL, C, r = self.nvae_sd, self.rows, self.spec.num_latent, gs.channels, gs.res
        key = f'{gs.s}:{gs.g}'
        ec_w = self.devd(f'enc_combiner_{key}', lambda: F.fold_wn_conv(sd, f'encoder_combiners.combiner_{key}.conv'))
        NLP = self.latent_pitch
        es_w = self.devd(f'enc_sampler_{key}', lambda: F.pad_conv_out(
            F.fold_wn_conv(sd, f'enc_sampler.sampler_{key}', out_slice=slice(0, NL)), NL, NLP))
        ds_w = self.devd(f'dec_sampler_{key}', lambda: F.fold_wn_conv(sd, f'dec_sampler.sampler_{key}.1'))

        def fold_comb():
            cb = F.wn_weight64(sd, f'decoder_combiners.combiner_{key}.conv')[:, :, 0, 0]         # [C, C+NL]
            bias = sd[f'decoder_combiners.combiner_{key}.conv.bias'].double()
            if self.spec.num_nf_cells:       # the group's flow is z - c: combiner(cat[x, z - c]) = ... - W_z c
                bias = bias - cb[:, C:] @ F.nf_constant_shift(sd, key, self.spec.num_nf_cells, NL)
            return {'w': F.pad_cols(cb.float(), C, NL, NLP), 'b': bias.float(),
                    'x_bwd': cb[:, :C].t().float(), 'z_bwd': F.pad_rows(cb[:, C:].t().float().contiguous(), NLP)}
        cbw = self.devd(f'combiner_{key}', fold_comb)
        cb_w, cb_b, cbx_bwd, cbz_bwd = cbw['w'], cbw['b'], cbw['x_bwd'], cbw['z_bwd']
        alpha = self.alphas[gs.latent_idx]
        eps = self.eps[gs.latent_idx]

        ec = Act(self, R, r, r, C, f'ec_{key}')
        # ec = conv(x) + enc_feat: d(enc_feat) = d(ec).  Without a shared encoder the two cotangents have the same shape, and the
        # decoder side runs first in the backward plan, so enc_feat's gradient buffer IS d(ec): no copy launch (r04: 23 axpby
        # launches per chunk less); the encoder cell that produced enc_feat later accumulates into it as into any written gradient
        alias_grad = enc_rep == 1 and self.need_backward and not enc_feat.g_written and enc_feat._g is None
        if alias_grad:
            ec._g = enc_feat.g
        d_ec = self.conv(self.fwd, f'enc_combiner_{key}', x.t, ec_w['w'], ec.t, bias=ec_w['b'], K=1, addend=enc_feat.t)
        d_ec.addend_rep = enc_rep
        muq = Act(self, R, r, r, NLP, f'mu_q_{key}')
        self.conv(self.fwd, f'enc_sampler_{key}', ec.t, es_w['w'], muq.t, bias=es_w['b'], K=3, pad=1)
        pp = Act(self, R, r, r, 2 * NL, f'p_{key}')
        self.conv(self.fwd, f'dec_sampler_{key}', x.t, ds_w['w'], pp.t, bias=ds_w['b'], K=1, pro_act=L.GA_ACT_ELU)
        z = Act(self, R, r, r, NLP, f'z_{key}')
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
            if alias_grad:          # d(ec) was written straight into enc_feat's gradient buffer
                assert not enc_feat.g_written
                enc_feat.g_written = True
                return
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

