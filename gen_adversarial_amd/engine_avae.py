"""A-VAE competitor defender plans (`AVaeDefenseModel` over `StyledGenerator`):
src/defenses/competitors/a_vae/purification_model.py:16-25, src/defenses/competitors/a_vae/model.py:9-141, modules.py.
Mixin of engine.Engine.  Layout of the chain (all tensors NHWC, stored PRE-activation; LeakyReLU(0.2) is the PReLU prologue of
the consumer, its derivative the PReLU epilogue of the backward GEMM, like the e4e encoder's units):

  image_io -> avg-pool k (ga_avae) -> [2x-1 as prologue] EncodeConvBlock x 3 -> sample (ga_avae) -> PixelNorm -> style MLP
  ConstantInput -> { [nearest x2 -> conv3x3 | transposed 4x4 / 2 conv] -> Blur } -> noise + LeakyReLU + AdaIN (ga_avae, one pass)
               -> conv3x3 -> noise + LeakyReLU + AdaIN -> ... -> to_rgb -> (x + 1) / 2 -> classifier
"""
from __future__ import annotations

from math import sqrt
from typing import List

import torch

from . import _lib as L
from . import folding as F
from .avae_spec import TEMP_INFERENCE, AvaeSpec
from .engine_core import IMG_LD, Act, _ptr
from .resnet_spec import ResNetSpec


def _eq_conv(sd, prefix: str, prescale: float = 1.0) -> dict:
    """EqualConv2d (modules.py:159-169): weight_orig * sqrt(2 / fan_in) folded once"""
    w = sd[f'{prefix}.conv.weight_orig'].double()
    w = w * sqrt(2.0 / (w.shape[1] * w.shape[2] * w.shape[3])) * prescale
    return {'w': F.f32(F.conv_fwd_layout(w)), 'w_bwd': F.f32(F.conv_bwd_layout(w)), 'b': F.f32(sd[f'{prefix}.conv.bias'].double() * prescale),
            'w4': w}


def _eq_linear(sd, prefix: str) -> dict:
    w = sd[f'{prefix}.linear.weight_orig'].double()
    w = w * sqrt(2.0 / w.shape[1])
    return {'w': F.f32(w), 'w_bwd': F.f32(w.t()), 'b': F.f32(sd[f'{prefix}.linear.bias'].double())}


class AvaeBuilder:
    def _avae_op(self, plan, name, mode, **kw) -> L.AvaeDesc:
        d = L.AvaeDesc()
        d.mode = mode
        for k, v in kw.items():
            setattr(d, k, _ptr(v) if torch.is_tensor(v) else v)
        plan.add(d, name)
        return d

    def _lrelu_slopes(self, c: int) -> torch.Tensor:
        return self.devd(f'avae.slope{c}', lambda: {'s': torch.full((c,), 0.2)})['s']

    def _dw_fixed(self, plan, name, x: torch.Tensor, y: torch.Tensor, taps: torch.Tensor, up2=0, pool2=0):
        d = L.DwDesc()
        n, h, w, c = (y.shape if not pool2 else x.shape)
        d.x, d.w, d.y = _ptr(x), _ptr(taps), _ptr(y)
        d.N, d.H, d.W, d.C, d.pro_act, d.up2, d.pool2 = n, h, w, c, L.GA_ACT_NONE, up2, pool2
        plan.add(d, name)

    def _avae_adain(self, name: str, t: Act, noise: torch.Tensor, wn: torch.Tensor, style_pre: Act, fc: dict) -> Act:
        """NoiseInjection -> LeakyReLU -> AdaptiveInstanceNorm (modules.py:372-375): gamma | beta = EqualLinear(style); the style
        is kept pre-activation (its LeakyReLU is this FC's PReLU prologue)"""
        R, C, S = t.n, t.c, style_pre.c
        sl = self._lrelu_slopes(S)
        gb = Act(self, R, 1, 1, 2 * C, name + '.style')
        self.conv(self.fwd, name + '.style', style_pre.t, fc['w'], gb.t, bias=fc['b'], K=1, pro_scale=sl, pro_shift=sl, flags=L.GA_CONV_PRO_PRELU)
        y = Act(self, R, t.h, t.w, C, name)
        stats = self.alloc((R, C, 2))
        if not hasattr(self, '_avae_kinks'):
            self._avae_kinks = []
        self._avae_kinks.append((t, noise, wn))            # the LeakyReLU decides on t + wn * noise (parity tests replay it)
        self._avae_op(self.fwd, name, L.GA_AVAE_ADAIN, x=t.t, a=noise, b=wn, c=gb.t, y=y.t, y2=stats, N=R, P=t.h * t.w, C=C, backward=0)

        def backward():
            self._avae_op(self.bwd, name + '^T', L.GA_AVAE_ADAIN, x=t.t, a=noise, b=wn, c=gb.t, s=stats, dy=y.g, y=t.g, y2=gb.g,
                          N=R, P=t.h * t.w, C=C, backward=1)
            t.g_written = gb.g_written = True
            self.grad_conv(name + '.style^T', gb.g, fc['w_bwd'], style_pre, K=1, dact_x=style_pre.t, dact_scale=sl, dact_shift=sl,
                           flags=L.GA_CONV_DACT_PRELU)
        self._bwd_steps.append(backward)
        return y

    def extra_kink_tensors(self) -> List[torch.Tensor]:
        """the tensors whose signs the generator's LeakyReLUs decided on in the last forward ([R, H, W, C] each): the engine stores
        the conv outputs BEFORE the noise injection, the decision is taken after it (tests/gradcheck.py replays it in the oracle)"""
        return [t.t + wn.view(1, 1, 1, -1) * nz.view(t.n, t.h, t.w, 1) for t, nz, wn in getattr(self, '_avae_kinks', [])]

    def build_avae_defense(self, asd, aspec: AvaeSpec, kernel_size: int, csd, cspec):
        """AVaeDefenseModel.forward (purification_model.py:22-25) as one forward / backward plan pair.  Caller-visible: x_in, eps =
        [latent draw [R, C, 4, 4] (NCHW like the reference), noise image of block i [R, 1, s_i, s_i] for every generator block],
        logits / dlogits, dx; the purified image is `purified_nhwc` (NHWC, pitch IMG_LD; not clamped, like the reference's)."""
        R, D = self.rows, aspec.output_size
        assert self.resolution[1] == D and D % kernel_size == 0 and D // kernel_size == aspec.enc_res, (self.resolution, D, kernel_size)
        assert self.cot_rep == 1, 'K-cotangent plans are built for the NVAE + VGG defender'
        self.nvae_sd = asd
        self.image_s2d = False
        self.share_encoder, self.enc_rows = False, R
        x0 = self._build_input()
        E, Sd = aspec.c512, aspec.style_dim
        self.eps = [self.alloc((R, E, 4, 4))] + [self.alloc((R, 1, b.res, b.res)) for b in aspec.blocks]
        noise = self.eps[1:]

        # ---- avg_pool2d(x * 2 - 1, k): the pool here, the affine as the first conv's prologue (the two commute)
        er = aspec.enc_res
        xp = Act(self, R, er, er, IMG_LD, 'avae.pooled')
        self._avae_op(self.fwd, 'avae.avgpool', L.GA_AVAE_AVGPOOL, x=x0.t, y=xp.t, N=R, H=D, W=D, C=IMG_LD, k=kernel_size, backward=0)

        def bwd_pool():
            self._avae_op(self.bwd, 'avae.avgpool^T', L.GA_AVAE_AVGPOOL, x=x0.t, dy=xp.g, y=x0.g, N=R, H=D, W=D, C=IMG_LD, k=kernel_size, backward=1)
            x0.g_written = True
        self._bwd_steps.append(bwd_pool)

        # ---- Encoder (model.py:20-27): three EncodeConvBlocks, conv3x3 -> LeakyReLU -> conv3x3 / 2 -> LeakyReLU (no normalisation)
        norm = self.devd('norm05', lambda: {'two': torch.full((IMG_LD,), 2.0), 'mone': torch.full((IMG_LD,), -1.0)})
        two, mone = norm['two'], norm['mone']
        cur, pre = xp, None              # pre: slopes of the LeakyReLU in front of `cur` (None: the image affine)
        enc_out: List[Act] = []
        for name, cout in (('conv2', E // 2), ('conv3', E), ('conv4', 2 * E)):
            p = f'encoder.{name}'
            first = pre is None
            w1 = self.devd('avae.' + p + '.conv1', lambda p=p, first=first: (F.pad_image_conv(_eq_conv(asd, p + '.conv1'), 3, IMG_LD) if first
                                                                            else _eq_conv(asd, p + '.conv1')))
            w2 = self.devd('avae.' + p + '.conv2', lambda p=p: dict(_eq_conv(asd, p + '.conv2'), **{
                f'sub{a}{b}': wm for (a, b), (wm, kh, kw) in F.subpixel_weights(_eq_conv(asd, p + '.conv2')['w4']).items()}))
            ta = Act(self, R, cur.h, cur.w, cout, p + '.t1')
            tb = Act(self, R, cur.h // 2, cur.w // 2, cout, p + '.t2')
            sl = self._lrelu_slopes(cout)
            if first:
                self.conv(self.fwd, p + '.conv1', cur.t, w1['w'], ta.t, bias=w1['b'], K=3, pad=1, pro_scale=two, pro_shift=mone)
            else:
                self.conv(self.fwd, p + '.conv1', cur.t, w1['w'], ta.t, bias=w1['b'], K=3, pad=1, pro_scale=pre, pro_shift=pre,
                          flags=L.GA_CONV_PRO_PRELU)
            self.conv(self.fwd, p + '.conv2', ta.t, w2['w'], tb.t, bias=w2['b'], K=3, sn=2, pad=1, pro_scale=sl, pro_shift=sl,
                      flags=L.GA_CONV_PRO_PRELU)

            def bwd_block(p=p, cur=cur, pre=pre, first=first, w1=w1, w2=w2, ta=ta, tb=tb, sl=sl):
                self.grad_conv_up2(p + '.conv2^T', tb.g, w2, 'sub', ta, dact_x=ta.t, dact_scale=sl, dact_shift=sl, dact_prelu=True)
                if first:
                    self.grad_conv(p + '.conv1^T', ta.g, w1['w_bwd'], cur, K=3, pad=1, dact_x=cur.t, dact_scale=two, dact_shift=mone,
                                   dact_act=L.GA_ACT_NONE)
                else:
                    self.grad_conv(p + '.conv1^T', ta.g, w1['w_bwd'], cur, K=3, pad=1, dact_x=cur.t, dact_scale=pre, dact_shift=pre,
                                   flags=L.GA_CONV_DACT_PRELU)
            self._bwd_steps.append(bwd_block)
            cur, pre = tb, sl
            enc_out.append(tb)
        x_skip, t4 = enc_out[0], enc_out[2]                 # pre-activation tensors: LeakyReLU is applied by their consumers

        # ---- sample (model.py:80-83) and the style MLP (model.py:116-125)
        z = Act(self, R, 1, 1, 16 * E, 'avae.z')
        self._avae_op(self.fwd, 'avae.sample', L.GA_AVAE_SAMPLE, x=t4.t, a=self.eps[0], y=z.t, N=R, P=16, C=E, f0=TEMP_INFERENCE, backward=0)
        zn = Act(self, R, 1, 1, 16 * E, 'avae.zn')
        self._avae_op(self.fwd, 'avae.pixelnorm', L.GA_AVAE_PIXELNORM, x=z.t, y=zn.t, N=R, C=16 * E, backward=0)

        def fold_fc1():           # z is flattened channel-major by the reference (out.view(B, -1) of NCHW), pixel-major here
            f = _eq_linear(asd, 'style.1')
            w = f['w'].view(Sd, E, 16).permute(0, 2, 1).reshape(Sd, 16 * E).contiguous()
            return {'w': w, 'w_bwd': w.t().contiguous(), 'b': f['b']}
        fcs = [self.devd('avae.style.1', fold_fc1)] + [self.devd(f'avae.style.{3 + 2 * i}', lambda i=i: _eq_linear(asd, f'style.{3 + 2 * i}'))
                                                      for i in range(aspec.n_mlp)]
        sls = self._lrelu_slopes(Sd)
        hs: List[Act] = []
        src = zn
        for i, fc in enumerate(fcs):
            h = Act(self, R, 1, 1, Sd, f'avae.style.h{i}')
            if i == 0:
                self.conv(self.fwd, h.name, src.t, fc['w'], h.t, bias=fc['b'], K=1)
            else:
                self.conv(self.fwd, h.name, src.t, fc['w'], h.t, bias=fc['b'], K=1, pro_scale=sls, pro_shift=sls, flags=L.GA_CONV_PRO_PRELU)
            hs.append(h)
            src = h
        style_pre = hs[-1]

        def bwd_mlp():
            for i in range(len(fcs) - 1, 0, -1):
                self.grad_conv(f'avae.style.h{i}^T', hs[i].g, fcs[i]['w_bwd'], hs[i - 1], K=1, dact_x=hs[i - 1].t, dact_scale=sls,
                               dact_shift=sls, flags=L.GA_CONV_DACT_PRELU)
            self.grad_conv('avae.style.h0^T', hs[0].g, fcs[0]['w_bwd'], zn, K=1)
            self._avae_op(self.bwd, 'avae.pixelnorm^T', L.GA_AVAE_PIXELNORM, x=z.t, dy=zn.g, y=z.g, N=R, C=16 * E, backward=1)
            # d t4: the sample's adjoint WRITES the whole cotangent of the encoder's last conv output (its only consumer)
            self._avae_op(self.bwd, 'avae.sample^T', L.GA_AVAE_SAMPLE, x=t4.t, a=self.eps[0], dy=z.g, y=t4.g, N=R, P=16, C=E,
                          f0=TEMP_INFERENCE, backward=1)
            t4.g_written = True
        self._bwd_steps.append(bwd_mlp)

        # ---- Generator (model.py:73-105)
        def dw_taps(c: int) -> dict:
            """depthwise 5x5 tap tables [25][c] of ga_dwconv5: the identity (nearest x2 through its `up2` read) and Blur's
            [1, 2, 1] x [1, 2, 1] / 16 (modules.py:142-156; pad 1 = the 3 x 3 kernel in the middle of the 5 x 5 frame)"""
            k5 = torch.tensor([[0., 0, 0, 0, 0], [0, 1, 2, 1, 0], [0, 2, 4, 2, 0], [0, 1, 2, 1, 0], [0, 0, 0, 0, 0]]) / 16.0
            return self.devd(f'avae.dwtaps{c}', lambda: {'id': F.f32(torch.zeros(25, c).index_fill_(0, torch.tensor([12]), 1.0)),
                                                        'blur': F.f32(k5.reshape(25, 1).repeat(1, c))})
        out = None
        for b in aspec.blocks:
            p, C = f'generator.progression.{b.idx}', b.cout
            taps = dw_taps(C)
            if b.kind == 'initial':
                ta = Act(self, R, 4, 4, C, p + '.const')
                if not self.dry_run:
                    ta.t.copy_(asd[f'{p}.conv1.input'].float().permute(0, 2, 3, 1).to(ta.t.device).expand(R, -1, -1, -1))
            else:
                h, w = out.h, out.w
                pre_blur = Act(self, R, 2 * h, 2 * w, C, p + '.conv1')
                ta = Act(self, R, 2 * h, 2 * w, C, p + '.blur')
                if b.kind == 'up':          # nn.Upsample(nearest) -> EqualConv2d 3x3 -> Blur (modules.py:345-351)
                    wc = self.devd('avae.' + p + '.conv1', lambda p=p: _eq_conv(asd, p + '.conv1.1'))
                    tid = dw_taps(out.c)['id']
                    up = Act(self, R, 2 * h, 2 * w, out.c, p + '.up')
                    self._dw_fixed(self.fwd, p + '.nearest_up', out.t, up.t, tid, up2=1)
                    self.conv(self.fwd, p + '.conv1', up.t, wc['w'], pre_blur.t, bias=wc['b'], K=3, pad=1)

                    def bwd_conv1(p=p, wc=wc, up=up, src=out, pre_blur=pre_blur, tid=tid):
                        self.grad_conv(p + '.conv1^T', pre_blur.g, wc['w_bwd'], up, K=3, pad=1)
                        assert not src.g_written
                        self._dw_fixed(self.bwd, p + '.nearest_up^T', up.g, src.g, tid, pool2=1)
                        src.g_written = True
                else:                       # FusedUpsample (modules.py:38-65): transposed conv with the 2x2-averaged 4x4 kernel, stride 2
                    def fold_fused(p=p):
                        w = asd[f'{p}.conv1.0.weight'].double()                               # [in, out, 3, 3]
                        w = torch.nn.functional.pad(w * sqrt(2.0 / (w.shape[0] * 9)), [1, 1, 1, 1])
                        w = (w[:, :, 1:, 1:] + w[:, :, :-1, 1:] + w[:, :, 1:, :-1] + w[:, :, :-1, :-1]) / 4      # [in, out, 4, 4]
                        # as a ga_conv2d with sd = 2 (hi' = ho - pad + kh must be 2 hi): flipped taps, pad = K - 1 - 1 = 2
                        wf = w.flip(2, 3).permute(1, 2, 3, 0).reshape(w.shape[1], -1)            # [out][(kh, kw, in)]
                        # its adjoint: the stride-2 conv dx[hi] = sum dy[2 hi - 1 + kh] w[in, out, kh, kw]
                        wb = w.permute(0, 2, 3, 1).reshape(w.shape[0], -1)                       # [in][(kh, kw, out)]
                        return {'wf': F.f32(wf), 'wb': F.f32(wb), 'b': F.f32(asd[f'{p}.conv1.0.bias'].double())}
                    wc = self.devd('avae.' + p + '.conv1', fold_fused)
                    ca = out.c                                                       # generator channels; the rest is the encoder skip
                    wfa = wc['wf'].view(C, 16, b.cin)[:, :, :ca].reshape(C, -1).contiguous()
                    wba = wc['wb'][:ca].contiguous()
                    self._keep += [wfa, wba]
                    self.conv(self.fwd, p + '.conv1', out.t, wfa, pre_blur.t, bias=wc['b'], K=4, sn=1, sd=2, pad=2)
                    if b.skip:
                        cs = x_skip.c
                        assert ca + cs == b.cin and (x_skip.h, x_skip.w) == (h, w), (b, x_skip.h, h)
                        wfs = wc['wf'].view(C, 16, b.cin)[:, :, ca:].reshape(C, -1).contiguous()
                        wbs = wc['wb'][ca:].contiguous()
                        self._keep += [wfs, wbs]
                        slk = self._lrelu_slopes(cs)
                        self.conv(self.fwd, p + '.conv1.skip', x_skip.t, wfs, pre_blur.t, K=4, sn=1, sd=2, pad=2, addend=pre_blur.t,
                                  pro_scale=slk, pro_shift=slk, flags=L.GA_CONV_PRO_PRELU)

                    def bwd_conv1(p=p, src=out, pre_blur=pre_blur, wba=wba, skip=b.skip, wbs=(wbs if b.skip else None),
                                  slk=(slk if b.skip else None)):
                        self.grad_conv(p + '.conv1^T', pre_blur.g, wba, src, K=4, sn=2, pad=1)
                        if skip:
                            self.grad_conv(p + '.conv1.skip^T', pre_blur.g, wbs, x_skip, K=4, sn=2, pad=1, dact_x=x_skip.t, dact_scale=slk,
                                           dact_shift=slk, flags=L.GA_CONV_DACT_PRELU)
                self._dw_fixed(self.fwd, p + '.blur', pre_blur.t, ta.t, taps['blur'])

                def bwd_blur(p=p, pre_blur=pre_blur, ta=ta, taps=taps):
                    self._dw_fixed(self.bwd, p + '.blur^T', ta.g, pre_blur.g, taps['blur'])
                    pre_blur.g_written = True
                self._bwd_steps.append(bwd_conv1)
                self._bwd_steps.append(bwd_blur)
            wn = self.devd('avae.' + p + '.noise', lambda p=p, C=C: {
                'n1': F.f32(asd[f'{p}.noise1.weight_orig'].double().view(C) * sqrt(2.0 / C)),
                'n2': F.f32(asd[f'{p}.noise2.weight_orig'].double().view(C) * sqrt(2.0 / C))})
            f1 = self.devd('avae.' + p + '.adain1', lambda p=p: _eq_linear(asd, p + '.adain1.style'))
            f2 = self.devd('avae.' + p + '.adain2', lambda p=p: _eq_linear(asd, p + '.adain2.style'))
            nz = noise[b.idx].view(R, -1)
            ya = self._avae_adain(p + '.adain1', ta, nz, wn['n1'], style_pre, f1)
            w2 = self.devd('avae.' + p + '.conv2', lambda p=p: _eq_conv(asd, p + '.conv2'))
            tb = Act(self, R, ya.h, ya.w, C, p + '.conv2')
            self.conv(self.fwd, p + '.conv2', ya.t, w2['w'], tb.t, bias=w2['b'], K=3, pad=1)

            def bwd_conv2(p=p, w2=w2, ya=ya, tb=tb):
                self.grad_conv(p + '.conv2^T', tb.g, w2['w_bwd'], ya, K=3, pad=1)
            self._bwd_steps.append(bwd_conv2)
            out = self._avae_adain(p + '.adain2', tb, nz, wn['n2'], style_pre, f2)

        # ---- to_rgb (1x1) and anti_transform (x + 1) / 2
        n_purifier_steps = None
        if isinstance(cspec, ResNetSpec):
            rgb = self.devd('avae.to_rgb.raw', lambda: F.pad_conv_out(_eq_conv(asd, 'generator.to_rgb'), 3, 4))
            raw = Act(self, R, D, D, 4, 'avae.rgb')              # ga_pool_denorm reads the generated image at a pitch of 4 lanes
            self.conv(self.fwd, 'avae.to_rgb', out.t, rgb['w'], raw.t, bias=rgb['b'], K=1)
            img = Act(self, R, D // 2, D // 2, 4 * IMG_LD, 'purified_s2d')
            pd = L.PoolDenormDesc()
            pd.x, pd.y, pd.N, pd.H, pd.W, pd.k, pd.ld, pd.backward = _ptr(raw.t), _ptr(img.t), R, D, D, 1, IMG_LD, 0
            self.fwd.add(pd, 'avae.denorm_s2d')
            last = out

            def bwd_rgb():
                b_ = L.PoolDenormDesc()
                b_.dy, b_.dx, b_.N, b_.H, b_.W, b_.k, b_.ld, b_.backward = _ptr(img.g), _ptr(raw.g), R, D, D, 1, IMG_LD, 1
                self.bwd.add(b_, 'avae.denorm_s2d^T')
                raw.g_written = True
                self.grad_conv('avae.to_rgb^T', raw.g, rgb['w_bwd'], last, K=1)
            self._bwd_steps.append(bwd_rgb)
            self.purified_s2d = img
        else:
            def fold_rgb():         # (W x + b + 1) / 2 = (W / 2) x + (b + 1) / 2, pad channels stay exact zeros
                f = _eq_conv(asd, 'generator.to_rgb', prescale=0.5)
                f['b'] = f['b'] + 0.5
                return F.pad_conv_out(f, 3, IMG_LD)
            rgb = self.devd('avae.to_rgb', fold_rgb)
            img = Act(self, R, D, D, IMG_LD, 'purified_nhwc')
            self.conv(self.fwd, 'avae.to_rgb', out.t, rgb['w'], img.t, bias=rgb['b'], K=1)
            last = out

            def bwd_rgb():
                self.grad_conv('avae.to_rgb^T', img.g, rgb['w_bwd'], last, K=1)
            self._bwd_steps.append(bwd_rgb)
            self.purified_nhwc = img
        self.purified, self.dpurified = None, None
        self._purified_grad_nhwc = img
        n_purifier_steps = len(self._bwd_steps)
        self.vspec = cspec
        if isinstance(cspec, ResNetSpec):
            self.image_s2d = True
            self.logits = self._build_resnet(csd, img)
            self.image_s2d = False
        else:
            self.logits = self._build_vgg(csd, img)
        self._finish(n_purifier_steps)
        return self
