"""Classifier plans: VGG-11-BN, ResNet-50 / ResNeXt-50 with the reference's projector head
(src/classifier/model.py:10-70).  Mixin of engine.Engine."""
from __future__ import annotations

from typing import Optional

import torch

from . import _lib as L
from . import folding as F
from .engine_core import IMG_LD, RES_SCALE, Act, _ptr


class ClassifierBuilder:
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
                    b.x, b.dy, b.dx, b.N, b.H, b.W, b.C, b.backward = (_ptr(src.t), _ptr(pl.g), _ptr(src.g), R * self.cot_rep,
                                                                       src.h, src.w, src.c, 1)
                    b.act_rep = self.cot_rep
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
            gflat = feat.g.view(R * self.cot_rep, 1, 1, f * f * feat.c)
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

