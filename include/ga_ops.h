/*
 * ga_ops.h — C-ABI of libga_ops.so: the MI355X (gfx950) kernels of the purification-under-attack hot path.
 *
 * The reference (SerezD/gen_adversarial) is pure Python on PyTorch and has no FFI for this path; its only native
 * ABI is the pybind pair `fused_bias_act` / `upfirdn2d` of the StyleGAN ops
 * (src/mlvgms_autoencoders/StyleGan_E4E/stylegan2/op/fused_bias_act.cpp:11-20, upfirdn2d.cpp:12-22), which is not on
 * the NVAE path.  The entry points below are therefore the boundary SURVEY.md §8(b) prescribes: what a maintainer
 * would bind (ctypes / torch custom op) in place of the ATen calls made by the reference modules cited per op.
 *
 * Conventions
 *   - every tensor is fp32 and lives in device memory owned by the caller; nothing is allocated inside;
 *   - activations are NHWC ("pixel-major"): element (n,h,w,c) at ((n*H+h)*W+w)*ld + c, ld >= C is the channel pitch;
 *   - every call enqueues on `stream` (a hipStream_t passed as void*) and returns immediately;
 *   - return value 0 = enqueued, <0 = rejected (GA_E_*), nothing launched;
 *   - "prologue" = element-wise function applied to an operand as it is read, "epilogue" = applied to the result
 *     before it is written.  Activations are always prologues of the consuming op so that the tensor kept for the
 *     backward pass is the pre-activation one.
 */
#ifndef GA_OPS_H
#define GA_OPS_H

#ifdef __cplusplus
extern "C" {
#endif

#define GA_OK            0
#define GA_E_BADARG     -1   /* null pointer / non-positive size */
#define GA_E_ALIGN      -2   /* pointer or pitch not aligned as the op requires */
#define GA_E_UNSUPPORTED -3  /* shape outside what the kernel implements */
#define GA_E_LAUNCH     -4   /* hipLaunch failed (see ga_last_hip_error) */

enum ga_act { GA_ACT_NONE = 0, GA_ACT_SILU = 1, GA_ACT_ELU = 2, GA_ACT_RELU = 3,
              GA_ACT_LRELU = 4 /* nn.LeakyReLU(): slope 0.01 (GradualStyleBlock, encoding/encoder.py:41-46) */,
              GA_ACT_FLRELU = 5 /* fused_leaky_relu without its bias: leaky_relu(x, 0.2) * sqrt(2) (stylegan2/op/fused_act.py:80-85,
                                   fused_bias_act_kernel.cu:18-49); elementwise passes only, not a conv prologue */ };
/* ga_conv_desc.flags — residual networks that keep PRE-activation sums (torchvision Bottleneck: out = relu(f(x) + identity)):
 *   GA_CONV_ADDEND_RELU     forward : y = ... + relu(addend)            (the identity branch is relu of the stored pre-activation)
 *   GA_CONV_ADDEND_PRE_DACT backward: y = (acc + addend) * act'(dact_x) (+ addend2): the identity branch's cotangent passes
 *                                      through the same act' as the convolution branch's */
/*   GA_CONV_PRO_PRELU       prologue = nn.PReLU(C): x -> x > 0 ? x : pro_scale[c] * x  (pro_scale = the slopes, pro_shift any
 *                           non-NULL pointer, pro_act GA_ACT_NONE, per-channel form) — bottleneck_IR_SE, encoding/helpers.py:112
 *   GA_CONV_DACT_PRELU      epilogue act' of the same: y *= dact_x > 0 ? 1 : dact_scale[c]  (dact_shift any non-NULL pointer) */
enum ga_conv_flags { GA_CONV_ADDEND_RELU = 1, GA_CONV_ADDEND_PRE_DACT = 2, GA_CONV_PRO_PRELU = 4, GA_CONV_DACT_PRELU = 8 };

/* ------------------------------------------------------------------------------------------------------------------
 * ga_conv2d — dense convolution as an fp32-MFMA implicit GEMM (v_mfma_f32_32x32x2_f32, exact fp32 fma chains).
 * Serves, with different descriptors, every dense contraction on the path, forward and backward-to-input:
 *   weight-normed 3x3 / 1x1 convs of ResidualCellEncoder, SkipDown, EncCombinerCell, DecCombinerCell, samplers,
 *   encoder_0, to_logits (NVAE/modules/architecture.py:64-218, NVAE/model.py:97-315), the 1x1 convs of
 *   ResidualCellDecoder / SkipUp (architecture.py:85-93,139-186), VGG convs and the projector head
 *   (src/classifier/model.py:31-49), i.e. what the reference runs as aten::conv2d / aten::linear.
 *
 *   y[n,ho,wo,co] = epi( bias[co] + sum_{kh,kw,c} w[co][(kh*KW+kw)*(C1+C2)+c] * in(n, hi, wi, c) )
 *   hi = (ho*sn - pad + kh) / sd   (tap skipped unless divisible and 0<=hi<Hi; same for wi)
 *   in(.,c) = c <  C1 : act_pro( pro_scale[c]*x[.,c] + pro_shift[c] )      (zero outside the image, after act)
 *             c >= C1 : x2[., c-C1]                                        (concat-free second source)
 *   epi(v)  = v * act'_dact( dact_scale[co]*dact_x[n,ho,wo,co] + dact_shift[co] ) * dact_scale[co]   (if dact_x)
 *             + addend[(bcast ? (ho,wo) : (n,ho,wo)), co]                                             (if addend)
 *             + addend2[n,ho,wo,co]                                                                   (if addend2)
 * sn/sd: (stride,1) = forward strided conv; (1,stride) = its transpose (backward-to-input) with flipped weights.
 * addend may alias y (in-place accumulation of a gradient).
 * splits > 1: K is cut into `splits` slices computed by separate workgroups into `ws`, then summed in slice order
 * (deterministic) by a second kernel that applies the epilogue — for small N*Ho*Wo*Cout with a long K.
 * ------------------------------------------------------------------------------------------------------------------ */
typedef struct ga_conv_desc {
    const float* x;          int ldx;      /* [N,Hi,Wi,ldx], channels [0,C1) used */
    const float* x2;         int ldx2;     /* optional [N,Hi,Wi,ldx2], channels [0,C2) */
    const float* w;                        /* [Cout][KH*KW*(C1+C2)] */
    const float* bias;                     /* [Cout] or NULL */
    const float* pro_scale;                /* [C1] or [N][C1] (pro_per_row) or NULL */
    const float* pro_shift;
    const float* addend;     int ldadd;    /* optional, may alias y */
    const float* addend2;    int ldadd2;   /* optional second addend [N,Ho,Wo,ldadd2], may alias y */
    const float* dact_x;     int lddact;   /* optional [N,Ho,Wo,lddact] */
    const float* dact_scale;               /* [Cout] or NULL */
    const float* dact_shift;
    float* y;                int ldy;      /* [N,Ho,Wo,ldy] */
    int N, Hi, Wi, C1, C2;
    int Ho, Wo, Cout;
    int KH, KW, sn, sd, pad;
    int pro_act, pro_per_row;
    int dact_act;
    int addend_bcast_n;                    /* addend is [Ho,Wo,ldadd], shared by all n */
    int tile;                              /* 0 = auto; 1 = 128x128, 2 = 128x64, 3 = 64x64, 4 = 128x32 (M x N);
                                              5 = 128x128, 6 = 128x64, 7 = 128x32 on the halo-staged 3x3 kernel (3x3, stride 1,
                                              pad 1, C1 % 32 == 0, 128 % Wo == 0 or Wo % 128 == 0, w_hi/w_lo given;
                                              GA_E_UNSUPPORTED otherwise); 8 = 128x128 on the same kernel with the weight
                                              fragments read from global memory (w_frag given, 128 % Wo == 0);
                                              11 = persistent weights-resident 3x3 for C1 == 32 or 64 on 8 x 16 pixel tiles (3x3, stride 1,
                                              pad 1, Ho % 8 == 0, Wo % 16 == 0, no split-K, w_frag given in the tile-11 order below) */
    int splits;                            /* split-K factor (<=1: none); needs ws */
    float* ws;                             /* split-K workspace, >= splits*N*Ho*Wo*Cout floats, or NULL */
    long ws_floats;
    unsigned x_bytes, x2_bytes, w_bytes;   /* filled in by ga_conv2d (buffer extents); callers leave them 0 */
    int dact_rep;                          /* > 1: dact_x has N/dact_rep rows, output row n reads dact_x row n / dact_rep (K cotangents
                                              per saved forward activation: the K-cotangent backward plans, see "act_rep" below) */
    const void* w_hi;                      /* optional: w split as bf16 hi + lo, both [Cout][KH*KW*(C1+C2)] (ga_split_bf16). */
    const void* w_lo;                      /* When given and the shape allows, the contraction runs as 3 bf16 MFMAs per
                                              product on the bf16 matrix cores (~2e-5 relative), else exact fp32 MFMA. */
    int addend_rep;                        /* > 1: addend has N/addend_rep rows, row n reads addend row n / addend_rep
                                              (EoT replicas sharing one encoder feature map) */
    int flags;                             /* GA_CONV_* bits below */
    const void* w_frag;                    /* optional, tile 8 only: the split weights of a 3x3 conv in MFMA-fragment order, bf16
                                              [ceil(Cout/128)][C1/32][9 taps][4 waves][2 k steps][hi | lo][64 lanes][8]: element e of
                                              lane l = W[128 t + 32 wave + (l & 31)][tap * C1 + 32 chunk + 16 kstep + 8 (l >> 5) + e]
                                              (rows >= Cout zero).  The halo kernel then reads its B fragments from global memory:
                                              no weight staging through LDS, one barrier per 32-channel chunk.
                                              Tile 11 (C1 == 32 or 64) takes another order of the same weights:
                                              [ceil(Cout/32)][9 taps][C1/16 k steps][hi | lo][64 lanes][8], element e of lane l =
                                              W[32 t + (l & 31)][tap * C1 + 16 kstep + 8 (l >> 5) + e] (rows >= Cout zero): each workgroup
                                              keeps one 36-KB (C1 == 64: 72-KB) block in LDS for its lifetime */
} ga_conv_desc;
int ga_conv2d(const ga_conv_desc* d, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * ga_dwconv5 — depthwise 5x5, pad 2 (the middle of ResidualCellDecoder, architecture.py:169), LDS-tiled, HBM-bound.
 *   forward : y = bias + dw5( act_pro(x) )              x may be half resolution (`up2`: nearest x2 folded into the
 *                                                        read — nn.UpsamplingNearest2d commutes with the 1x1/BN/SiLU
 *                                                        in front of it, architecture.py:162-168)
 *   backward: y = dw5_flipped(x) * act'_dact(dact_x)    with `pool2`: the 2x2 sum that is the adjoint of `up2`
 * ------------------------------------------------------------------------------------------------------------------ */
typedef struct ga_dwconv5_desc {
    const float* x;      /* [N,Hs,Ws,C]; Hs = H/2 if up2 else H */
    const float* w;      /* [25][C] tap-major (already flipped for the backward use) */
    const float* bias;   /* [C] or NULL */
    const float* dact_x; /* optional, [N,Ho,Wo,C] at OUTPUT resolution */
    float* y;            /* [N,Ho,Wo,C]; Ho = H/2 if pool2 else H */
    int N, H, W, C;      /* H,W = resolution at which the 5x5 window slides */
    int pro_act, dact_act, up2, pool2;
    int act_rep;         /* > 1: dact_x has N/act_rep rows, row n reads dact_x row n / act_rep */
    int _reserved;
} ga_dwconv5_desc;
int ga_dwconv5(const ga_dwconv5_desc* d, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Squeeze-and-excite (architecture.py:37-61) split into its reduction, its two tiny FCs and the residual merge.
 * ------------------------------------------------------------------------------------------------------------------ */
/* out[n,c] = scale * sum_p a[n,p,c] * (b ? b[n,p,c] : 1)      (squeeze: scale=1/HW; d(gate): b = t, scale = 0.1)
 * ws (optional, ws_floats >= 2*N*C): workspace for a two-stage reduction — the pixels of a row are split over up to
 * ws_floats / (N*C) workgroups and the partial sums added in a fixed order (StyleGAN2 style gradients: 1024^2 pixels, 32
 * channels, a handful of rows).  Without it one workgroup per (row, 64 channels) walks all P pixels. */
typedef struct ga_rowchan_reduce_desc {
    const float* a; const float* b; float* out;
    int N, P, C; float scale;
    float* ws; long ws_floats;
    /* optional second output of the same pass (round 4; StyledConv backward, generator.py:166-203 differentiated):
     *   scaled[n,p,c] = a[n,p,c] * gate[n,c] (+ skip[n,p,c])     — d x = d(x*s) * s beside the style gradient sum_p d(x*s) * x,
     * one read of `a` for both.  skip may alias scaled (accumulating into an already written gradient). */
    const float* gate; const float* skip; float* scaled;
    /* a == NULL: `a` is formed on the fly as a 1x1 transposed conv of a 4-lane tensor — a[n,p,c] = sum_{k<4} a_w[c][k] * a_src[n,p,k]
     * (a_src [N,P,4], a_w [C][4]: ToRGB's backward, generator.py:268-290 differentiated: the 32 .. 512-channel d(x*s) of a 3-channel
     * cotangent is never stored; exact fp32 products) */
    const float* a_src; const float* a_w;
} ga_rowchan_reduce_desc;
int ga_rowchan_reduce(const ga_rowchan_reduce_desc* d, void* stream);

/* forward : hid[n,:] = w1 m[n,:] + b1 (pre-ReLU, kept) ; gate[n,:] = sigmoid(w2 relu(hid) + b2)
 * backward: given dgate (already d/d gate), writes pro_scale[n,c] = res_scale*gate, pro_shift[n,c] = dm[n,c]/P where
 *           dm is the gradient w.r.t. the squeezed mean — the per-row affine prologue of the next backward GEMM. */
typedef struct ga_se_excite_desc {
    const float* m;        /* fwd: [N,C] squeezed mean */
    const float* w1; const float* b1;   /* [Hd][C], [Hd] */
    const float* w2; const float* b2;   /* [C][Hd], [C] */
    float* hid;            /* [N,Hd] (written fwd, read bwd) */
    float* gate;           /* [N,C]  (written fwd, read bwd) */
    const float* dgate;    /* bwd only */
    float* pro_scale; float* pro_shift;  /* bwd only, [N,C] */
    int N, C, Hd, P; float res_scale; int backward;
    /* fused form: when t != NULL the kernel first reduces over the P pixels itself (one workgroup per row):
     *   forward  m[c]     = (1/P) sum_p t[n,p,c]                       (m input ignored)
     *   backward dgate[c] = res_scale * sum_p dout[n,p,c] * t[n,p,c]   (dgate input ignored)            */
    const float* t; const float* dout;
    /* forward, fused form only: when out != NULL the same workgroup also writes the merge of ga_se_apply (skip_mode 0),
     *   out[n,p,c] = skip[n,p,c] + res_scale * gate[n,c] * t[n,p,c]      (skip may be NULL)
     * — one launch and one pass over t less per cell; same expression, bitwise the result of the separate launch */
    const float* skip; float* out;
    /* backward, fused form: act_rep > 1 = K cotangents per forward row.  N counts cotangent rows (dout, pro_scale, pro_shift have N
     * rows); t, gate and hid are the forward's [N/act_rep, ...] tensors and cotangent row n reads their row n / act_rep. */
    int act_rep; int _reserved;
} ga_se_excite_desc;
int ga_se_excite(const ga_se_excite_desc* d, void* stream);

/* out[n,h,w,c] = skip(n,h,w,c) + res_scale * gate[n,c] * t[n,h,w,c]      (skip may be NULL with skip_mode 0: a pure row-scale)
 * skip_mode 0: skip is [N,H,W,C]; 1: skip is [N,H/2,W/2,C] read through bilinear x2, align_corners=True;
 * 2: skip is [N,2H,2W,C] sub-sampled at the even pixels (MaxPool2d(1, 2) shortcut of bottleneck_IR_SE, helpers.py:100-101)
 * (SkipUp, architecture.py:91-93; the 1x1 conv commutes with the interpolation and is applied at low resolution). */
typedef struct ga_se_apply_desc {
    const float* skip; const float* t; const float* gate; float* out;
    int N, H, W, C; int skip_mode; float res_scale;
} ga_se_apply_desc;
int ga_se_apply(const ga_se_apply_desc* d, void* stream);

/* adjoint of bilinear x2 (align_corners=True): dlow[n,h,w,c] (+)= sum over the high-res pixels it feeds */
typedef struct ga_bilinear_up2_bwd_desc {
    const float* dhigh; float* dlow; int N, h, w, C; int accumulate;
} ga_bilinear_up2_bwd_desc;
int ga_bilinear_up2_bwd(const ga_bilinear_up2_bwd_desc* d, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Latent interpolation of NVAEDefenseModel.purify (src/defenses/ours/models.py:199-206, 246-250) with
 * Normal/soft_clamp (NVAE/modules/distributions.py:20-48):
 *   enc_mu = 5 tanh((mu_p + mu_q)/5);  dec_mu = 5 tanh(mu_p/5);  sigma = temp * exp(5 tanh(logsig_p/5))
 *   z = (1-alpha) enc_mu + alpha (eps*sigma + dec_mu)
 * mu_q: [N,h,w,ldq] channels [0,NL); p: [N,h,w,ldp] = (mu_p | logsig_p) or NULL for the first group (prior N(0,1)).
 * eps is read in the reference's NCHW order [N,NL,h,w] when eps_nchw, else NHWC.
 * backward (dz given): writes dmu_q [N,h,w,ldq] and dp [N,h,w,ldp].
 * ------------------------------------------------------------------------------------------------------------------ */
typedef struct ga_sampler_desc {
    const float* mu_q; int ldq;
    const float* p;    int ldp;
    const float* eps;  int eps_nchw;
    float* z;          /* fwd out [N,h,w,ldz] channels [0,NL) */
    const float* dz;   /* bwd in */
    float* dmu_q; float* dp;   /* bwd out */
    int N, h, w, NL; float alpha, one_minus_alpha, temp; int backward;   /* one_minus_alpha = (float)(1.0 - alpha_double) */
    int q_rep;         /* > 1: mu_q (and dmu_q) have N/q_rep rows; row n uses row n / q_rep.  In the backward pass dmu_q is then
                          NOT written (several rows map to one): the caller gets d(mu_q) per row in `dmu_q_rows` [N,h,w,NL]
                          and reduces it with ga_rep_sum */
    float* dmu_q_rows;
    int ldz;           /* channel pitch of z and dz (0: NL).  Latent tensors padded to a multiple of 8 channels (zeros) keep the convs
                          around them on the split-bf16 kernels; dmu_q_rows then has pitch ldq like dmu_q */
    int act_rep;       /* backward, > 1: K cotangents per forward row.  N counts cotangent rows (dz, dp, dmu_q / dmu_q_rows); p and eps
                          are the forward's [N/act_rep, ...] tensors read at row n / act_rep, mu_q at row (n / act_rep) / q_rep.  dmu_q
                          is then written per cotangent row only through dmu_q_rows when q_rep > 1 (as without act_rep) */
    int mode;          /* 0: the interpolation above.  1: the posterior sample of the ND-VAE competitor's Sampler
                          (src/defenses/competitors/nd_vae/modules/models/NVAE.py:608-634, Normal :88-101): mu_q is [N,h,w,ldq] =
                          (mu_q | logsig_q), p is [N,h,w,ldp] = (mu_p | logsig_p), ldq, ldp >= 2 NL,
                            z = 5 tanh((mu_q + mu_p)/5) + (exp(5 tanh((logsig_q + logsig_p)/5)) + 0.01) * eps;
                          backward writes dmu_q [N,h,w,ldq] and dp [N,h,w,ldp], channels [0, 2 NL) (the two are equal: the
                          parameters enter as sums).  alpha, temp, q_rep unused. */
    int _reserved;
} ga_sampler_desc;
int ga_sampler_mix(const ga_sampler_desc* d, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * DiscMixLogistic(...).mean() + denormalisation (distributions.py:103-129, 231-254; models.py:271-274).
 * logits [N,H,W,ld] with channel layout (nmix | nmix x (3 mu, 3 log_scale, 3 coeff)).
 * forward writes the purified image twice: NCHW [N,3,H,W] (API output) and NHWC [N,H,W,3] (classifier input).
 * backward: dimg given as NHWC [N,H,W,3] (+ optional NCHW addend), writes dlogits [N,H,W,ld] (all ld channels).
 * ------------------------------------------------------------------------------------------------------------------ */
typedef struct ga_dml_desc {
    const float* logits; int ld; int nmix;
    float* img_nchw; float* img_nhwc;
    const float* dimg_nhwc; const float* dimg_nchw; float* dlogits;
    int N, H, W; int backward;
    int ld_img;               /* channel pitch of img_nhwc / dimg_nhwc (0 = 3).  With a pitch > 3 the forward also zeroes the
                                 pad channels, so that the image can feed a vectorised conv as a ld_img-channel tensor */
    int act_rep;              /* backward, > 1: N counts cotangent rows (dimg_*, dlogits); logits has N/act_rep rows, row n reads
                                 row n / act_rep */
} ga_dml_desc;
int ga_dml_mean(const ga_dml_desc* d, void* stream);

/* 2x2/2 max pool on pre-activation maps (ReLU commutes with max), torchvision VGG 'M' entries. backward routes dy to
 * the first maximal element in (h,w) scan order. */
typedef struct ga_maxpool2_desc {
    const float* x; float* y; const float* dy; float* dx; int N, H, W, C; int backward;
    int act_rep;      /* backward, > 1: N counts cotangent rows (dy, dx); x has N/act_rep rows, row n reads x row n / act_rep */
} ga_maxpool2_desc;
int ga_maxpool2(const ga_maxpool2_desc* d, void* stream);

/* Grouped convolution with few channels per group (torchvision ResNeXt Bottleneck `conv2`: 3x3, groups = 32, 4..32
 * channels per group; classifier/model.py:52-70) and the small kernels of its backward-to-input pass.  HBM-bound,
 * vector ALU only.  x: [N,Hi,Wi,C], y: [N,Ho,Wo,C], C = groups * cg, cg % 4 == 0; w: [C][KH*KW*cg] with
 * k = (kh*KW + kw)*cg + ci_local (the rows of group g read input channels g*cg .. g*cg+cg-1).
 *   y[n,ho,wo,co] = bias[co] + sum_{kh,kw,ci} act(x[n, ho*stride - pad + kh, wo*stride - pad + kw, g*cg + ci]) * w[co][...]
 *   then y *= act'(dact_x) when dact_x is given (backward use).  Taps outside the image are skipped (zero padding), so
 *   windows anchored at the output pixel (pad 0, Ho = Hi) work as for ga_conv2d. */
typedef struct ga_gconv_desc {
    const float* x; const float* w; const float* bias; const float* dact_x; float* y;
    int N, Hi, Wi, Ho, Wo, C, cg;
    int KH, KW, stride, pad;
    int pro_act, dact_act;
    int _reserved;
} ga_gconv_desc;
int ga_gconv(const ga_gconv_desc* d, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * StyleGAN2 modulated convolution (stylegan2/generator.py:108-207) = ga_conv2d with the per-(row, channel) prologue scale
 * (style) + the pieces below.  With W the shared weights (scaled by 1/sqrt(fan_in)), s[n,ci] the style:
 *   t = conv(W, x * s);  demod[n,co] = rsqrt(sum_ci W2[co,ci] s[n,ci]^2 + 1e-8),  W2 = sum_taps W^2;  y = demod * t
 * ------------------------------------------------------------------------------------------------------------------ */
/* small per-(row, channel) maps on [n] floats:
 *   mode 0  y = x^2                         mode 1  y = 2 x g                (backward of 0: g = d/dy, x the forward input)
 *   mode 2  y = rsqrt(x + eps)              mode 3  y = -0.5 g x^2           (x = the forward OUTPUT demod, g = sum_p dt*t:
 *                                                    d/d(sum W2 s^2) of the loss when g is taken w.r.t. demod*... see DESIGN) */
typedef struct ga_unary_desc { const float* x; const float* g; float* y; long n; int mode; float eps; } ga_unary_desc;
int ga_unary(const ga_unary_desc* d, void* stream);

/* StyledConv's tail (generator.py:258-265: demodulation folded out of the weights, NoiseInjection, FusedLeakyReLU):
 *   forward : out[n,p,c] = act(u),  u = scale[n,c] * t[n,p,c] + add[p,c]        (scale / add may be NULL = 1 / 0)
 *   backward: dt[n,p,c]  = dout * act'(u) * scale[n,c]                          (u recomputed from t)
 * t, out, dout, dt: [N,P,C]; scale: [N,C]; add: [P,C] (noise strength * noise[p] + bias[c], row independent). C % 4 == 0. */
typedef struct ga_modout_desc {
    const float* t; const float* scale; const float* add; float* out; const float* dout; float* dt;
    int N, P, C; int act; int backward;
    int W;                    /* row width of the P = H*W pixels; only read when dt_planes is used */
    float* dt_planes[4];      /* backward, optional: dt is ALSO written de-interleaved, pixel (h, w) -> plane (h&1)*2 + (w&1) at
                                 [n, h/2, w/2, C] — the operands of the up-sampling layer's four parity backward convs */
    int ld_planes;            /* channel pitch of dt_planes (0 = C) */
    int _reserved2;
    float* red;               /* backward, optional: red[n, c] = sum_p dt[n,p,c] * t[n,p,c] (the demodulation gradient's reduction,
                                 fused into this pass; deterministic two-stage sum through ws, ws_floats >= N*C) */
    float* ws; long ws_floats;
    const float* t_planes[4]; /* optional (round 4): t is READ in the depth-to-space form the up-sampling layer's parity conv writes
                                 (pixel (h, w) at plane (h&1)*2 + (w&1), [n, h/2, w/2, ld_planes]) instead of from `t` (then NULL):
                                 no interleave pass between the conv and its tail.  With t_planes AND dt_planes given, `dt` may be
                                 NULL (the parity adjoint reads the planes only). */
} ga_modout_desc;
int ga_modout(const ga_modout_desc* d, void* stream);

/* ToRGB's skip path (generator.py:29-46,286-288): Upsample = upfirdn2d(skip, outer(k,k)/16 * 4, up 2, pad (2, 1)), k = [1,3,3,1];
 * per axis  out[2U] = 1/4 s[U-1] + 3/4 s[U],  out[2U+1] = 3/4 s[U] + 1/4 s[U+1]  (zero beyond the border).
 *   forward : hi[n,2H,2W,C] += up(lo[n,H,W,C])          backward: lo[n,H,W,C] = up^T(hi)   (written, not accumulated)
 * C % 4 == 0. */
typedef struct ga_up2_blur_desc { const float* lo_in; float* hi; const float* hi_in; float* lo; int N, H, W, C; int backward; int _reserved; } ga_up2_blur_desc;
int ga_up2_blur(const ga_up2_blur_desc* d, void* stream);

/* PixelNorm on [rows, C] vectors (generator.py:10-15): y = x * rsqrt(mean_c(x^2) + 1e-8); the mapping network's first step.
 * Forward only: on this path its input is fresh Gaussian noise (E4EStyleGanDefenseModel.purify, src/defenses/ours/models.py:118-120). */
int ga_pixelnorm(const float* x, float* y, long rows, int C, void* stream);

/* Latent mixing of the e4e defender (src/defenses/ours/models.py:116-127 + pSp.encode's latent_avg, psp.py:93-103):
 *   forward : out[r, j, :] = (1 - alpha[j]) * (codes[r, j, :] + avg[j, :]) + alpha[j] * styles[r, j, :]
 *   backward: dcodes[r, j, :] = (1 - alpha[j]) * dout[r, j, :]
 * styles / out / dout: [R, J, D]; codes / dcodes: [R / max(rep, 1), J, D]; avg: [J, D] or NULL; alpha: [J] (device).  D % 4 == 0. */
typedef struct ga_latent_mix_desc {
    const float* codes; const float* avg; const float* styles; const float* alpha; float* out;
    const float* dout; float* dcodes; int R, J, D; int backward;
    int rep;      /* > 1: codes / dcodes have R / rep rows — the encoder ran once per image and its `rep` EoT replicas (consecutive
                     rows) share the codes; dcodes sums the replicas' cotangents in row order.  0 or 1: one code row per row */
    int _reserved;
} ga_latent_mix_desc;
int ga_latent_mix(const ga_latent_mix_desc* d, void* stream);

/* pSp.face_pool + de-normalisation + hand-over to the classifier (psp.py:26,117; abstract_models.py:184-185): k x k average
 * pooling of the generated image [N, k*H, k*W, 4] (lanes 0..2 = RGB in [-1, 1]) and y = 0.5 * mean + 0.5, written as the
 * space-to-depth image [N, H/2, W/2, 4, ld] the ResNet stem reads (pixel (h, w) -> phase (h&1)*2 + (w&1); lanes >= 3 zero).
 *   backward: dx[n, k*h + a, k*w + b, c] = 0.5 / k^2 * (dy[n, h, w, c] + dy_nchw[n, c, h, w])  (c < 3; lane 3 zero).
 *   H, W even; ld % 4 == 0. */
typedef struct ga_pool_denorm_desc {
    const float* x; float* y; const float* dy; float* dx; int N, H, W, k, ld; int backward;
    const float* dy_nchw;     /* backward, optional: a second cotangent on the pooled image as the API returns it, [N,3,H,W]
                                 (MLVGMDefenseModel.__call__(preds_only=False), abstract_models.py:190-193), added to dy */
    int band;                 /* > 0: the first and last `band` rows of the POOLED image are set to -1 before the de-normalisation
                                 (y = 0) and pass no gradient — `images[:, :, :32] = -1; images[:, :, -32:] = -1` of
                                 TransStyleGanDefenseModel.purify (models.py:348-349) at the pooled resolution */
    int _reserved;
} ga_pool_denorm_desc;
int ga_pool_denorm(const ga_pool_denorm_desc* d, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Style-Transformer encoder (src/mlvgms_autoencoders/StyleGan_Trans/models/transformer.py:17-100 as used by GradualStyleEncoder,
 * models/encoders/style_transformer_encoders.py:33-85) and the resize / crop glue of TransStyleGanDefenseModel.purify
 * (src/defenses/ours/models.py:299-353).  The linear layers (in_proj, out_proj, FFN) are ga_conv2d 1x1 over the token axis.
 * ------------------------------------------------------------------------------------------------------------------ */
/* Core of nn.MultiheadAttention for the encoder's 16 style queries: per (row n, head h)
 *   P = softmax_over_keys( scale * q_h k_h^T )   [Tq, Tk]        out_h = P v_h        (q, k, v AFTER their input projections)
 * q: [N, Tq, ldq], k: [N, Tk, ldk], v: [N, Tk, ldv], out: [N, Tq, ldo]; head h = channels [h*dh, (h+1)*dh) of each (pointers may
 * address the q / k / v thirds of one in_proj output).  p: [N, heads, Tq, Tk], written by the forward, read by the backward.
 * backward: dout [N, Tq, ldo] given; WRITES dq [N, Tq, lddq], dk [N, Tk, lddk], dv [N, Tk, lddv]; ds: scratch like p.
 * Tq == 16 and dh == 128 (512 channels, 4 heads) are what the path uses; dh may be any multiple of 16 up to 128, Tq is fixed. */
typedef struct ga_attn_desc {
    const float* q; const float* k; const float* v; float* out; float* p;
    const float* dout; float* ds; float* dq; float* dk; float* dv;
    int ldq, ldk, ldv, ldo, lddq, lddk, lddv;
    int N, Tq, Tk, heads, dh;
    float scale;              /* 1 / sqrt(dh) */
    int backward;
} ga_attn_desc;
int ga_attn(const ga_attn_desc* d, void* stream);

/* nn.LayerNorm(C) over the channels of x = a (+ b): y = (x - mean) * rstd * gamma + beta, biased variance, eps inside the root.
 * a, b, y, dy, dx: [rows, C]; stats: [rows, 2] = (mean, rstd), written forward, read backward.
 * backward: dx (+)= d loss / d x (the same for both summands a and b). */
typedef struct ga_layernorm_desc {
    const float* a; const float* b; const float* gamma; const float* beta; float* y; float* stats;
    const float* dy; float* dx;
    long rows; int C; float eps; int backward; int accumulate;
} ga_layernorm_desc;
int ga_layernorm(const ga_layernorm_desc* d, void* stream);

/* kornia.geometry.resize(x, 2H) (bilinear, align_corners=False) followed by the row crop x[:, :, crop:-crop] of
 * TransStyleGanDefenseModel.purify (models.py:307-308: 128 -> 256 px, rows 32:-32): x [N,H,W,C] -> y [N, 2H - 2 crop, 2W, C].
 * backward: dx (+)= the exact adjoint of dy (border clamping included).  C % 4 == 0. */
typedef struct ga_resize2_crop_desc {
    const float* x; float* y; const float* dy; float* dx; int N, H, W, C, crop; int backward; int accumulate; int _reserved;
} ga_resize2_crop_desc;
int ga_resize2_crop(const ga_resize2_crop_desc* d, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * The residual branch of NVAE's ResidualCellDecoder without up-sampling (NVAE/modules/architecture.py:139-186, BatchNorms
 * folded), ONE launch per direction; the two Hd = 6C wide tensors live in LDS / registers only (whole images per workgroup):
 *   forward  (backward = 0): y = t3 [N,H,W,C]   = W2 . silu( dw5( silu(W1 . x + b1) ) + bd ) + b2
 *   backward (backward = 1): y = dt1 [N,H,W,Hd] = silu'(t1) * dw5^T( silu'(t2) * (W2^T . (dout * pro_scale[n] + pro_shift[n])) )
 *            with t1 = W1 . x + b1 and t2 = dw5(silu(t1)) + bd recomputed from x; d x follows as one 1x1 ga_conv2d of dt1.
 * Contractions are the split-bf16 products of ga_conv2d (w*_hi / w*_lo from ga_split_bf16), the depthwise part is fp32.
 *   w1 [Hd][C];  wd [25][Hd] (tap-major, forward order), wd_bwd the flipped taps;  w2: forward [C][Hd], backward W2^T [Hd][C].
 * Shapes: ga_dec_cell_supported() (C 128 or 256, H and W powers of two, whole images per 256- / 128-pixel workgroup). */
typedef struct ga_dec_cell_desc {
    const float* x;
    const void* w1_hi; const void* w1_lo; const float* b1;
    const float* wd; const float* wd_bwd; const float* bd;
    const void* w2_hi; const void* w2_lo; const float* b2;
    const float* dout; const float* pro_scale; const float* pro_shift;
    float* y;
    int N, H, W, C, Hd; int backward;
    int act_rep;              /* backward, > 1: N counts cotangent rows (dout, pro_scale, pro_shift, y); x has N/act_rep rows, cotangent
                                 row n recomputes from x row n / act_rep */
    int variant;              /* 0: four waves per workgroup (one per SIMD); 1: eight waves (two per SIMD, half the rows each), built
                                 for C = 128 — same arithmetic, same results */
} ga_dec_cell_desc;
int ga_dec_cell(const ga_dec_cell_desc* d, void* stream);
int ga_dec_cell_supported(int N, int H, int W, int C, int Hd);   /* 1 when ga_dec_cell takes the shape */

/* ------------------------------------------------------------------------------------------------------------------
 * The same residual branch for the FEW-CHANNEL cells whose images are larger than a workgroup (the post-processing cells of
 * NVAE's decoder, architecture.py:139-186 as model.py:211-228 instantiates them: 32 channels at 64 x 64, 64 at 32 x 32): a workgroup
 * owns an 8 x 16 pixel tile and recomputes the expand conv on the tile's halo; csrc/dec_cell_halo.hip.
 *   forward  (backward = 0): y = t3 [N,H,W,C] as ga_dec_cell
 *   backward (backward = 1): y = dx [N,H,W,C] = addend + addend2 + W1^T . dt1, dt1 as ga_dec_cell's backward (never stored);
 *            w2 = W2^T [Hd][C] as there, w1t = W1^T [C][Hd] (the backward weight of the expand conv), addends optional;
 *            each addend is [N,H,W,C], 16-byte aligned, and MAY ALIAS y (every element is read and written by the same thread:
 *            in-place accumulation into an already written gradient).
 * Cin == Cout in {32, 64}, Hd % 32 == 0, H % 8 == 0, W % 16 == 0 (ga_dec_cell_halo_supported); up must be 0. */
typedef struct ga_dec_cell_halo_desc {
    const float* x;
    const void* w1_hi; const void* w1_lo; const float* b1;
    const float* wd; const float* wd_bwd; const float* bd;
    const void* w2_hi; const void* w2_lo; const float* b2;
    const void* w1t_hi; const void* w1t_lo;
    const float* dout; const float* pro_scale; const float* pro_shift;
    const float* addend; const float* addend2;
    float* y;
    int N, H, W, Cin, Cout, Hd; int backward, up;
} ga_dec_cell_halo_desc;
int ga_dec_cell_halo(const ga_dec_cell_halo_desc* d, void* stream);
int ga_dec_cell_halo_supported(int N, int H, int W, int C, int Hd);

/* ------------------------------------------------------------------------------------------------------------------
 * Pieces of the A-VAE competitor purifier (src/defenses/competitors/a_vae; csrc/avae.hip), selected by `mode`:
 *   GA_AVAE_ADAIN     forward : u = lrelu_0.2(x + b[c] * a[n,p]);  y = c[n,c] * (u - mean_p u) * rstd + c[n,C+c];  y2 = stats [N,C,2]
 *                     backward: dy -> y = d/dx [N,P,C], y2 = (d gamma | d beta) [N,2C]; s = the forward's stats
 *                     x: [N,P,C]; a: noise [N,P] or NULL (then b NULL); b: [C]; c: style output [N,2C] = (gamma | beta)
 *   GA_AVAE_AVGPOOL   k x k / stride k mean, x [N,H,W,C] -> y [N,H/k,W/k,C]; backward dy [N,H/k,W/k,C] -> y [N,H,W,C]
 *   GA_AVAE_PIXELNORM x [N,C]: y = x * rsqrt(mean_c x^2 + 1e-8); backward (dy, x) -> y = dx
 *   GA_AVAE_SAMPLE    x = t [N,P,2C], a = eps [N,C,P] (NCHW): z[n,p,c] = lrelu(t[..c]) + eps * exp(0.5 lrelu(t[..C+c])) * f0;
 *                     backward dy = dz [N,P,C] -> y = dt [N,P,2C]
 * ------------------------------------------------------------------------------------------------------------------ */
enum { GA_AVAE_ADAIN = 0, GA_AVAE_AVGPOOL = 1, GA_AVAE_PIXELNORM = 2, GA_AVAE_SAMPLE = 3 };
typedef struct ga_avae_desc {
    const float* x; const float* a; const float* b; const float* c; const float* s; const float* dy;
    float* y; float* y2;
    int mode, backward;
    int N, P, C, k, H, W;
    float f0; int _reserved;
} ga_avae_desc;
int ga_avae(const ga_avae_desc* d, void* stream);

/* On-box peak microbenchmarks for bench.py's roofline.frac_of_measured_peak; not part of the path.
 * ga_microbench_hbm_copy: dst = src, n_floats % 4 == 0, 16 B per lane (2 * 4 * n_floats bytes of traffic per call).
 * ga_microbench_mfma_bf16: a bare v_mfma_f32_32x32x16_bf16 loop on pseudo-random operands, `blocks` workgroups of 4 waves;
 *   flops per call = blocks * 4 * iters * 8 * (2 * 32 * 32 * 16); out: >= blocks * 256 floats (keeps the accumulators live). */
int ga_microbench_hbm_copy(const float* src, float* dst, long n_floats, void* stream);
int ga_microbench_mfma_bf16(float* out, int blocks, int iters, void* stream);
/* the same loop and the same flops per call on the MFMA shape `shape`: 32 = v_mfma_f32_32x32x16_bf16 (as above),
 * 16 = v_mfma_f32_16x16x32_bf16 -- under the chip's power limit the clock it holds depends on the shape. */
int ga_microbench_mfma_bf16_shape(float* out, int blocks, int iters, int shape, void* stream);

/* nn.PReLU(C) as a stand-alone pass (the input layer of the e4e encoder, encoder.py:72-74, whose output feeds both an
 * affine prologue and a shortcut): forward y = x > 0 ? x : slope[c] * x; backward dx = dy * (x > 0 ? 1 : slope[c]).
 * x: [rows][C], C % 4 == 0. */
typedef struct ga_prelu_desc {
    const float* x; const float* slope; float* y; const float* dy; float* dx; long rows; int C; int backward;
} ga_prelu_desc;
int ga_prelu(const ga_prelu_desc* d, void* stream);

/* 3x3 / stride 2 / pad 1 max pool (torchvision ResNet stem, resnet.py `self.maxpool`), on pre-activation maps (ReLU
 * commutes with max).  x: [N,H,W,C], y: [N,H/2,W/2,C] (H, W even).  backward: dx[p] = sum of dy over the windows whose
 * first maximal element in scan order is p (aten::max_pool2d_with_indices keeps the first), gathered per input pixel:
 * deterministic, no atomics. */
typedef struct ga_maxpool3s2_desc {
    const float* x; float* y; const float* dy; float* dx; int N, H, W, C; int backward;
} ga_maxpool3s2_desc;
int ga_maxpool3s2(const ga_maxpool3s2_desc* d, void* stream);

/* Global average pool with an activation prologue (torchvision ResNet `avgpool` after the last block's ReLU):
 * forward  y[n,c] = mean_p act(x[n,p,c]);  backward dx[n,p,c] = dy[n,c] / P * act'(x[n,p,c]).  x: [N,P,C], C % 4 == 0. */
typedef struct ga_avgpool_act_desc {
    const float* x; float* y; const float* dy; float* dx; int N, P, C; int act; int backward; int _reserved;
} ga_avgpool_act_desc;
int ga_avgpool_act(const ga_avgpool_act_desc* d, void* stream);

/* Image boundary: NCHW [N,3,H,W] in [0,1] <-> NHWC with MLVGMDefenseModel.add_gaussian_noise
 * (abstract_models.py:129-143): out = clamp(x[n / rep] + noise * noise_coef[n], 0, 1); `rep` folds EoTWrapper's
 * x.repeat(eot,1,1,1) (wrappers.py:20).  noise may be NULL.  backward: dx[n,c,h,w] = dy_nhwc * 1[0 <= pre <= 1]. */
typedef struct ga_image_io_desc {
    const float* x_nchw;      /* [N/rep,C,H,W] */
    const float* noise_nchw;  /* [N,C,H,W] or NULL */
    const float* noise_coef;  /* [N] = eps / ||noise_n||_2, or NULL */
    float* y_nhwc;            /* fwd out [N,H,W,C] */
    const float* dy_nhwc;     /* bwd in  [N,H,W,C] */
    float* dx_nchw;           /* bwd out [N/rep,C,H,W]: sum over the rep rows of each image */
    int N, C, H, W; int rep; int backward;
    int ld;                   /* channel pitch of y_nhwc / dy_nhwc (0 = C); the forward zero-fills channels C..ld-1 */
    int s2d;                  /* 1: space-to-depth layout — y_nhwc is [N, H/2, W/2, 4*ld], pixel (h, w) channel c at
                                 [n, h/2, w/2, ((h&1)*2 + (w&1))*ld + c]: a stride-2 conv over the image becomes a stride-1
                                 conv over this tensor (ResNet's 7x7/2 stem = 4x4 taps x 4*ld channels) */
    int cot_rep;              /* backward, > 1: K cotangents per forward row.  dy_nhwc has N rows, cotangent row r*K + k belonging to
                                 forward row r = image*rep + j; dx_nchw is [(N/K/rep)*K, C, H, W] with
                                 dx[image*K + k] = sum_j dy[(image*rep + j)*K + k] * 1[0 <= pre(image, j) <= 1];
                                 x_nchw / noise are the forward's ([N/K/rep] images, [N/K] noise rows) */
    int _reserved;
} ga_image_io_desc;
int ga_image_io(const ga_image_io_desc* d, void* stream);

/* Separable Gaussian blur of image planes with 'reflect' border (kornia.filters.gaussian_blur2d as called by
 * MLVGMDefenseModel.apply_gaussian_blur, src/defenses/ours/abstract_models.py:145-159: k = 2^(sqrt(H)//2) - 1 taps,
 * sigma 1).  x, y: [planes][H][W] (NCHW images seen as N*C planes); taps: [k] normalised weights, k odd, k/2 < H, W.
 * backward = the exact adjoint (reflect padding included): given dy in `x`, writes dx to `y`.
 * Small planes run as one kernel with the plane in LDS; larger ones as two passes (rows, then column strips) through `tmp`. */
typedef struct ga_blur_desc {
    const float* x; float* y; const float* taps;
    int planes, H, W, k; int backward;
    int radius;               /* > 0: taps farther than `radius` from the centre are skipped — the caller asserts that they lie
                                 below the fp32 resolution of the result (sigma 1: exp(-r^2/2) < 1e-31 at r = 12; the reference's
                                 255-tap kernel at 256 px has 25 such taps).  0: all k taps */
    float* tmp;               /* intermediate planes [planes][H][W] for planes too large to blur inside LDS ((2 H W + k) * 4 >
                                 64 KB, i.e. above ~90 x 90: the 128-px cars and 256-px gender images); may be NULL otherwise */
} ga_blur_desc;
int ga_gauss_blur(const ga_blur_desc* d, void* stream);

/* w[n] fp32 -> hi[n], lo[n] bf16 with hi = bf16(w), lo = bf16(w - hi) (weight preparation for w_hi / w_lo) */
int ga_split_bf16(const float* w, void* hi, void* lo, long n, void* stream);

/* y[b, i] (+)= sum_{r < rep} x[b*rep + r, i]   (x: [rows][inner], y: [rows/rep][inner]; fixed order, deterministic):
 * gradient of a tensor shared by `rep` EoT replicas */
int ga_rep_sum(const float* x, float* y, long rows, long inner, int rep, int accumulate, void* stream);

/* Sub-pixel assembly of a stride-2 transposed convolution (the backward-to-input of a stride-2 conv, e.g.
 * ResidualCellEncoder's down-sampling conv and SkipDown, architecture.py:64-82,119-122).  The transposed conv splits
 * into four stride-1 convolutions, one per output parity (a, b): output pixel (2i+a, 2j+b) only sees the taps whose
 * parity matches, so each runs as an ordinary ga_conv2d over the half-resolution cotangent with a 1x1 / 1x2 / 2x1 / 2x2
 * kernel (weights: folding.subpixel_weights).  This op interleaves their dense results s[a][b] = [N, H/2, W/2, C] into
 * y = [N, H, W, C] and applies the conv epilogue on the way:
 *   y[n, 2i+a, 2j+b, c] = s[a][b][n, i, j, c] * act'(dact_x*dact_scale + dact_shift)*dact_scale + addend + addend2
 * A NULL s[a][b] is a zero plane (a 1x1 stride-2 conv only feeds parity (0, 0)).  addend may alias y. */
typedef struct ga_interleave2_desc {
    const float* s[2][2];
    float* y;
    const float* dact_x; const float* dact_scale; const float* dact_shift;   /* [N,H,W,C], [C], [C] or NULL */
    const float* addend; const float* addend2;                              /* [N,H,W,C] or NULL */
    int N, H, W, C;           /* output size; H, W even, C % 4 == 0 */
    int dact_act;             /* ga_act */
    int dact_prelu;           /* 1: act' = dact_x > 0 ? 1 : dact_scale[c] (nn.PReLU slopes in dact_scale, dact_shift ignored) */
    int lds;                  /* channel pitch of the source planes (0 = C): the four planes may be channel slices of ONE
                                 [N, H/2, W/2, 4C] tensor (StyleGAN2 up-sampling conv: one GEMM for all parities) */
    int dact_rep;             /* > 1: dact_x has N/dact_rep rows, output row n reads dact_x row n / dact_rep */
} ga_interleave2_desc;
int ga_interleave2(const ga_interleave2_desc* d, void* stream);

/* y = alpha*x + beta*y over n floats */
int ga_axpby(const float* x, float* y, long n, float alpha, float beta, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Plans: a forward or backward pass is a flat list of the ops above, built once on the host and replayed by ONE call.
 * ------------------------------------------------------------------------------------------------------------------ */
enum ga_op_kind { GA_OP_CONV = 1, GA_OP_DWCONV5 = 2, GA_OP_REDUCE = 3, GA_OP_SE_EXCITE = 4, GA_OP_SE_APPLY = 5,
                  GA_OP_BILINEAR_BWD = 6, GA_OP_SAMPLER = 7, GA_OP_DML = 8, GA_OP_MAXPOOL = 9, GA_OP_IMAGE_IO = 10,
                  GA_OP_AXPBY = 11, GA_OP_BLUR = 12, GA_OP_REP_SUM = 13, GA_OP_INTERLEAVE2 = 14, GA_OP_MAXPOOL3S2 = 15,
                  GA_OP_AVGPOOL_ACT = 16, GA_OP_GCONV = 17, GA_OP_PRELU = 18, GA_OP_UNARY = 19,
                  GA_OP_MODOUT = 20, GA_OP_UP2_BLUR = 21, GA_OP_PIXELNORM = 22, GA_OP_LATENT_MIX = 23,
                  GA_OP_POOL_DENORM = 24, GA_OP_ATTN = 25, GA_OP_LAYERNORM = 26, GA_OP_RESIZE2_CROP = 27, GA_OP_DEC_CELL = 28,
                  GA_OP_AVAE = 29, GA_OP_DEC_CELL_HALO = 30 };
typedef struct ga_axpby_desc { const float* x; float* y; long n; float alpha, beta; } ga_axpby_desc;
typedef struct ga_rep_sum_desc { const float* x; float* y; long rows, inner; int rep, accumulate; } ga_rep_sum_desc;
typedef struct ga_op {
    int kind;
    int _pad;
    union {
        ga_conv_desc conv; ga_dwconv5_desc dw; ga_rowchan_reduce_desc red; ga_se_excite_desc se; ga_se_apply_desc app;
        ga_bilinear_up2_bwd_desc bil; ga_sampler_desc smp; ga_dml_desc dml; ga_maxpool2_desc mp; ga_image_io_desc io;
        ga_axpby_desc ax; ga_blur_desc blur; ga_rep_sum_desc rs; ga_interleave2_desc il;
        ga_maxpool3s2_desc mp3; ga_avgpool_act_desc ap; ga_gconv_desc gc; ga_prelu_desc pr;
        ga_unary_desc un; ga_modout_desc mo; ga_up2_blur_desc ub; ga_latent_mix_desc lm; ga_pool_denorm_desc pd;
        ga_attn_desc at; ga_layernorm_desc ln; ga_resize2_crop_desc rc; ga_dec_cell_desc dc; ga_avae_desc av; ga_dec_cell_halo_desc dh;
        struct { const float* x; float* y; long rows; int C; } pn;
    } u;
} ga_op;
/* runs ops[0..n); returns 0 or the first failing op's error; *failed_index set when non-NULL */
int ga_plan_run(const ga_op* ops, int n, void* stream, int* failed_index);

/* timing helper for bench.py: runs the plan `iters` times between two hipEvents recorded on `stream`, returns ms,
 * and when conv_ms != NULL also the summed duration of the GA_OP_CONV launches (per-op events). */
int ga_plan_time(const ga_op* ops, int n, void* stream, int iters, float* total_ms, float* conv_ms, long* conv_launches);

/* HIP graphs: capture one replay of a plan on `stream` (not the NULL stream; the plan must have run eagerly once) into
 * an executable graph, launch it with one host call, destroy it.  The descriptors' pointers are baked in. */
int ga_graph_capture(const ga_op* ops, int n, void* stream, void** graph_out);
int ga_graph_launch(void* graph, void* stream);
int ga_graph_destroy(void* graph);

/* per-op device time (ms) of one replay: per_op_ms[n] written */
int ga_plan_profile(const ga_op* ops, int n, void* stream, float* per_op_ms);

/* testing hook: ga_conv2d convolves row sub-batches once one of its per-row operands passes 2 GB (the fast loaders' 31-bit
 * offsets); this lowers that limit so that the sub-batch path can be exercised at small sizes.  bytes <= 0 restores the
 * default; returns the previous value. */
long ga_debug_set_conv_row_limit(long bytes);

const char* ga_last_hip_error(void);
/* GA_ABI_VERSION is bumped with EVERY change of a descriptor's layout or meaning (a field added, a reserved field put to use) and
 * with every entry point added; the binding (gen_adversarial_amd/_lib.py: ABI_VERSION) refuses a library that reports another one. */
#define GA_ABI_VERSION 7
int ga_abi_version(void);
unsigned long ga_sizeof_op(void);

#ifdef __cplusplus
}
#endif
#endif
