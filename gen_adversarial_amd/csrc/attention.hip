// Style-Transformer encoder pieces (src/mlvgms_autoencoders/StyleGan_Trans/models/transformer.py:17-100 — DETR post-norm
// TransformerDecoderLayer — as GradualStyleEncoder uses it, models/encoders/style_transformer_encoders.py:33-85; and the
// resize / crop glue of TransStyleGanDefenseModel.purify, src/defenses/ours/models.py:299-353):
//   ga_attn        nn.MultiheadAttention's core for a HANDFUL of queries (16 style queries) against up to a few thousand
//                  memory tokens: softmax(q k^T / sqrt(dh)) v per (row, head), forward and backward.  ~0.1 GFLOP per row
//                  against ~290 GFLOP of the IR-SE50 trunk: vector ALU, one workgroup per (row, head), fixed summation order.
//   ga_layernorm   LayerNorm(a + b) over the channel dimension, one wavefront per token, forward and backward.
//   ga_resize2_crop  bilinear x2 (align_corners=False: kornia.geometry.resize 128 -> 256) + row crop, forward and exact adjoint.
#include "ga_common.h"

namespace ga {

__device__ __forceinline__ floatx4 ld4a(const float* p) { return *reinterpret_cast<const floatx4*>(p); }

// ------------------------------------------------------------------------------------------------------------------
// attention forward.  block = (row n, head h), 256 threads.
//   phase 1: thread t owns keys t, t + 256, ...: raw scores s[q][key] = scale * <q_q, k_key> for the Tq queries (queries in LDS)
//            -> P buffer; per-query running max
//   phase 2: block max, exp, block sum, normalise P in place
//   phase 3: out = P V with the keys split over thread groups (attn_weighted_rows)
// P: [N, heads, Tq, Tk] (kept for the backward pass).
constexpr int ATT_TQ = 16;     // style queries of GradualStyleEncoder (self.z: 1 x 16 x 512)

// out[a][0 .. dh) = post * sum_key W[a][key] * M[key][0 .. dh) for the 16 queries of one (row, head) — P V of the forward pass, dS K of the
// backward pass.  thread = (key group g, channel quad cq): 256 / (dh / 4) groups walk the keys in parallel (Tk / 8 serial steps at dh = 128;
// round 4: one query per 16 threads walked ALL keys, 3072 dependent steps, 0.9 ms per launch); the groups' partial sums are added
// through LDS in group order (fixed summation order).  red: 4096 floats.
__device__ __forceinline__ void attn_weighted_rows(const float* __restrict__ W, const int Tk, const float* __restrict__ M, const int ldm,
                                                   const int dh, float* __restrict__ out, const int ldo, const float post, float* red) {
    const int tid = threadIdx.x, Q4 = dh >> 2, G = 256 / Q4;
    const int g = tid / Q4, cq = tid - g * Q4;
    floatx4 o[ATT_TQ];
#pragma unroll
    for (int a = 0; a < ATT_TQ; ++a) o[a] = floatx4{0.f, 0.f, 0.f, 0.f};
    if (g < G) {
#pragma unroll 2
        for (int key = g; key < Tk; key += G) {
            const floatx4 m = ld4a(M + (size_t)key * ldm + 4 * cq);
#pragma unroll
            for (int a = 0; a < ATT_TQ; ++a) o[a] += W[(size_t)a * Tk + key] * m;
        }
    }
#pragma unroll
    for (int b = 0; b < ATT_TQ; b += 4) {               // four queries at a time through the 16-KB buffer
        __syncthreads();
        if (g < G) {
#pragma unroll
            for (int j = 0; j < 4; ++j) *reinterpret_cast<floatx4*>(red + (g * 4 + j) * dh + 4 * cq) = o[b + j];
        }
        __syncthreads();
        for (int t = tid; t < 4 * dh; t += 256) {
            const int j = t / dh, c = t - j * dh;
            float sum = 0.f;
            for (int gg = 0; gg < G; ++gg) sum += red[(gg * 4 + j) * dh + c];
            out[(size_t)(b + j) * ldo + c] = sum * post;
        }
    }
}

__global__ void __launch_bounds__(256) attn_fwd_kernel(const ga_attn_desc d) {
    __shared__ float qs[ATT_TQ][132];
    __shared__ __attribute__((aligned(16))) float wred[4096];
    __shared__ float red[ATT_TQ][256 / 64 + 1];
    __shared__ float stat[ATT_TQ];
    const int n = blockIdx.x / d.heads, h = blockIdx.x % d.heads;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int dh = d.dh, Tk = d.Tk;
    const float* q = d.q + (size_t)n * ATT_TQ * d.ldq + h * dh;
    const float* k = d.k + (size_t)n * Tk * d.ldk + h * dh;
    const float* v = d.v + (size_t)n * Tk * d.ldv + h * dh;
    float* P = d.p + ((size_t)n * d.heads + h) * ATT_TQ * Tk;
    for (int i = tid; i < ATT_TQ * dh; i += 256) qs[i / dh][i % dh] = q[(size_t)(i / dh) * d.ldq + (i % dh)] * d.scale;
    __syncthreads();
    float mx[ATT_TQ];
#pragma unroll
    for (int a = 0; a < ATT_TQ; ++a) mx[a] = -3.0e38f;
    for (int key = tid; key < Tk; key += 256) {
        float acc[ATT_TQ];
#pragma unroll
        for (int a = 0; a < ATT_TQ; ++a) acc[a] = 0.f;
        const float* kr = k + (size_t)key * d.ldk;
        for (int c = 0; c < dh; c += 4) {
            const floatx4 kv = ld4a(kr + c);
#pragma unroll
            for (int a = 0; a < ATT_TQ; ++a)
                acc[a] += qs[a][c] * kv[0] + qs[a][c + 1] * kv[1] + qs[a][c + 2] * kv[2] + qs[a][c + 3] * kv[3];
        }
#pragma unroll
        for (int a = 0; a < ATT_TQ; ++a) { P[(size_t)a * Tk + key] = acc[a]; mx[a] = fmaxf(mx[a], acc[a]); }
    }
    // block max per query (fixed order: lanes by shuffles, then waves)
#pragma unroll
    for (int a = 0; a < ATT_TQ; ++a) {
        float m = mx[a];
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        if (lane == 0) red[a][wave] = m;
    }
    __syncthreads();
    if (tid < ATT_TQ) stat[tid] = fmaxf(fmaxf(red[tid][0], red[tid][1]), fmaxf(red[tid][2], red[tid][3]));
    __syncthreads();
    float sm[ATT_TQ];
#pragma unroll
    for (int a = 0; a < ATT_TQ; ++a) sm[a] = 0.f;
    for (int key = tid; key < Tk; key += 256) {
#pragma unroll
        for (int a = 0; a < ATT_TQ; ++a) {
            const float e = expf(P[(size_t)a * Tk + key] - stat[a]);
            P[(size_t)a * Tk + key] = e;
            sm[a] += e;
        }
    }
    __syncthreads();
#pragma unroll
    for (int a = 0; a < ATT_TQ; ++a) {
        float s = sm[a];
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        if (lane == 0) red[a][wave] = s;
    }
    __syncthreads();
    if (tid < ATT_TQ) stat[tid] = 1.0f / (((red[tid][0] + red[tid][1]) + red[tid][2]) + red[tid][3]);
    __syncthreads();
    for (int key = tid; key < Tk; key += 256) {
#pragma unroll
        for (int a = 0; a < ATT_TQ; ++a) P[(size_t)a * Tk + key] *= stat[a];
    }
    __syncthreads();                                // P of this block is complete (same-block global writes, made visible by the barrier)
    __threadfence_block();
    // phase 3: out = P V
    attn_weighted_rows(P, Tk, v, d.ldv, dh, d.out + (size_t)n * ATT_TQ * d.ldo + h * dh, d.ldo, 1.0f, wred);
}

// attention backward.  block = (row, head).  With P the saved probabilities, dO the cotangent of the head's output:
//   dV[key] = sum_q P[q][key] dO[q]          dP[q][key] = <dO[q], V[key]>        r[q] = sum_key dP P
//   dS = P (dP - r)                          dK[key] = scale sum_q dS[q][key] Q[q]        dQ[q] = scale sum_key dS[q][key] K[key]
// dS overwrites the `ds` scratch ([N, heads, Tq, Tk]).  dq / dk / dv are WRITTEN (each block owns its head's channel slice).
__global__ void __launch_bounds__(256) attn_bwd_kernel(const ga_attn_desc d) {
    __shared__ float qs[ATT_TQ][132];
    __shared__ __attribute__((aligned(16))) float wred[4096];
    __shared__ float dos[ATT_TQ][132];
    __shared__ float red[ATT_TQ][256 / 64 + 1];
    __shared__ float rdot[ATT_TQ];
    const int n = blockIdx.x / d.heads, h = blockIdx.x % d.heads;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int dh = d.dh, Tk = d.Tk;
    const float* q = d.q + (size_t)n * ATT_TQ * d.ldq + h * dh;
    const float* k = d.k + (size_t)n * Tk * d.ldk + h * dh;
    const float* v = d.v + (size_t)n * Tk * d.ldv + h * dh;
    const float* dO = d.dout + (size_t)n * ATT_TQ * d.ldo + h * dh;
    const float* P = d.p + ((size_t)n * d.heads + h) * ATT_TQ * Tk;
    float* dS = d.ds + ((size_t)n * d.heads + h) * ATT_TQ * Tk;
    for (int i = tid; i < ATT_TQ * dh; i += 256) {
        qs[i / dh][i % dh] = q[(size_t)(i / dh) * d.ldq + (i % dh)];
        dos[i / dh][i % dh] = dO[(size_t)(i / dh) * d.ldo + (i % dh)];
    }
    __syncthreads();
    float rp[ATT_TQ];
#pragma unroll
    for (int a = 0; a < ATT_TQ; ++a) rp[a] = 0.f;
    for (int key = tid; key < Tk; key += 256) {
        float pk[ATT_TQ], dp[ATT_TQ];
#pragma unroll
        for (int a = 0; a < ATT_TQ; ++a) { pk[a] = P[(size_t)a * Tk + key]; dp[a] = 0.f; }
        const float* vr = v + (size_t)key * d.ldv;
        float* dvr = d.dv + ((size_t)n * Tk + key) * d.lddv + h * dh;
        for (int c = 0; c < dh; c += 4) {
            const floatx4 vv = ld4a(vr + c);
            floatx4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int a = 0; a < ATT_TQ; ++a) {
                dp[a] += dos[a][c] * vv[0] + dos[a][c + 1] * vv[1] + dos[a][c + 2] * vv[2] + dos[a][c + 3] * vv[3];
                acc[0] += pk[a] * dos[a][c]; acc[1] += pk[a] * dos[a][c + 1]; acc[2] += pk[a] * dos[a][c + 2]; acc[3] += pk[a] * dos[a][c + 3];
            }
            *reinterpret_cast<floatx4*>(dvr + c) = acc;
        }
#pragma unroll
        for (int a = 0; a < ATT_TQ; ++a) { dS[(size_t)a * Tk + key] = dp[a]; rp[a] += dp[a] * pk[a]; }
    }
#pragma unroll
    for (int a = 0; a < ATT_TQ; ++a) {
        float s = rp[a];
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        if (lane == 0) red[a][wave] = s;
    }
    __syncthreads();
    if (tid < ATT_TQ) rdot[tid] = ((red[tid][0] + red[tid][1]) + red[tid][2]) + red[tid][3];
    __syncthreads();
    for (int key = tid; key < Tk; key += 256) {
        float ds[ATT_TQ];
#pragma unroll
        for (int a = 0; a < ATT_TQ; ++a) {
            ds[a] = P[(size_t)a * Tk + key] * (dS[(size_t)a * Tk + key] - rdot[a]);
            dS[(size_t)a * Tk + key] = ds[a];
        }
        float* dkr = d.dk + ((size_t)n * Tk + key) * d.lddk + h * dh;
        for (int c = 0; c < dh; c += 4) {
            floatx4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int a = 0; a < ATT_TQ; ++a) {
                acc[0] += ds[a] * qs[a][c]; acc[1] += ds[a] * qs[a][c + 1]; acc[2] += ds[a] * qs[a][c + 2]; acc[3] += ds[a] * qs[a][c + 3];
            }
            *reinterpret_cast<floatx4*>(dkr + c) = acc * d.scale;
        }
    }
    __syncthreads();
    __threadfence_block();
    // dQ = scale * dS K
    attn_weighted_rows(dS, Tk, k, d.ldk, dh, d.dq + (size_t)n * ATT_TQ * d.lddq + h * dh, d.lddq, d.scale, wred);
}

// ------------------------------------------------------------------------------------------------------------------
// LayerNorm over C channels of x = a (+ b): one wavefront per token, 4 tokens per block.  stats[token] = (mean, rstd).
__global__ void __launch_bounds__(256) layernorm_kernel(const ga_layernorm_desc d) {
    const int lane = threadIdx.x & 63;
    const long tok = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tok >= d.rows) return;
    const int C = d.C;
    const float* a = d.a + tok * C;
    const float* b = d.b ? d.b + tok * C : nullptr;
    if (!d.backward) {
        float s = 0.f;
        for (int c = lane * 4; c < C; c += 256) {
            floatx4 x = ld4a(a + c);
            if (b) x += ld4a(b + c);
            s += (x[0] + x[1]) + (x[2] + x[3]);
        }
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        const float mean = s / (float)C;
        float vs = 0.f;
        for (int c = lane * 4; c < C; c += 256) {
            floatx4 x = ld4a(a + c);
            if (b) x += ld4a(b + c);
            x -= mean;
            vs += (x[0] * x[0] + x[1] * x[1]) + (x[2] * x[2] + x[3] * x[3]);
        }
        for (int o = 32; o > 0; o >>= 1) vs += __shfl_xor(vs, o);
        const float rstd = 1.0f / sqrtf(vs / (float)C + d.eps);          // biased variance, like torch.nn.LayerNorm
        if (lane == 0) { d.stats[2 * tok] = mean; d.stats[2 * tok + 1] = rstd; }
        for (int c = lane * 4; c < C; c += 256) {
            floatx4 x = ld4a(a + c);
            if (b) x += ld4a(b + c);
            *reinterpret_cast<floatx4*>(d.y + tok * C + c) = (x - mean) * rstd * ld4a(d.gamma + c) + ld4a(d.beta + c);
        }
    } else {
        // dx = rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * gamma
        const float mean = d.stats[2 * tok], rstd = d.stats[2 * tok + 1];
        float s1 = 0.f, s2 = 0.f;
        for (int c = lane * 4; c < C; c += 256) {
            floatx4 x = ld4a(a + c);
            if (b) x += ld4a(b + c);
            const floatx4 xh = (x - mean) * rstd;
            const floatx4 g = ld4a(d.dy + tok * C + c) * ld4a(d.gamma + c);
            s1 += (g[0] + g[1]) + (g[2] + g[3]);
            s2 += (g[0] * xh[0] + g[1] * xh[1]) + (g[2] * xh[2] + g[3] * xh[3]);
        }
        for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
        s1 /= (float)C; s2 /= (float)C;
        for (int c = lane * 4; c < C; c += 256) {
            floatx4 x = ld4a(a + c);
            if (b) x += ld4a(b + c);
            const floatx4 xh = (x - mean) * rstd;
            const floatx4 g = ld4a(d.dy + tok * C + c) * ld4a(d.gamma + c);
            floatx4 r = (g - s1 - xh * s2) * rstd;
            if (d.accumulate) r += ld4a(d.dx + tok * C + c);
            *reinterpret_cast<floatx4*>(d.dx + tok * C + c) = r;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// bilinear x2, align_corners=False (F.interpolate as kornia.geometry.resize calls it), rows [crop, 2H - crop) kept.
// per axis: out[2j] = 1/4 x[j-1] + 3/4 x[j], out[2j+1] = 3/4 x[j] + 1/4 x[j+1], indices clamped to the image.
__device__ __forceinline__ void up2_src(int i, int n, int& j0, int& j1, float& w0, float& w1) {
    const int j = i >> 1;
    if (i & 1) { j0 = j; j1 = min(j + 1, n - 1); w0 = 0.75f; w1 = 0.25f; }
    else { j0 = max(j - 1, 0); j1 = j; w0 = 0.25f; w1 = 0.75f; }
}

// the outputs an input index j feeds and with which weight (adjoint of the above, clamping included): up to 4 outputs
__device__ __forceinline__ int up2_dst(int j, int n, int idx[4], float wt[4]) {
    int m = 0;
    idx[m] = 2 * j; wt[m++] = (j == 0) ? 1.0f : 0.75f;                   // out[2j] = 1/4 x[j-1] + 3/4 x[j]; x[-1] is x[0]
    idx[m] = 2 * j + 1; wt[m++] = (j == n - 1) ? 1.0f : 0.75f;           // out[2j+1] = 3/4 x[j] + 1/4 x[j+1]; x[n] is x[n-1]
    if (j + 1 < n) { idx[m] = 2 * j + 2; wt[m++] = 0.25f; }              // out[2(j+1)] takes 1/4 x[j]
    if (j - 1 >= 0) { idx[m] = 2 * j - 1; wt[m++] = 0.25f; }             // out[2(j-1)+1] takes 1/4 x[j]
    return m;
}

__global__ void __launch_bounds__(256) resize2_crop_kernel(const ga_resize2_crop_desc d, const long total4) {
    const int C4 = d.C / 4, Ho = 2 * d.H - 2 * d.crop, Wo = 2 * d.W;
    if (!d.backward) {
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long)gridDim.x * 256) {
            const int c = (int)(i % C4); long r = i / C4;
            const int w = (int)(r % Wo); r /= Wo;
            const int h = (int)(r % Ho); const long n = r / Ho;
            int h0, h1, w0, w1; float a0, a1, b0, b1;
            up2_src(h + d.crop, d.H, h0, h1, a0, a1);
            up2_src(w, d.W, w0, w1, b0, b1);
            const float* x = d.x + (size_t)n * d.H * d.W * d.C + 4 * c;
            const floatx4 v = a0 * (b0 * ld4a(x + ((size_t)h0 * d.W + w0) * d.C) + b1 * ld4a(x + ((size_t)h0 * d.W + w1) * d.C)) +
                              a1 * (b0 * ld4a(x + ((size_t)h1 * d.W + w0) * d.C) + b1 * ld4a(x + ((size_t)h1 * d.W + w1) * d.C));
            *reinterpret_cast<floatx4*>(d.y + i * 4) = v;
        }
    } else {
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long)gridDim.x * 256) {
            const int c = (int)(i % C4); long r = i / C4;
            const int w = (int)(r % d.W); r /= d.W;
            const int h = (int)(r % d.H); const long n = r / d.H;
            int hi[4], wi[4]; float hw[4], ww[4];
            const int nh = up2_dst(h, d.H, hi, hw), nw = up2_dst(w, d.W, wi, ww);
            floatx4 acc = {0.f, 0.f, 0.f, 0.f};
            const float* dy = d.dy + (size_t)n * Ho * Wo * d.C + 4 * c;
            for (int a = 0; a < nh; ++a) {
                const int ho = hi[a] - d.crop;
                if (ho < 0 || ho >= Ho) continue;
                for (int b = 0; b < nw; ++b) acc += (hw[a] * ww[b]) * ld4a(dy + ((size_t)ho * Wo + wi[b]) * d.C);
            }
            float* o = d.dx + i * 4;
            if (d.accumulate) acc += ld4a(o);
            *reinterpret_cast<floatx4*>(o) = acc;
        }
    }
}

static inline unsigned grid_for2(long items) {
    long b = (items + 255) / 256;
    if (b > 8192) b = 8192;
    if (b < 1) b = 1;
    return (unsigned)b;
}

}  // namespace ga

using namespace ga;

extern "C" int ga_attn(const ga_attn_desc* d, void* s) {
    ga::clear_stale_error();
    if (!d || !d->q || !d->k || !d->v || !d->p || d->N <= 0 || d->Tk <= 0 || d->heads <= 0) return GA_E_BADARG;
    // 16 style queries, 512 / 4 heads = 128 channels per head (style_transformer_encoders.py:37-41); reduced widths down to 16
    if (d->Tq != ATT_TQ || d->dh > 128 || d->dh % 16) return GA_E_UNSUPPORTED;
    if ((d->ldq | d->ldk | d->ldv | d->ldo) % 4) return GA_E_UNSUPPORTED;
    const int E = d->heads * d->dh;                 // a row of q / k / v / out holds every head's slice
    if (E > d->ldq || E > d->ldk || E > d->ldv || E > d->ldo) return GA_E_BADARG;
    {
        const void* ptrs[] = {d->q, d->k, d->v, d->out, d->p, d->dout, d->ds, d->dq, d->dk, d->dv};
        for (const void* p : ptrs) if (p && !aligned16(p)) return GA_E_ALIGN;
    }
    if (!d->backward) {
        if (!d->out) return GA_E_BADARG;
        hipLaunchKernelGGL(attn_fwd_kernel, dim3(d->N * d->heads), dim3(256), 0, (hipStream_t)s, *d);
    } else {
        if (!d->dout || !d->ds || !d->dq || !d->dk || !d->dv) return GA_E_BADARG;
        if ((d->lddq | d->lddk | d->lddv) % 4) return GA_E_UNSUPPORTED;
        if (E > d->lddq || E > d->lddk || E > d->lddv) return GA_E_BADARG;
        hipLaunchKernelGGL(attn_bwd_kernel, dim3(d->N * d->heads), dim3(256), 0, (hipStream_t)s, *d);
    }
    return check_launch();
}

extern "C" int ga_layernorm(const ga_layernorm_desc* d, void* s) {
    ga::clear_stale_error();
    if (!d || !d->a || !d->gamma || !d->stats || d->rows <= 0 || d->C <= 0) return GA_E_BADARG;
    if (d->C % 4) return GA_E_UNSUPPORTED;
    if (!d->backward && (!d->y || !d->beta)) return GA_E_BADARG;
    if (d->backward && (!d->dy || !d->dx)) return GA_E_BADARG;
    {
        const void* ptrs[] = {d->a, d->b, d->gamma, d->beta, d->y, d->dy, d->dx};      // all read / written 16 bytes per lane
        for (const void* p : ptrs) if (p && !aligned16(p)) return GA_E_ALIGN;
    }
    hipLaunchKernelGGL(layernorm_kernel, dim3((unsigned)((d->rows + 3) / 4)), dim3(256), 0, (hipStream_t)s, *d);
    return check_launch();
}

extern "C" int ga_resize2_crop(const ga_resize2_crop_desc* d, void* s) {
    ga::clear_stale_error();
    if (!d || d->N <= 0 || d->H <= 0 || d->W <= 0 || d->C <= 0 || d->crop < 0 || 2 * d->crop >= 2 * d->H) return GA_E_BADARG;
    if (d->C % 4) return GA_E_UNSUPPORTED;
    if (!d->backward && (!d->x || !d->y)) return GA_E_BADARG;
    if (d->backward && (!d->dy || !d->dx)) return GA_E_BADARG;
    {
        const void* ptrs[] = {d->x, d->y, d->dy, d->dx};
        for (const void* p : ptrs) if (p && !aligned16(p)) return GA_E_ALIGN;
    }
    const long total4 = (long)d->N * (d->backward ? (long)d->H * d->W : (long)(2 * d->H - 2 * d->crop) * 2 * d->W) * (d->C / 4);
    hipLaunchKernelGGL(resize2_crop_kernel, dim3(grid_for2(total4)), dim3(256), 0, (hipStream_t)s, *d, total4);
    return check_launch();
}
