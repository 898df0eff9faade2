// HBM-bound element-wise / reduction kernels of the purification path (NHWC, float4 = 16 B per lane where the
// channel count allows).  Each kernel cites the reference code whose arithmetic it restates.
#include "ga_common.h"

namespace ga {

// operand a of the reduce pass at (n, p, c .. c+3): stored, or formed from a 4-lane tensor by a 1x1 transposed conv (a_src, a_w)
struct reduce_a_src {
    const float* a; const float* src; floatx4 w0, w1, w2, w3; int C;
    __device__ __forceinline__ reduce_a_src(const ga_rowchan_reduce_desc& d, const int n, const int c) : C(d.C) {
        a = d.a ? d.a + (size_t)n * d.P * d.C + c : nullptr;
        src = d.a ? nullptr : d.a_src + (size_t)n * d.P * 4;
        if (!d.a) {
            w0 = *reinterpret_cast<const floatx4*>(d.a_w + (size_t)(c + 0) * 4); w1 = *reinterpret_cast<const floatx4*>(d.a_w + (size_t)(c + 1) * 4);
            w2 = *reinterpret_cast<const floatx4*>(d.a_w + (size_t)(c + 2) * 4); w3 = *reinterpret_cast<const floatx4*>(d.a_w + (size_t)(c + 3) * 4);
        }
    }
    __device__ __forceinline__ floatx4 operator()(const int p) const {
        if (a) return *reinterpret_cast<const floatx4*>(a + (size_t)p * C);
        const floatx4 s = *reinterpret_cast<const floatx4*>(src + (size_t)p * 4);
        floatx4 v;
        v[0] = w0[0] * s[0] + w0[1] * s[1] + w0[2] * s[2] + w0[3] * s[3];
        v[1] = w1[0] * s[0] + w1[1] * s[1] + w1[2] * s[2] + w1[3] * s[3];
        v[2] = w2[0] * s[0] + w2[1] * s[1] + w2[2] * s[2] + w2[3] * s[3];
        v[3] = w3[0] * s[0] + w3[1] * s[1] + w3[2] * s[2] + w3[3] * s[3];
        return v;
    }
};

// optional second output of the reduce pass: scaled = a * gate[n,c] (+ skip) with ga_se_apply's rounding (one fma per element)
__device__ __forceinline__ void reduce_scaled_store(const ga_rowchan_reduce_desc& d, const floatx4 gt, const size_t o, const floatx4 a) {
    floatx4 sk = {0.f, 0.f, 0.f, 0.f}, r;
    if (d.skip) sk = *reinterpret_cast<const floatx4*>(d.skip + o);
#pragma unroll
    for (int e = 0; e < 4; ++e) r[e] = __builtin_fmaf(gt[e], a[e], sk[e]);
    *reinterpret_cast<floatx4*>(d.scaled + o) = r;
}

// ---------------------------------------------------------------------------------------------------------------
// out[n,c] = scale * sum_p a[n,p,c] * (b ? b[n,p,c] : 1)
// SE squeeze = torch.mean(x, dim=[2,3]) (NVAE/modules/architecture.py:56) and its backward partner d(gate).
// block = 16 channel-quads x 16 pixel lanes; fixed summation order => bitwise reproducible.
// ---------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) rowchan_reduce_kernel(const ga_rowchan_reduce_desc d, const int nchunks) {
    __shared__ floatx4 part[16][16];
    const int tid = threadIdx.x, c4 = tid & 15, pl = tid >> 4;
    const int n = blockIdx.x / nchunks, chunk = blockIdx.x % nchunks;
    const int c = chunk * 64 + 4 * c4;
    floatx4 acc = {0.f, 0.f, 0.f, 0.f};
    if (c < d.C) {
        const reduce_a_src a(d, n, c);
        const float* b = d.b ? d.b + (size_t)n * d.P * d.C + c : nullptr;
        if (d.scaled) {
            const floatx4 gt = *reinterpret_cast<const floatx4*>(d.gate + (size_t)n * d.C + c);
#pragma unroll 4
            for (int p = pl; p < d.P; p += 16) {
                floatx4 v = a(p);
                reduce_scaled_store(d, gt, ((size_t)n * d.P + p) * d.C + c, v);
                if (b) v *= *reinterpret_cast<const floatx4*>(b + (size_t)p * d.C);
                acc += v;
            }
        } else {
#pragma unroll 8
            for (int p = pl; p < d.P; p += 16) {
                floatx4 v = a(p);
                if (b) v *= *reinterpret_cast<const floatx4*>(b + (size_t)p * d.C);
                acc += v;
            }
        }
    }
    part[pl][c4] = acc;
    __syncthreads();
    if (pl == 0 && c < d.C) {
        floatx4 s = part[0][c4];
#pragma unroll
        for (int i = 1; i < 16; ++i) s += part[i][c4];
        *reinterpret_cast<floatx4*>(d.out + (size_t)n * d.C + c) = s * d.scale;
    }
}

// Two-stage variant for long rows: block = (row, 64-channel chunk, pixel segment); ql channel-quad lanes x 256/ql pixel
// lanes; partial sums go to ws[(n*S + seg)*C + c] and are added in segment order by the second kernel (deterministic).
__global__ void __launch_bounds__(256) rowchan_reduce_split_kernel(const ga_rowchan_reduce_desc d, const int nchunks, const int ql,
                                                                   const int S, const int seg_len) {
    __shared__ floatx4 part[256];
    const int tid = threadIdx.x, c4 = tid % ql, pl = tid / ql, PL = 256 / ql;
    int bi = blockIdx.x;
    const int seg = bi % S; bi /= S;
    const int chunk = bi % nchunks, n = bi / nchunks;
    const int c = chunk * 64 + 4 * c4;
    floatx4 acc = {0.f, 0.f, 0.f, 0.f};
    if (c < d.C) {
        const reduce_a_src a(d, n, c);
        const float* b = d.b ? d.b + (size_t)n * d.P * d.C + c : nullptr;
        const int p1 = min(d.P, (seg + 1) * seg_len);
        if (d.scaled) {
            const floatx4 gt = *reinterpret_cast<const floatx4*>(d.gate + (size_t)n * d.C + c);
#pragma unroll 4
            for (int p = seg * seg_len + pl; p < p1; p += PL) {
                floatx4 v = a(p);
                reduce_scaled_store(d, gt, ((size_t)n * d.P + p) * d.C + c, v);
                if (b) v *= *reinterpret_cast<const floatx4*>(b + (size_t)p * d.C);
                acc += v;
            }
        } else {
#pragma unroll 8
            for (int p = seg * seg_len + pl; p < p1; p += PL) {
                floatx4 v = a(p);
                if (b) v *= *reinterpret_cast<const floatx4*>(b + (size_t)p * d.C);
                acc += v;
            }
        }
    }
    part[tid] = acc;
    __syncthreads();
    for (int st = PL >> 1; st > 0; st >>= 1) {
        if (pl < st) part[tid] += part[tid + st * ql];
        __syncthreads();
    }
    if (pl == 0 && c < d.C) *reinterpret_cast<floatx4*>(d.ws + ((size_t)n * S + seg) * d.C + c) = part[tid];
}

__global__ void __launch_bounds__(256) rowchan_reduce_final_kernel(const ga_rowchan_reduce_desc d, const int S) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)d.N * d.C) return;
    const long n = i / d.C; const int c = (int)(i % d.C);
    float acc = 0.f;
    for (int sgm = 0; sgm < S; ++sgm) acc += d.ws[((size_t)n * S + sgm) * d.C + c];
    d.out[i] = acc * d.scale;
}

// ---------------------------------------------------------------------------------------------------------------
// SE excite: relu(linear_1) -> sigmoid(linear_2) (architecture.py:57-58) and its backward.  One block per row.
// ---------------------------------------------------------------------------------------------------------------
// out = skip + res_scale * gate * t with the rounding spelled out (one multiply, one fma): ga_se_apply and the merge fused into
// ga_se_excite must give the same bits wherever the compiler would have contracted differently
__device__ __forceinline__ floatx4 se_merge_expr(const floatx4 s, const float rs, const floatx4 g, const floatx4 t) {
    floatx4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = __builtin_fmaf(rs * g[e], t[e], s[e]);
    return o;
}

__global__ void __launch_bounds__(256) se_excite_kernel(const ga_se_excite_desc d) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* s_in = sm;            // C
    float* s_hid = sm + d.C;     // Hd
    float* s_part = sm + ((d.C + d.Hd + 3) & ~3);   // fused reduction [PL][C] / FC partials, 16-B aligned
    const int n = blockIdx.x, tid = threadIdx.x;
    // backward with K cotangents per forward row (act_rep = K): t, gate and hid are the forward's tensors, row n / K
    const int na = (d.backward && d.act_rep > 1) ? n / d.act_rep : n;
    if (d.t) {
        // fused squeeze / d(gate): channel-quad lanes x pixel lanes, fixed summation order
        const int C4 = d.C >> 2, PL = 256 / C4;
        const int q = tid % C4, pl = tid / C4;
        if (pl < PL) {
            floatx4 acc = {0.f, 0.f, 0.f, 0.f};
            const float* a = d.t + (size_t)na * d.P * d.C + 4 * q;
            const float* b = d.backward ? d.dout + (size_t)n * d.P * d.C + 4 * q : nullptr;
#pragma unroll 8
            for (int p = pl; p < d.P; p += PL) {
                floatx4 v = *reinterpret_cast<const floatx4*>(a + (size_t)p * d.C);
                if (b) v *= *reinterpret_cast<const floatx4*>(b + (size_t)p * d.C);
                acc += v;
            }
            *reinterpret_cast<floatx4*>(s_part + pl * d.C + 4 * q) = acc;
        }
        __syncthreads();
        const float sc = d.backward ? d.res_scale : 1.0f / (float)d.P;
        for (int c = tid; c < d.C; c += 256) {
            float s = s_part[c];
            for (int k = 1; k < PL; ++k) s += s_part[k * d.C + c];
            s_in[c] = s * sc;
        }
        __syncthreads();
    }
    // thread -> (hidden unit j, channel chunk): all 256 threads stream w1 / w2 with 16-B loads, chunk partials are
    // combined through LDS in a fixed order (bitwise reproducible)
    const int C4 = d.C >> 2;
    int chunks = 256 / d.Hd;
    if (chunks > C4) chunks = C4;
    const int per = (C4 + chunks - 1) / chunks;                 // channel-quads per chunk
    float* s_red = s_part;                                       // [Hd][chunks] partials (reuses the reduction scratch)
    if (!d.backward) {
        if (!d.t) {
            for (int c = tid; c < d.C; c += 256) s_in[c] = d.m[(size_t)n * d.C + c];
            __syncthreads();
        }
        {
            const int j = tid / chunks, ch = tid % chunks;
            if (j < d.Hd) {
                float acc = 0.f;
                const int q0 = ch * per, q1 = min(C4, q0 + per);
                if ((d.C & 3) == 0) {
#pragma unroll 8
                    for (int q = q0; q < q1; ++q) {
                        const floatx4 w = *reinterpret_cast<const floatx4*>(d.w1 + (size_t)j * d.C + 4 * q);
                        const floatx4 v = *reinterpret_cast<const floatx4*>(s_in + 4 * q);
                        acc += w[0] * v[0] + w[1] * v[1] + w[2] * v[2] + w[3] * v[3];
                    }
                }
                s_red[j * chunks + ch] = acc;
            }
        }
        __syncthreads();
        if (tid < d.Hd) {
            float h = d.b1[tid];
            for (int k = 0; k < chunks; ++k) h += s_red[tid * chunks + k];
            if (d.C & 3) {                                       // odd channel counts: plain loop
                h = d.b1[tid];
                for (int c = 0; c < d.C; ++c) h += d.w1[(size_t)tid * d.C + c] * s_in[c];
            }
            s_hid[tid] = h;
            d.hid[(size_t)n * d.Hd + tid] = h;
        }
        __syncthreads();
        for (int c = tid; c < d.C; c += 256) {
            float acc = d.b2[c];
            const float* w = d.w2 + (size_t)c * d.Hd;
            if ((d.Hd & 3) == 0) {
#pragma unroll 8
                for (int j = 0; j < d.Hd; j += 4) {
                    const floatx4 wv = *reinterpret_cast<const floatx4*>(w + j);
                    acc += wv[0] * fmaxf(s_hid[j], 0.f) + wv[1] * fmaxf(s_hid[j + 1], 0.f) +
                           wv[2] * fmaxf(s_hid[j + 2], 0.f) + wv[3] * fmaxf(s_hid[j + 3], 0.f);
                }
            } else {
                for (int j = 0; j < d.Hd; ++j) acc += w[j] * fmaxf(s_hid[j], 0.f);
            }
            const float gv = sigmoidf_(acc);
            d.gate[(size_t)n * d.C + c] = gv;
            if (d.out) s_part[c] = gv;                  // the FC partials are consumed: the gate stays in LDS for the merge below
        }
        if (d.out) {
            // merge of ga_se_apply (skip_mode 0) on this row, t re-read while it is hot in L2 / MALL: one launch less per cell
            __syncthreads();
            const int C4m = d.C >> 2;
            const float* tb = d.t + (size_t)n * d.P * d.C;
            const float* sb = d.skip ? d.skip + (size_t)n * d.P * d.C : nullptr;
            float* ob = d.out + (size_t)n * d.P * d.C;
#pragma unroll 4
            for (int i = tid; i < d.P * C4m; i += 256) {
                const int q = i % C4m;
                const floatx4 t = *reinterpret_cast<const floatx4*>(tb + (size_t)i * 4);
                const floatx4 g = *reinterpret_cast<const floatx4*>(s_part + 4 * q);
                const floatx4 s = sb ? *reinterpret_cast<const floatx4*>(sb + (size_t)i * 4) : floatx4{0.f, 0.f, 0.f, 0.f};
                *reinterpret_cast<floatx4*>(ob + (size_t)i * 4) = se_merge_expr(s, d.res_scale, g, t);
            }
        }
    } else {
        // ds[c] = dgate * gate * (1 - gate)
        for (int c = tid; c < d.C; c += 256) {
            const float g = d.gate[(size_t)na * d.C + c];
            const float dg = d.t ? s_in[c] : d.dgate[(size_t)n * d.C + c];
            s_in[c] = dg * g * (1.f - g);
        }
        __syncthreads();
        // dh[j] = sum_c w2[c][j] * ds[c]: the [C][Hd] matrix is streamed once with coalesced loads (thread t takes
        // elements t, t+256, ... : its j = t % Hd is fixed because Hd divides 256), partials combined in a fixed order
        if (256 % d.Hd == 0) {
            const int j = tid % d.Hd, nch = 256 / d.Hd, ch = tid / d.Hd;
            float acc = 0.f;
#pragma unroll 16
            for (int c = ch; c < d.C; c += nch) acc += d.w2[(size_t)c * d.Hd + j] * s_in[c];
            s_red[j * nch + ch] = acc;
            __syncthreads();
            if (tid < d.Hd) {
                float a2 = 0.f;
                for (int k = 0; k < nch; ++k) a2 += s_red[tid * nch + k];
                s_hid[tid] = d.hid[(size_t)na * d.Hd + tid] > 0.f ? a2 : 0.f;
            }
        } else {
            if (tid < d.Hd) {
                float a2 = 0.f;
                for (int c = 0; c < d.C; ++c) a2 += d.w2[(size_t)c * d.Hd + tid] * s_in[c];
                s_hid[tid] = d.hid[(size_t)na * d.Hd + tid] > 0.f ? a2 : 0.f;
            }
        }
        __syncthreads();
        const float invP = 1.0f / (float)d.P;
        for (int c = tid; c < d.C; c += 256) {
            float acc = 0.f;
#pragma unroll 16
            for (int j = 0; j < d.Hd; ++j) acc += d.w1[(size_t)j * d.C + c] * s_hid[j];
            d.pro_scale[(size_t)n * d.C + c] = d.res_scale * d.gate[(size_t)na * d.C + c];
            d.pro_shift[(size_t)n * d.C + c] = acc * invP;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// bilinear x2, align_corners=True (F.interpolate in SkipUp, architecture.py:92): source index and lambda
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void bil_src(const int o, const int in, const int out, int& i0, int& i1, float& l1) {
    const float scale = out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f;
    const float src = scale * (float)o;
    i0 = (int)src;
    i1 = i0 + (i0 < in - 1 ? 1 : 0);
    l1 = src - (float)i0;
}

// out = skip + res_scale * gate[n,c] * t   (ResidualCell*.forward: `x + 0.1 * residual`, architecture.py:133-136,183-186,
// with the SE channel scale x * se (:61) folded in)
__global__ void __launch_bounds__(256) se_apply_kernel(const ga_se_apply_desc d, const long total4) {
    const int C4 = d.C / 4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long)gridDim.x * 256) {
        const int c4 = (int)(i % C4); long p = i / C4;
        const int w = (int)(p % d.W); p /= d.W;
        const int h = (int)(p % d.H); const int n = (int)(p / d.H);
        const int c = 4 * c4;
        const floatx4 t = *reinterpret_cast<const floatx4*>(d.t + i * 4);
        const floatx4 g = *reinterpret_cast<const floatx4*>(d.gate + (size_t)n * d.C + c);
        floatx4 s;
        if (d.skip_mode == 0) {
            s = d.skip ? *reinterpret_cast<const floatx4*>(d.skip + i * 4) : floatx4{0.f, 0.f, 0.f, 0.f};
        } else if (d.skip_mode == 2) {
            s = *reinterpret_cast<const floatx4*>(d.skip + (((size_t)n * 2 * d.H + 2 * h) * 2 * d.W + 2 * w) * d.C + c);
        } else {
            const int hl = d.H / 2, wl = d.W / 2;
            int h0, h1, w0, w1; float lh, lw;
            bil_src(h, hl, d.H, h0, h1, lh);
            bil_src(w, wl, d.W, w0, w1, lw);
            const float* base = d.skip + (size_t)n * hl * wl * d.C + c;
            const floatx4 x00 = *reinterpret_cast<const floatx4*>(base + ((size_t)h0 * wl + w0) * d.C);
            const floatx4 x01 = *reinterpret_cast<const floatx4*>(base + ((size_t)h0 * wl + w1) * d.C);
            const floatx4 x10 = *reinterpret_cast<const floatx4*>(base + ((size_t)h1 * wl + w0) * d.C);
            const floatx4 x11 = *reinterpret_cast<const floatx4*>(base + ((size_t)h1 * wl + w1) * d.C);
            const float h0l = 1.f - lh, w0l = 1.f - lw;
            s = h0l * (w0l * x00 + lw * x01) + lh * (w0l * x10 + lw * x11);
        }
        *reinterpret_cast<floatx4*>(d.out + i * 4) = se_merge_expr(s, d.res_scale, g, t);
    }
}

// adjoint of the bilinear x2 read above (gather form, deterministic)
__global__ void __launch_bounds__(256) bilinear_up2_bwd_kernel(const ga_bilinear_up2_bwd_desc d, const long total4) {
    const int C4 = d.C / 4;
    const int H = 2 * d.h, W = 2 * d.w;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long)gridDim.x * 256) {
        const int c4 = (int)(i % C4); long p = i / C4;
        const int wl = (int)(p % d.w); p /= d.w;
        const int hl = (int)(p % d.h); const int n = (int)(p / d.h);
        floatx4 acc = {0.f, 0.f, 0.f, 0.f};
        const float* base = d.dhigh + (size_t)n * H * W * d.C + 4 * c4;
        // candidate high-res rows/cols: src = o*(in-1)/(out-1) in (l-1, l+1)
        const int hlo = max(0, 2 * hl - 2), hhi = min(H - 1, 2 * hl + 4);
        const int wlo = max(0, 2 * wl - 2), whi = min(W - 1, 2 * wl + 4);
        for (int hh = hlo; hh <= hhi; ++hh) {
            int h0, h1; float lh;
            bil_src(hh, d.h, H, h0, h1, lh);
            const float wh = (h0 == hl ? 1.f - lh : 0.f) + (h1 == hl ? lh : 0.f);
            if (wh == 0.f) continue;
            for (int ww = wlo; ww <= whi; ++ww) {
                int w0, w1; float lw;
                bil_src(ww, d.w, W, w0, w1, lw);
                const float wwt = (w0 == wl ? 1.f - lw : 0.f) + (w1 == wl ? lw : 0.f);
                if (wwt == 0.f) continue;
                acc += (wh * wwt) * *reinterpret_cast<const floatx4*>(base + ((size_t)hh * W + ww) * d.C);
            }
        }
        floatx4* o = reinterpret_cast<floatx4*>(d.dlow + i * 4);
        *o = d.accumulate ? (*o + acc) : acc;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// latent interpolation (src/defenses/ours/models.py:199-206,246-250; distributions.py:20-48)
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float softclamp5(float x) { return tanhf(x / 5.0f) * 5.0f; }
__device__ __forceinline__ float dsoftclamp5(float x) { const float t = tanhf(x / 5.0f); return 1.0f - t * t; }

__global__ void __launch_bounds__(256) sampler_kernel(const ga_sampler_desc d, const long total) {
    const int hw = d.h * d.w;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = (int)(i % d.NL); const long pix = i / d.NL;            // pix = n*hw + p
        const int n = (int)(pix / hw), p = (int)(pix % hw);
        const long zi = d.ldz > 0 ? pix * d.ldz + c : i;
        // backward with K cotangents per forward row (act_rep = K): p, eps and mu_q are the forward's tensors, row n / K
        const int na = (d.backward && d.act_rep > 1) ? n / d.act_rep : n;
        const long apix = (long)na * hw + p;
        if (d.mode == 1) {      // ND-VAE posterior sample: both distributions' parameters enter as sums (NVAE.py:608-634)
            const float e1 = d.eps_nchw ? d.eps[((size_t)na * d.NL + c) * hw + p] : d.eps[apix * d.NL + c];
            const float mu_s = d.mu_q[apix * d.ldq + c] + d.p[apix * d.ldp + c];
            const float ls_s = d.mu_q[apix * d.ldq + d.NL + c] + d.p[apix * d.ldp + d.NL + c];
            const float ex = expf(softclamp5(ls_s));
            if (!d.backward) {
                d.z[zi] = softclamp5(mu_s) + (ex + 0.01f) * e1;
            } else {
                const float dz = d.dz[zi];
                const float dmu = dz * dsoftclamp5(mu_s), dls = dz * e1 * ex * dsoftclamp5(ls_s);
                d.dmu_q[pix * d.ldq + c] = dmu;
                d.dmu_q[pix * d.ldq + d.NL + c] = dls;
                d.dp[pix * d.ldp + c] = dmu;
                d.dp[pix * d.ldp + d.NL + c] = dls;
            }
            continue;
        }
        const long qpix = d.q_rep > 1 ? (long)(na / d.q_rep) * hw + p : apix;
        const float mq = d.mu_q[qpix * d.ldq + c];
        const float mp = d.p ? d.p[apix * d.ldp + c] : 0.f;
        const float ls = d.p ? d.p[apix * d.ldp + d.NL + c] : 0.f;
        const float e = d.eps_nchw ? d.eps[((size_t)na * d.NL + c) * hw + p] : d.eps[apix * d.NL + c];
        const float sig = d.temp * expf(softclamp5(ls));
        const float a = d.alpha, om = d.one_minus_alpha;
        if (!d.backward) {
            const float enc_mu = softclamp5(mp + mq);
            const float smp = e * sig + softclamp5(mp);
            d.z[zi] = om * enc_mu + a * smp;
        } else {
            const float dz = d.dz[zi];
            const float denc = om * dz * dsoftclamp5(mp + mq);
            if (d.q_rep > 1) d.dmu_q_rows[d.ldz > 0 ? pix * d.ldq + c : i] = denc; else d.dmu_q[pix * d.ldq + c] = denc;
            if (d.dp) {
                d.dp[pix * d.ldp + c] = denc + a * dz * dsoftclamp5(mp);
                d.dp[pix * d.ldp + d.NL + c] = a * dz * e * sig * dsoftclamp5(ls);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// DiscMixLogistic.mean + denormalise (distributions.py:103-129,231-254; models.py:271-274); one thread per pixel
// ---------------------------------------------------------------------------------------------------------------
constexpr int DML_MAXMIX = 16;
constexpr int DML_ROWS = 128;      // pixels per workgroup pass (LDS: 128 x (ld+1) floats)

__global__ void __launch_bounds__(DML_ROWS) dml_kernel(const ga_dml_desc d, const long npix) {
    // each thread owns one pixel, but a pixel's `ld` logits are 400 B apart from its neighbour's: the block first copies
    // its 128 rows into LDS with coalesced 4-B accesses (row pitch ld+1: conflict-free per-thread reads), computes from
    // LDS, and in the backward pass writes dlogits back through the same staging buffer.
    extern __shared__ float dml_s[];
    const int HW = d.H * d.W;
    const int pitch = d.ld + 1;
    const int ldi = d.ld_img > 0 ? d.ld_img : 3;
    const int arep = (d.backward && d.act_rep > 1) ? d.act_rep : 1;
    for (long base = (long)blockIdx.x * DML_ROWS; base < npix; base += (long)gridDim.x * DML_ROWS) {
        const int nrow = (int)min((long)DML_ROWS, npix - base);
        __syncthreads();
        if (arep > 1) {
            // K cotangents per forward row: cotangent pixel (n, p) reads the logits of forward pixel (n / K, p); a 128-pixel pass may
            // straddle rows, so the source pixel is computed per staged row
            for (int k = threadIdx.x; k < nrow * d.ld; k += DML_ROWS) {
                const int r = k / d.ld, c = k - r * d.ld;
                const long px = base + r, n = px / HW;
                dml_s[r * pitch + c] = d.logits[((n / arep) * HW + (px - n * HW)) * d.ld + c];
            }
        } else if ((d.ld & 3) == 0 && (reinterpret_cast<uintptr_t>(d.logits) & 15) == 0) {   // 16-B global accesses (a quad never straddles two pixels)
            const floatx4* src = reinterpret_cast<const floatx4*>(d.logits + base * d.ld);
            const int ld4 = d.ld >> 2;
#pragma unroll 4
            for (int k = threadIdx.x; k < nrow * ld4; k += DML_ROWS) {
                const int r = k / ld4, c = 4 * (k - r * ld4);
                const floatx4 v = src[k];
                float* o = dml_s + r * pitch + c;
                o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; o[3] = v[3];
            }
        } else {
            for (int k = threadIdx.x; k < nrow * d.ld; k += DML_ROWS) {
                const int r = k / d.ld, c = k - r * d.ld;
                dml_s[r * pitch + c] = d.logits[base * d.ld + k];
            }
        }
        __syncthreads();
        const long i = base + threadIdx.x;
        const bool active = threadIdx.x < nrow;
        float* l = dml_s + threadIdx.x * pitch;
        if (active) {
        const int nm = d.nmix;
        float p[DML_MAXMIX];
        float mx = -INFINITY;
        for (int k = 0; k < nm; ++k) mx = fmaxf(mx, l[k]);
        float den = 0.f;
        for (int k = 0; k < nm; ++k) { p[k] = expf(l[k] - mx); den += p[k]; }
        const float inv = 1.0f / den;
        float mu0 = 0.f, mu1 = 0.f, mu2 = 0.f, K0 = 0.f, K1 = 0.f, K2 = 0.f;
        for (int k = 0; k < nm; ++k) {
            p[k] *= inv;
            const float* q = l + nm + 9 * k;
            mu0 += q[0] * p[k]; mu1 += q[1] * p[k]; mu2 += q[2] * p[k];
            K0 += tanhf(q[6]) * p[k]; K1 += tanhf(q[7]) * p[k]; K2 += tanhf(q[8]) * p[k];
        }
        const float r = fminf(fmaxf(mu0, -1.f), 1.f);
        const float gpre = mu1 + K0 * r;
        const float g = fminf(fmaxf(gpre, -1.f), 1.f);
        const float bpre = mu2 + K1 * r + K2 * g;
        const float bl = fminf(fmaxf(bpre, -1.f), 1.f);
        const int n = (int)(i / HW), pp = (int)(i % HW);
        if (!d.backward) {
            const float o0 = r * 0.5f + 0.5f, o1 = g * 0.5f + 0.5f, o2 = bl * 0.5f + 0.5f;
            if (d.img_nhwc) { float* o = d.img_nhwc + i * ldi; o[0] = o0; o[1] = o1; o[2] = o2; for (int z = 3; z < ldi; ++z) o[z] = 0.f; }
            if (d.img_nchw) {
                float* o = d.img_nchw + (size_t)n * 3 * HW + pp;
                o[0] = o0; o[HW] = o1; o[2 * HW] = o2;
            }
        } else {
            float dr = 0.f, dg = 0.f, db = 0.f;
            if (d.dimg_nhwc) { const float* q = d.dimg_nhwc + i * ldi; dr += q[0]; dg += q[1]; db += q[2]; }
            if (d.dimg_nchw) { const float* q = d.dimg_nchw + (size_t)n * 3 * HW + pp; dr += q[0]; dg += q[HW]; db += q[2 * HW]; }
            dr *= 0.5f; dg *= 0.5f; db *= 0.5f;
            const float dbp = (bpre >= -1.f && bpre <= 1.f) ? db : 0.f;
            const float dmu2 = dbp, dK1 = dbp * r, dK2 = dbp * g;
            dr += dbp * K1; dg += dbp * K2;
            const float dgp = (gpre >= -1.f && gpre <= 1.f) ? dg : 0.f;
            const float dmu1 = dgp, dK0 = dgp * r;
            dr += dgp * K0;
            const float dmu0 = (mu0 >= -1.f && mu0 <= 1.f) ? dr : 0.f;
            float* o = l;                                   // staged: written back coalesced below
            float dpk[DML_MAXMIX];
            float dot = 0.f;
            for (int k = 0; k < nm; ++k) {
                const float* q = l + nm + 9 * k;
                const float t0 = tanhf(q[6]), t1 = tanhf(q[7]), t2 = tanhf(q[8]);
                dpk[k] = q[0] * dmu0 + q[1] * dmu1 + q[2] * dmu2 + t0 * dK0 + t1 * dK1 + t2 * dK2;
                dot += p[k] * dpk[k];
                float* oq = o + nm + 9 * k;
                oq[0] = p[k] * dmu0; oq[1] = p[k] * dmu1; oq[2] = p[k] * dmu2;
                oq[3] = 0.f; oq[4] = 0.f; oq[5] = 0.f;
                oq[6] = p[k] * dK0 * (1.f - t0 * t0); oq[7] = p[k] * dK1 * (1.f - t1 * t1); oq[8] = p[k] * dK2 * (1.f - t2 * t2);
            }
            for (int k = 0; k < nm; ++k) o[k] = p[k] * (dpk[k] - dot);
            for (int k = nm + 9 * nm; k < d.ld; ++k) o[k] = 0.f;
        }
        }   // active
        if (d.backward) {
            __syncthreads();
            if ((d.ld & 3) == 0 && (reinterpret_cast<uintptr_t>(d.dlogits) & 15) == 0) {
                floatx4* dst = reinterpret_cast<floatx4*>(d.dlogits + base * d.ld);
                const int ld4 = d.ld >> 2;
#pragma unroll 4
                for (int k = threadIdx.x; k < nrow * ld4; k += DML_ROWS) {
                    const int r = k / ld4, c = 4 * (k - r * ld4);
                    const float* o = dml_s + r * pitch + c;
                    const floatx4 v = {o[0], o[1], o[2], o[3]};
                    dst[k] = v;
                }
            } else {
                for (int k = threadIdx.x; k < nrow * d.ld; k += DML_ROWS) {
                    const int r = k / d.ld, c = k - r * d.ld;
                    d.dlogits[base * d.ld + k] = dml_s[r * pitch + c];
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// 2x2/2 max pool (torchvision VGG 'M'), on pre-activation maps
// ---------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) maxpool2_kernel(const ga_maxpool2_desc d, const long total4) {
    const int C4 = d.C / 4, Ho = d.H / 2, Wo = d.W / 2;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long)gridDim.x * 256) {
        const int c4 = (int)(i % C4); long p = i / C4;
        const int wo = (int)(p % Wo); p /= Wo;
        const int ho = (int)(p % Ho); const int n = (int)(p / Ho);
        const size_t base = (((size_t)n * d.H + 2 * ho) * d.W + 2 * wo) * d.C + 4 * c4;
        // backward with K cotangents per forward row (act_rep = K): the decisions come from the forward's x, row n / K
        const int na = (d.backward && d.act_rep > 1) ? n / d.act_rep : n;
        const size_t xb = (((size_t)na * d.H + 2 * ho) * d.W + 2 * wo) * d.C + 4 * c4;
        const floatx4 a = *reinterpret_cast<const floatx4*>(d.x + xb);
        const floatx4 b = *reinterpret_cast<const floatx4*>(d.x + xb + d.C);
        const floatx4 c = *reinterpret_cast<const floatx4*>(d.x + xb + (size_t)d.W * d.C);
        const floatx4 e = *reinterpret_cast<const floatx4*>(d.x + xb + (size_t)d.W * d.C + d.C);
        if (!d.backward) {
            floatx4 m;
#pragma unroll
            for (int k = 0; k < 4; ++k) m[k] = fmaxf(fmaxf(a[k], b[k]), fmaxf(c[k], e[k]));
            *reinterpret_cast<floatx4*>(d.y + i * 4) = m;
        } else {
            const floatx4 g = *reinterpret_cast<const floatx4*>(d.dy + i * 4);
            floatx4 ga_ = {0.f, 0.f, 0.f, 0.f}, gb = ga_, gc = ga_, ge = ga_;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                // first maximal element in scan order (a, b, c, e), as aten::max_pool2d_with_indices keeps it
                float m = a[k]; int w = 0;
                if (b[k] > m) { m = b[k]; w = 1; }
                if (c[k] > m) { m = c[k]; w = 2; }
                if (e[k] > m) { m = e[k]; w = 3; }
                if (w == 0) ga_[k] = g[k]; else if (w == 1) gb[k] = g[k]; else if (w == 2) gc[k] = g[k]; else ge[k] = g[k];
            }
            *reinterpret_cast<floatx4*>(d.dx + base) = ga_;
            *reinterpret_cast<floatx4*>(d.dx + base + d.C) = gb;
            *reinterpret_cast<floatx4*>(d.dx + base + (size_t)d.W * d.C) = gc;
            *reinterpret_cast<floatx4*>(d.dx + base + (size_t)d.W * d.C + d.C) = ge;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// 3x3 / stride 2 / pad 1 max pool (torchvision ResNet stem) on pre-activation maps; backward gathers per input pixel
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ floatx4 ld4(const float* p) { return *reinterpret_cast<const floatx4*>(p); }

__global__ void __launch_bounds__(256) maxpool3s2_kernel(const ga_maxpool3s2_desc d, const long total4) {
    const int C4 = d.C / 4, Ho = d.H / 2, Wo = d.W / 2;
    if (!d.backward) {
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long)gridDim.x * 256) {
            const int c4 = (int)(i % C4); long p = i / C4;
            const int wo = (int)(p % Wo); p /= Wo;
            const int ho = (int)(p % Ho); const int n = (int)(p / Ho);
            floatx4 m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
            for (int kh = 0; kh < 3; ++kh) {
                const int h = 2 * ho - 1 + kh;
                if (h < 0 || h >= d.H) continue;
                for (int kw = 0; kw < 3; ++kw) {
                    const int w = 2 * wo - 1 + kw;
                    if (w < 0 || w >= d.W) continue;
                    const floatx4 v = ld4(d.x + (((size_t)n * d.H + h) * d.W + w) * d.C + 4 * c4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) m[e] = fmaxf(m[e], v[e]);
                }
            }
            *reinterpret_cast<floatx4*>(d.y + i * 4) = m;
        }
    } else {
        // one thread per INPUT pixel x channel quad: the (at most 4) windows that contain it are re-scanned in the
        // forward order; the pixel receives dy of a window iff it is that window's first maximum
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long)gridDim.x * 256) {
            const int c4 = (int)(i % C4); long p = i / C4;
            const int w = (int)(p % d.W); p /= d.W;
            const int h = (int)(p % d.H); const int n = (int)(p / d.H);
            const floatx4 me = ld4(d.x + (((size_t)n * d.H + h) * d.W + w) * d.C + 4 * c4);
            floatx4 g = {0.f, 0.f, 0.f, 0.f};
            const int ho0 = max(0, h / 2), ho1 = min(Ho - 1, (h + 1) / 2);                 // windows with 2ho-1 <= h <= 2ho+1
            const int wo0 = max(0, w / 2), wo1 = min(Wo - 1, (w + 1) / 2);
            for (int ho = ho0; ho <= ho1; ++ho)
                for (int wo = wo0; wo <= wo1; ++wo) {
                    // is (h, w) the first maximum of window (ho, wo)?  earlier positions must be strictly smaller is NOT the
                    // rule: aten keeps an earlier element on ties (it replaces only on val > max), so "first maximum" = no
                    // earlier element >= me and no later element > me
                    unsigned win = 0xf;                                  // per channel of the quad
                    for (int kh = 0; kh < 3; ++kh) {
                        const int hh = 2 * ho - 1 + kh;
                        if (hh < 0 || hh >= d.H) continue;
                        for (int kw = 0; kw < 3; ++kw) {
                            const int ww = 2 * wo - 1 + kw;
                            if (ww < 0 || ww >= d.W || (hh == h && ww == w)) continue;
                            const floatx4 v = ld4(d.x + (((size_t)n * d.H + hh) * d.W + ww) * d.C + 4 * c4);
                            const bool earlier = hh < h || (hh == h && ww < w);
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                if (earlier ? v[e] >= me[e] : v[e] > me[e]) win &= ~(1u << e);
                        }
                    }
                    const floatx4 dy = ld4(d.dy + (((size_t)n * Ho + ho) * Wo + wo) * d.C + 4 * c4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) if (win & (1u << e)) g[e] += dy[e];
                }
            *reinterpret_cast<floatx4*>(d.dx + i * 4) = g;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// StyleGAN2 modulated conv pieces: small maps on the style / demodulation vectors, and the StyledConv tail
// ---------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) unary_kernel(const ga_unary_desc d) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < d.n; i += (long)gridDim.x * 256) {
        const float x = d.x[i];
        float y;
        switch (d.mode) {
            case 0: y = x * x; break;
            case 1: y = 2.0f * x * d.g[i]; break;
            case 2: y = rsqrtf(x + d.eps); break;
            default: y = -0.5f * d.g[i] * x * x; break;
        }
        d.y[i] = y;
    }
}

// address of t[n, p, c] for the StyledConv tail: interleaved [N,P,C] or the depth-to-space planes of the up-sampling layer's conv
__device__ __forceinline__ const float* modout_t_ptr(const ga_modout_desc& d, const long n, const int p, const int c) {
    if (!d.t_planes[0]) return d.t + ((size_t)n * d.P + p) * d.C + c;
    const int h = p / d.W, w = p - h * d.W, H2 = (d.P / d.W) >> 1, W2 = d.W >> 1;
    return d.t_planes[(h & 1) * 2 + (w & 1)] + (((size_t)n * H2 + (h >> 1)) * W2 + (w >> 1)) * (d.ld_planes > 0 ? d.ld_planes : d.C) + c;
}

__global__ void __launch_bounds__(256) modout_kernel(const ga_modout_desc d, const long total4) {
    const int C4 = d.C / 4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long)gridDim.x * 256) {
        const int q = (int)(i % C4); const long r = i / C4;
        const int p = (int)(r % d.P); const long n = r / d.P;
        floatx4 sc = {1.f, 1.f, 1.f, 1.f}, ad = {0.f, 0.f, 0.f, 0.f};
        if (d.scale) sc = ld4(d.scale + n * d.C + 4 * q);
        if (d.add) ad = ld4(d.add + (size_t)p * d.C + 4 * q);
        const floatx4 u = sc * ld4(modout_t_ptr(d, n, p, 4 * q)) + ad;
        floatx4 o;
        if (!d.backward) {
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = act_fwd(u[e], d.act);
            *reinterpret_cast<floatx4*>(d.out + i * 4) = o;
        } else {
            const floatx4 g = ld4(d.dout + i * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = g[e] * act_bwd(u[e], d.act) * sc[e];
            if (d.dt) *reinterpret_cast<floatx4*>(d.dt + i * 4) = o;
            if (d.dt_planes[0]) {
                const int h = p / d.W, w = p - h * d.W, H2 = (d.P / d.W) >> 1, W2 = d.W >> 1;
                float* pl = d.dt_planes[(h & 1) * 2 + (w & 1)];
                *reinterpret_cast<floatx4*>(pl + (((size_t)n * H2 + (h >> 1)) * W2 + (w >> 1)) * (d.ld_planes > 0 ? d.ld_planes : d.C) + 4 * q) = o;
            }
        }
    }
}

__global__ void __launch_bounds__(256) up2_blur_kernel(const ga_up2_blur_desc d, const long total4) {
    const int C4 = d.C / 4;
    const int Ho = d.backward ? d.H : 2 * d.H, Wo = d.backward ? d.W : 2 * d.W;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long)gridDim.x * 256) {
        const int q = (int)(i % C4); long r = i / C4;
        const int v = (int)(r % Wo); r /= Wo;
        const int u = (int)(r % Ho); const long n = r / Ho;
        floatx4 acc = {0.f, 0.f, 0.f, 0.f};
        if (!d.backward) {
            // rows U0 / U0+1 with weights wy0 / wy1 (even u: U-1, U with 1/4, 3/4; odd u: U, U+1 with 3/4, 1/4)
            const int U0 = (u >> 1) - 1 + (u & 1), V0 = (v >> 1) - 1 + (v & 1);
            const float wy0 = (u & 1) ? 0.75f : 0.25f, wx0 = (v & 1) ? 0.75f : 0.25f;
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const int U = U0 + a;
                if (U < 0 || U >= d.H) continue;
                const float wy = a ? 1.0f - wy0 : wy0;
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const int V = V0 + b;
                    if (V < 0 || V >= d.W) continue;
                    const float wgt = wy * (b ? 1.0f - wx0 : wx0);
                    acc += wgt * ld4(d.lo_in + (((size_t)n * d.H + U) * d.W + V) * d.C + 4 * q);
                }
            }
            float* o = d.hi + i * 4;
            *reinterpret_cast<floatx4*>(o) = ld4(o) + acc;
        } else {
            const float k4[4] = {0.25f, 0.75f, 0.75f, 0.25f};
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const int uu = 2 * u - 1 + a;
                if (uu < 0 || uu >= 2 * d.H) continue;
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const int vv = 2 * v - 1 + b;
                    if (vv < 0 || vv >= 2 * d.W) continue;
                    acc += (k4[a] * k4[b]) * ld4(d.hi_in + (((size_t)n * 2 * d.H + uu) * 2 * d.W + vv) * d.C + 4 * q);
                }
            }
            *reinterpret_cast<floatx4*>(d.lo + i * 4) = acc;
        }
    }
}

// one wavefront per row
__global__ void __launch_bounds__(256) pixelnorm_kernel(const float* __restrict__ x, float* __restrict__ y, const long rows, const int C) {
    const int lane = threadIdx.x & 63;
    for (long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6); r < rows; r += (long)gridDim.x * 4) {
        const float* xr = x + r * C;
        float acc = 0.f;
        for (int c = lane; c < C; c += 64) acc += xr[c] * xr[c];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
        const float inv = rsqrtf(acc / (float)C + 1e-8f);
        for (int c = lane; c < C; c += 64) y[r * C + c] = xr[c] * inv;
    }
}

__global__ void __launch_bounds__(256) latent_mix_kernel(const ga_latent_mix_desc d, const long total4) {
    const int D4 = d.D / 4;
    const int rep = d.rep > 1 ? d.rep : 1;
    const long row4 = (long)d.J * D4;                                  // quads per row
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long)gridDim.x * 256) {
        const int q = (int)(i % D4); const int j = (int)((i / D4) % d.J);
        const float a = d.alpha[j];
        if (!d.backward) {                                             // i runs over the R output rows
            const long r = i / row4;
            floatx4 c = ld4(d.codes + ((r / rep) * row4 + (i - r * row4)) * 4);
            if (d.avg) c += ld4(d.avg + ((size_t)j * d.D + 4 * q));
            *reinterpret_cast<floatx4*>(d.out + i * 4) = (1.0f - a) * c + a * ld4(d.styles + i * 4);
        } else {                                                       // i runs over the R / rep code rows
            const long r0 = i / row4, off = i - r0 * row4;
            floatx4 acc = {0.f, 0.f, 0.f, 0.f};
            for (int k = 0; k < rep; ++k) acc += ld4(d.dout + ((r0 * rep + k) * row4 + off) * 4);
            *reinterpret_cast<floatx4*>(d.dcodes + i * 4) = (1.0f - a) * acc;
        }
    }
}

__global__ void __launch_bounds__(256) pool_denorm_kernel(const ga_pool_denorm_desc d, const long total) {
    const int k = d.k, Wi = d.W * k;
    if (!d.backward) {
        const float inv = 0.5f / (float)(k * k);
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {   // i = pooled pixel
            const int w = (int)(i % d.W); long r = i / d.W;
            const int h = (int)(r % d.H); const long n = r / d.H;
            floatx4 acc = {0.f, 0.f, 0.f, 0.f};
            for (int a = 0; a < k; ++a)
                for (int b = 0; b < k; ++b)
                    acc += ld4(d.x + (((size_t)n * d.H * k + (h * k + a)) * Wi + (w * k + b)) * 4);
            float* o = d.y + ((((size_t)n * (d.H >> 1) + (h >> 1)) * (d.W >> 1) + (w >> 1)) * 4 + ((h & 1) * 2 + (w & 1))) * d.ld;
            floatx4 v = acc * inv + 0.5f;
            if (d.band > 0 && (h < d.band || h >= d.H - d.band)) v = floatx4{0.f, 0.f, 0.f, 0.f};      // image set to -1: 0.5 * -1 + 0.5
            v[3] = 0.f;
            *reinterpret_cast<floatx4*>(o) = v;
            for (int z = 4; z < d.ld; z += 4) *reinterpret_cast<floatx4*>(o + z) = floatx4{0.f, 0.f, 0.f, 0.f};
        }
    } else {
        const float inv = 0.5f / (float)(k * k);
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {   // i = generated pixel
            const int X = (int)(i % Wi); long r = i / Wi;
            const int Y = (int)(r % (d.H * k)); const long n = r / (d.H * k);
            const int h = Y / k, w = X / k;
            const float* g = d.dy + ((((size_t)n * (d.H >> 1) + (h >> 1)) * (d.W >> 1) + (w >> 1)) * 4 + ((h & 1) * 2 + (w & 1))) * d.ld;
            floatx4 v = ld4(g);
            if (d.dy_nchw) {                 // cotangent on the RETURNED purified image [N,3,H,W], added to the classifier's
                const size_t hw = (size_t)d.H * d.W, o = (size_t)n * 3 * hw + (size_t)h * d.W + w;
                v[0] += d.dy_nchw[o]; v[1] += d.dy_nchw[o + hw]; v[2] += d.dy_nchw[o + 2 * hw];
            }
            v *= inv;
            if (d.band > 0 && (h < d.band || h >= d.H - d.band)) v = floatx4{0.f, 0.f, 0.f, 0.f};
            v[3] = 0.f;
            *reinterpret_cast<floatx4*>(d.dx + i * 4) = v;
        }
    }
}

// backward of the tail with the demodulation-gradient reduction fused in: block = (row, 64-channel chunk, pixel segment),
// ql channel-quad lanes x 256/ql pixel lanes (the layout of rowchan_reduce_split_kernel); partial sums of dt * t go to
// ws[(n*S + seg)*C + c], added in segment order by rowchan_reduce_final_kernel
__global__ void __launch_bounds__(256) modout_bwd_reduce_kernel(const ga_modout_desc d, const int nchunks, const int ql, const int S,
                                                                const int seg_len) {
    __shared__ floatx4 part[256];
    const int tid = threadIdx.x, c4 = tid % ql, pl = tid / ql, PL = 256 / ql;
    int bi = blockIdx.x;
    const int seg = bi % S; bi /= S;
    const int chunk = bi % nchunks, n = bi / nchunks;
    const int c = chunk * 64 + 4 * c4;
    floatx4 acc = {0.f, 0.f, 0.f, 0.f};
    if (c < d.C) {
        floatx4 sc = {1.f, 1.f, 1.f, 1.f};
        if (d.scale) sc = ld4(d.scale + (size_t)n * d.C + c);
        const int p1 = min(d.P, (seg + 1) * seg_len);
        const int H2 = d.W > 0 ? (d.P / d.W) >> 1 : 0, W2 = d.W >> 1;
        for (int p = seg * seg_len + pl; p < p1; p += PL) {
            const size_t o = ((size_t)n * d.P + p) * d.C + c;
            const floatx4 t = ld4(modout_t_ptr(d, n, p, c));
            floatx4 u = sc * t;
            if (d.add) u += ld4(d.add + (size_t)p * d.C + c);
            const floatx4 g = ld4(d.dout + o);
            floatx4 dt;
#pragma unroll
            for (int e = 0; e < 4; ++e) dt[e] = g[e] * act_bwd(u[e], d.act) * sc[e];
            if (d.dt) *reinterpret_cast<floatx4*>(d.dt + o) = dt;
            if (d.dt_planes[0]) {
                const int h = p / d.W, w = p - h * d.W;
                float* plane = d.dt_planes[(h & 1) * 2 + (w & 1)];
                *reinterpret_cast<floatx4*>(plane + (((size_t)n * H2 + (h >> 1)) * W2 + (w >> 1)) * (d.ld_planes > 0 ? d.ld_planes : d.C) + c) = dt;
            }
            acc += dt * t;
        }
    }
    part[tid] = acc;
    __syncthreads();
    for (int st = PL >> 1; st > 0; st >>= 1) {
        if (pl < st) part[tid] += part[tid + st * ql];
        __syncthreads();
    }
    if (pl == 0 && c < d.C) *reinterpret_cast<floatx4*>(d.ws + ((size_t)n * S + seg) * d.C + c) = part[tid];
}

// ---------------------------------------------------------------------------------------------------------------
// nn.PReLU as its own pass (forward / backward)
// ---------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) prelu_kernel(const ga_prelu_desc d, const long total4) {
    const int C4 = d.C / 4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long)gridDim.x * 256) {
        const floatx4 a = ld4(d.slope + 4 * (int)(i % C4));
        const floatx4 x = ld4(d.x + i * 4);
        floatx4 r;
        if (!d.backward) {
#pragma unroll
            for (int e = 0; e < 4; ++e) r[e] = x[e] > 0.f ? x[e] : a[e] * x[e];
            *reinterpret_cast<floatx4*>(d.y + i * 4) = r;
        } else {
            const floatx4 g = ld4(d.dy + i * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) r[e] = g[e] * (x[e] > 0.f ? 1.f : a[e]);
            *reinterpret_cast<floatx4*>(d.dx + i * 4) = r;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// grouped convolution, few channels per group (ResNeXt conv2 and the sub-kernels of its transposes): one thread per
// output pixel x output-channel quad; the quads of a group read the same input channels (L1 broadcast), weights from L1/L2
// ---------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) gconv_kernel(const ga_gconv_desc d, const long total4) {
    const int C4 = d.C / 4, K = d.KH * d.KW * d.cg;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long)gridDim.x * 256) {
        const int q = (int)(i % C4); long p = i / C4;
        const int wo = (int)(p % d.Wo); p /= d.Wo;
        const int ho = (int)(p % d.Ho); const int n = (int)(p / d.Ho);
        const int co = 4 * q, g0 = (co / d.cg) * d.cg;                  // first input channel of this quad's group
        floatx4 acc = {0.f, 0.f, 0.f, 0.f};
        if (d.bias) acc = ld4(d.bias + co);
        const float* wrow = d.w + (size_t)co * K;
        for (int kh = 0; kh < d.KH; ++kh) {
            const int h = ho * d.stride - d.pad + kh;
            if (h < 0 || h >= d.Hi) continue;
            for (int kw = 0; kw < d.KW; ++kw) {
                const int w = wo * d.stride - d.pad + kw;
                if (w < 0 || w >= d.Wi) continue;
                const float* xp = d.x + (((size_t)n * d.Hi + h) * d.Wi + w) * d.C + g0;
                const float* wp = wrow + (kh * d.KW + kw) * d.cg;
                for (int c = 0; c < d.cg; c += 4) {
                    floatx4 v = ld4(xp + c);
                    if (d.pro_act) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = act_fwd_fast(v[e], d.pro_act);
                    }
#pragma unroll
                    for (int o = 0; o < 4; ++o) {
                        const floatx4 w4 = ld4(wp + (size_t)o * K + c);
                        acc[o] += v[0] * w4[0] + v[1] * w4[1] + v[2] * w4[2] + v[3] * w4[3];
                    }
                }
            }
        }
        const size_t o = (((size_t)n * d.Ho + ho) * d.Wo + wo) * d.C + co;
        if (d.dact_x) {
            const floatx4 u = ld4(d.dact_x + o);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] *= act_bwd_fast(u[e], d.dact_act);
        }
        *reinterpret_cast<floatx4*>(d.y + o) = acc;
    }
}

// one thread = one output pixel x ALL cg output channels of one group (grid.y), OB of them at a time; the group's weights sit in
// LDS and are read as wave-wide broadcasts, the pixel's input channels come from global memory (L1 after the first output block)
// once per tap and block: 4*OB FMAs per LDS read.  OB = 8 keeps the 32-wide groups of ResNeXt's layer 4 in registers (the
// one-block form held 32 accumulators plus an unrolled 8 x 32 weight window: 466 spilled registers); every output still sums
// its taps and channels in the same order.
template <int CG>
__global__ void __launch_bounds__(256) gconv_group_kernel(const ga_gconv_desc d, const long npix) {
    constexpr int OB = CG < 8 ? CG : 8;
    extern __shared__ __attribute__((aligned(16))) float gw[];          // [CG out][KH*KW][CG in]
    const int g = blockIdx.y, taps = d.KH * d.KW, K = taps * CG;
    for (int i = threadIdx.x * 4; i < CG * K; i += 256 * 4)
        *reinterpret_cast<floatx4*>(gw + i) = ld4(d.w + (size_t)g * CG * K + i);
    __syncthreads();
    const long p = (long)blockIdx.x * 256 + threadIdx.x;
    if (p >= npix) return;
    const int wo = (int)(p % d.Wo); long q = p / d.Wo;
    const int ho = (int)(q % d.Ho); const int n = (int)(q / d.Ho);
    const size_t ob = (size_t)p * d.C + g * CG;
#pragma unroll 1
    for (int o0 = 0; o0 < CG; o0 += OB) {
        float acc[OB];
#pragma unroll
        for (int o = 0; o < OB; ++o) acc[o] = d.bias ? d.bias[g * CG + o0 + o] : 0.f;
        for (int kh = 0; kh < d.KH; ++kh) {
            const int h = ho * d.stride - d.pad + kh;
            if (h < 0 || h >= d.Hi) continue;
            for (int kw = 0; kw < d.KW; ++kw) {
                const int w = wo * d.stride - d.pad + kw;
                if (w < 0 || w >= d.Wi) continue;
                const float* xp = d.x + (((size_t)n * d.Hi + h) * d.Wi + w) * d.C + g * CG;
                const float* wp = gw + (size_t)o0 * K + (kh * d.KW + kw) * CG;
#pragma unroll 2
                for (int c = 0; c < CG; c += 4) {
                    floatx4 v = ld4(xp + c);
                    if (d.pro_act) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = act_fwd_fast(v[e], d.pro_act);
                    }
#pragma unroll
                    for (int o = 0; o < OB; ++o) {
                        const floatx4 w4 = *reinterpret_cast<const floatx4*>(wp + o * K + c);
                        acc[o] += v[0] * w4[0] + v[1] * w4[1] + v[2] * w4[2] + v[3] * w4[3];
                    }
                }
            }
        }
#pragma unroll
        for (int o = 0; o < OB; o += 4) {
            floatx4 r = {acc[o], acc[o + 1], acc[o + 2], acc[o + 3]};
            if (d.dact_x) {
                const floatx4 u = ld4(d.dact_x + ob + o0 + o);
#pragma unroll
                for (int e = 0; e < 4; ++e) r[e] *= act_bwd_fast(u[e], d.dact_act);
            }
            *reinterpret_cast<floatx4*>(d.y + ob + o0 + o) = r;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// global average pool with an activation prologue (torchvision ResNet avgpool after the last ReLU)
// ---------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) avgpool_act_kernel(const ga_avgpool_act_desc d, const int nchunks) {
    __shared__ floatx4 part[16][16];
    const int tid = threadIdx.x, c4 = tid & 15, pl = tid >> 4;
    const int n = blockIdx.x / nchunks, chunk = blockIdx.x % nchunks;
    const int c = chunk * 64 + 4 * c4;
    const float inv = 1.0f / (float)d.P;
    if (!d.backward) {
        floatx4 acc = {0.f, 0.f, 0.f, 0.f};
        if (c < d.C)
            for (int p = pl; p < d.P; p += 16) {
                floatx4 v = ld4(d.x + ((size_t)n * d.P + p) * d.C + c);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[e] += act_fwd(v[e], d.act);
            }
        part[pl][c4] = acc;
        __syncthreads();
        if (pl == 0 && c < d.C) {
            floatx4 s = part[0][c4];
#pragma unroll
            for (int i = 1; i < 16; ++i) s += part[i][c4];
            *reinterpret_cast<floatx4*>(d.y + (size_t)n * d.C + c) = s * inv;
        }
    } else if (c < d.C) {
        const floatx4 g = ld4(d.dy + (size_t)n * d.C + c) * inv;
        for (int p = pl; p < d.P; p += 16) {
            const size_t o = ((size_t)n * d.P + p) * d.C + c;
            const floatx4 v = ld4(d.x + o);
            floatx4 r;
#pragma unroll
            for (int e = 0; e < 4; ++e) r[e] = g[e] * act_bwd(v[e], d.act);
            *reinterpret_cast<floatx4*>(d.dx + o) = r;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// image boundary: NCHW <-> NHWC, EoT repeat, input noise + clamp (abstract_models.py:129-143; wrappers.py:20)
// ---------------------------------------------------------------------------------------------------------------
// offset of pixel p = h*W + w of row n, channel 0, in the NHWC image (plain or space-to-depth)
__device__ __forceinline__ size_t image_px(const ga_image_io_desc& d, const int ld, const int HW, const int n, const int p) {
    if (!d.s2d) return ((size_t)n * HW + p) * ld;
    const int h = p / d.W, w = p - h * d.W;
    return ((((size_t)n * (d.H >> 1) + (h >> 1)) * (d.W >> 1) + (w >> 1)) * 4 + ((h & 1) * 2 + (w & 1))) * ld;
}

__global__ void __launch_bounds__(256) image_io_kernel(const ga_image_io_desc d, const long total) {
    const int HW = d.H * d.W;
    const int ld = d.ld > 0 ? d.ld : d.C;
    if (!d.backward) {
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
            // i indexes NCHW of the N rows (coalesced reads), writes NHWC
            const int p = (int)(i % HW); long q = i / HW;
            const int c = (int)(q % d.C); const int n = (int)(q / d.C);
            float v = d.x_nchw[((size_t)(n / d.rep) * d.C + c) * HW + p];
            if (d.noise_nchw) v += d.noise_nchw[i] * d.noise_coef[n];
            float* o = d.y_nhwc + image_px(d, ld, HW, n, p);
            o[c] = fminf(fmaxf(v, 0.f), 1.f);
            if (c == 0) for (int z = d.C; z < ld; ++z) o[z] = 0.f;      // pad channels of a wider pitch
        }
    } else {
        // K = cot_rep cotangents per forward row: output (image, k) sums the cotangent rows (image * rep + r) * K + k over r
        const int K = d.cot_rep > 1 ? d.cot_rep : 1;
        const long total_img = total / d.rep;
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total_img; i += (long)gridDim.x * 256) {
            const int p = (int)(i % HW); long q = i / HW;
            const int c = (int)(q % d.C); const int o = (int)(q / d.C);
            const int img = o / K, k = o - img * K;
            const float x = d.x_nchw[((size_t)img * d.C + c) * HW + p];
            float acc = 0.f;
            for (int r = 0; r < d.rep; ++r) {
                const int nf = img * d.rep + r;             // forward row
                float v = x;
                if (d.noise_nchw) v += d.noise_nchw[((size_t)nf * d.C + c) * HW + p] * d.noise_coef[nf];
                if (v >= 0.f && v <= 1.f) acc += d.dy_nhwc[image_px(d, ld, HW, nf * K + k, p) + c];
            }
            d.dx_nchw[i] = acc;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// sub-pixel assembly of a stride-2 transposed conv: y[n, 2i+a, 2j+b, :] = epilogue(s[a][b][n, i, j, :])
// ---------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) interleave2_kernel(const ga_interleave2_desc d, const long total4) {
    const int C4 = d.C / 4, Hh = d.H / 2, Wh = d.W / 2;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long)gridDim.x * 256) {
        const int q = (int)(i % C4); long p = i / C4;               // p = output pixel (n, h, w)
        const int w = (int)(p % d.W); long r = p / d.W;
        const int h = (int)(r % d.H); const long n = r / d.H;
        const float* src = d.s[h & 1][w & 1];
        floatx4 v = {0.f, 0.f, 0.f, 0.f};
        if (src) v = *reinterpret_cast<const floatx4*>(src + (((size_t)n * Hh + (h >> 1)) * Wh + (w >> 1)) * (d.lds > 0 ? d.lds : d.C) + 4 * q);
        const size_t o = (size_t)p * d.C + 4 * q;
        if (d.dact_x) {
            // K cotangents per forward row (dact_rep = K): the act' source is the forward's tensor, row n / K
            const size_t od = d.dact_rep > 1 ? ((((size_t)(n / d.dact_rep) * d.H + h) * d.W + w) * d.C + 4 * q) : o;
            floatx4 u = *reinterpret_cast<const floatx4*>(d.dact_x + od);
            floatx4 ds = {1.f, 1.f, 1.f, 1.f};
            if (d.dact_prelu) {
                ds = *reinterpret_cast<const floatx4*>(d.dact_scale + 4 * q);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] *= u[e] > 0.f ? 1.f : ds[e];
            } else {
                if (d.dact_scale) {
                    ds = *reinterpret_cast<const floatx4*>(d.dact_scale + 4 * q);
                    u = u * ds + *reinterpret_cast<const floatx4*>(d.dact_shift + 4 * q);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] *= act_bwd_fast(u[e], d.dact_act) * ds[e];
            }
        }
        if (d.addend) v += *reinterpret_cast<const floatx4*>(d.addend + o);
        if (d.addend2) v += *reinterpret_cast<const floatx4*>(d.addend2 + o);
        *reinterpret_cast<floatx4*>(d.y + o) = v;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// separable Gaussian blur, reflect border (kornia gaussian_blur2d semantics), one image plane per workgroup in LDS
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int reflect_idx(int t, const int n) {       // 'reflect' (no edge repeat): -1 -> 1, n -> n-2
    if (t < 0) t = -t;
    if (t >= n) t = 2 * (n - 1) - t;
    return t;
}

// forward line filter: out[i] = sum_k g[k] in[reflect(i + k - p)];  adjoint: out[j] = sum over (i,k) with reflect(i+k-p) = j
// taps q in [p - r, p + r] only (r = the caller's `radius`, or k/2): a sigma-1 Gaussian of 255 taps (256-px images) has 25 taps
// above 1e-31 of its peak
__device__ __forceinline__ float blur_tap_sum(const float* line, const int stride, const int n, const int i, const float* g,
                                              const int k, const bool adjoint, const int r) {
    const int p = k / 2;
    float acc = 0.f;
    if (!adjoint) {
        for (int q = p - r; q <= p + r; ++q) acc += g[q] * line[reflect_idx(i + q - p, n) * stride];
    } else {
        // sources t that reflect onto j = i: t = j, t = -j (j >= 1), t = 2(n-1) - j (j <= n-2); t = i' + q - p
        for (int q = p - r; q <= p + r; ++q) {
            const int a = i - q + p;
            if (a >= 0 && a < n) acc += g[q] * line[a * stride];
            if (i >= 1) { const int b2 = -i - q + p; if (b2 >= 0 && b2 < n) acc += g[q] * line[b2 * stride]; }
            if (i <= n - 2) { const int c2 = 2 * (n - 1) - i - q + p; if (c2 >= 0 && c2 < n) acc += g[q] * line[c2 * stride]; }
        }
    }
    return acc;
}

// whole plane in LDS (planes up to ~90 x 90: the 64-px images of the ids experiment)
__global__ void __launch_bounds__(256) gauss_blur_kernel(const ga_blur_desc d, const int r) {
    extern __shared__ float bl_s[];
    float* a = bl_s;                       // [H][W]
    float* b = bl_s + d.H * d.W;           // [H][W]
    float* g = b + d.H * d.W;              // [k]
    const int HW = d.H * d.W;
    const float* src = d.x + (size_t)blockIdx.x * HW;
    for (int i = threadIdx.x; i < HW; i += 256) a[i] = src[i];
    for (int i = threadIdx.x; i < d.k; i += 256) g[i] = d.taps[i];
    __syncthreads();
    // kornia filters along x (kernel 1 x k) then along y; the adjoint applies the transposed passes in reverse order
    const bool adj = d.backward != 0;
    for (int i = threadIdx.x; i < HW; i += 256) {
        const int h = i / d.W, w = i - h * d.W;
        b[i] = adj ? blur_tap_sum(a + w, d.W, d.H, h, g, d.k, true, r) : blur_tap_sum(a + h * d.W, 1, d.W, w, g, d.k, false, r);
    }
    __syncthreads();
    float* dst = d.y + (size_t)blockIdx.x * HW;
    for (int i = threadIdx.x; i < HW; i += 256) {
        const int h = i / d.W, w = i - h * d.W;
        dst[i] = adj ? blur_tap_sum(b + h * d.W, 1, d.W, w, g, d.k, true, r) : blur_tap_sum(b + w, d.W, d.H, h, g, d.k, false, r);
    }
}

// one separable pass over larger planes (128 / 256-px images of the cars / gender experiments), through a caller-owned
// intermediate plane set: along x a workgroup stages RT whole rows, along y a strip of CT columns x all rows (128-B segments of
// consecutive columns: coalesced); every output reads its 2r + 1 taps from LDS
__global__ void __launch_bounds__(256) gauss_blur_pass_kernel(const float* __restrict__ src, float* __restrict__ dst, const float* __restrict__ taps,
                                                              const int H, const int W, const int k, const int r, const int along_y,
                                                              const int adjoint, const int T, const int tiles) {
    extern __shared__ float bl_s[];
    const int plane = blockIdx.x / tiles, tile = blockIdx.x - plane * tiles;
    const size_t base = (size_t)plane * H * W;
    float* g = bl_s;                       // [k]
    float* a = bl_s + ((k + 3) & ~3);
    for (int i = threadIdx.x; i < k; i += 256) g[i] = taps[i];
    if (!along_y) {                        // rows tile*T .. +T, each W long
        const int h0 = tile * T, nr = min(T, H - h0);
        for (int i = threadIdx.x; i < nr * W; i += 256) a[i] = src[base + (size_t)h0 * W + i];
        __syncthreads();
        for (int i = threadIdx.x; i < nr * W; i += 256) {
            const int hh = i / W, w = i - hh * W;
            dst[base + (size_t)h0 * W + i] = blur_tap_sum(a + hh * W, 1, W, w, g, k, adjoint != 0, r);
        }
    } else {                               // columns tile*T .. +T, each H long, stored [H][T]
        const int w0 = tile * T, nc = min(T, W - w0);
        for (int i = threadIdx.x; i < H * T; i += 256) {
            const int h = i / T, c = i - h * T;
            if (c < nc) a[i] = src[base + (size_t)h * W + w0 + c];
        }
        __syncthreads();
        for (int i = threadIdx.x; i < H * T; i += 256) {
            const int h = i / T, c = i - h * T;
            if (c < nc) dst[base + (size_t)h * W + w0 + c] = blur_tap_sum(a + c, T, H, h, g, k, adjoint != 0, r);
        }
    }
}

__global__ void __launch_bounds__(256) rep_sum_kernel(const float* x, float* y, const long total, const long inner, const int rep,
                                                        const int accumulate) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long b = i / inner, j = i - b * inner;
        float acc = accumulate ? y[i] : 0.f;
        for (int r = 0; r < rep; ++r) acc += x[(b * rep + r) * inner + j];
        y[i] = acc;
    }
}

__global__ void __launch_bounds__(256) axpby_kernel(const float* x, float* y, const long n, const float a, const float b) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256)
        y[i] = a * x[i] + (b == 0.f ? 0.f : b * y[i]);
}

static inline unsigned grid_for(long items) {
    long b = (items + 255) / 256;
    if (b > 2048 * 4) b = 2048 * 4;
    if (b < 1) b = 1;
    return (unsigned)b;
}

}  // namespace ga

using namespace ga;

extern "C" int ga_rowchan_reduce(const ga_rowchan_reduce_desc* d, void* s) {
    ga::clear_stale_error();
    if (!d || (!d->a && !(d->a_src && d->a_w)) || !d->out || d->N <= 0 || d->P <= 0 || d->C <= 0) return GA_E_BADARG;
    if (d->C % 4) return GA_E_UNSUPPORTED;
    if ((d->a && !aligned16(d->a)) || !aligned16(d->out) || (d->b && !aligned16(d->b))) return GA_E_ALIGN;
    if (!d->a && (!aligned16(d->a_src) || !aligned16(d->a_w))) return GA_E_ALIGN;
    if (d->scaled && (!d->gate || !aligned16(d->scaled) || !aligned16(d->gate) || (d->skip && !aligned16(d->skip)))) return GA_E_BADARG;
    const int nchunks = (d->C + 63) / 64;
    if (d->ws && d->P >= 4096 && d->N * nchunks < 1024) {
        if (!aligned16(d->ws)) return GA_E_ALIGN;
        long S = d->ws_floats / ((long)d->N * d->C);
        const long want = 2048 / ((long)d->N * nchunks);                 // ~8 workgroups per CU
        if (S > want) S = want;
        if (S > (d->P + 1023) / 1024) S = (d->P + 1023) / 1024;          // >= 1024 pixels per segment
        if (S >= 2) {
            const int q = d->C >= 64 ? 16 : d->C / 4;
            int ql = 1;
            while (ql < q) ql <<= 1;
            const int seg_len = (int)((d->P + S - 1) / S);
            hipLaunchKernelGGL(rowchan_reduce_split_kernel, dim3((unsigned)(d->N * nchunks * S)), dim3(256), 0, (hipStream_t)s, *d,
                               nchunks, ql, (int)S, seg_len);
            const long total = (long)d->N * d->C;
            hipLaunchKernelGGL(rowchan_reduce_final_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)s, *d, (int)S);
            return check_launch();
        }
    }
    hipLaunchKernelGGL(rowchan_reduce_kernel, dim3(d->N * nchunks), dim3(256), 0, (hipStream_t)s, *d, nchunks);
    return check_launch();
}

extern "C" int ga_se_excite(const ga_se_excite_desc* d, void* s) {
    ga::clear_stale_error();
    if (!d || !d->w1 || !d->b1 || !d->w2 || !d->b2 || !d->hid || !d->gate || d->N <= 0 || d->C <= 0 || d->Hd <= 0) return GA_E_BADARG;
    if (!d->backward && !d->m && !d->t) return GA_E_BADARG;
    if (d->backward && ((!d->dgate && !d->t) || !d->pro_scale || !d->pro_shift || d->P <= 0)) return GA_E_BADARG;
    if (d->out && (d->backward || !d->t || !aligned16(d->out) || (d->skip && !aligned16(d->skip)))) return GA_E_BADARG;
    if (d->act_rep > 1 && (!d->backward || !d->t || d->N % d->act_rep)) return GA_E_BADARG;
    size_t lds = (size_t)(d->C + d->Hd + 4 + 256) * sizeof(float);     // + [Hd][chunks] partials of the FC phases
    if (d->Hd > 256) return GA_E_UNSUPPORTED;
    if (d->t) {
        if (d->P <= 0 || (d->backward && !d->dout)) return GA_E_BADARG;
        if (d->C % 4 || d->C > 1024) return GA_E_UNSUPPORTED;
        if (!aligned16(d->t) || (d->dout && !aligned16(d->dout))) return GA_E_ALIGN;
        const size_t red = (size_t)(256 / (d->C / 4)) * d->C * sizeof(float);
        if (red > 256 * sizeof(float)) lds += red - 256 * sizeof(float);
    }
    if (lds > 64 * 1024) return GA_E_UNSUPPORTED;
    hipLaunchKernelGGL(se_excite_kernel, dim3(d->N), dim3(256), lds, (hipStream_t)s, *d);
    return check_launch();
}

extern "C" int ga_se_apply(const ga_se_apply_desc* d, void* s) {
    ga::clear_stale_error();
    if (!d || !d->t || !d->gate || !d->out || d->N <= 0 || d->H <= 0 || d->W <= 0 || d->C <= 0) return GA_E_BADARG;
    if (!d->skip && d->skip_mode != 0) return GA_E_BADARG;
    if (d->C % 4) return GA_E_UNSUPPORTED;
    if (d->skip_mode == 1 && ((d->H | d->W) & 1)) return GA_E_BADARG;
    if (d->skip_mode < 0 || d->skip_mode > 2) return GA_E_UNSUPPORTED;
    if ((d->skip && !aligned16(d->skip)) || !aligned16(d->t) || !aligned16(d->gate) || !aligned16(d->out)) return GA_E_ALIGN;
    const long total4 = (long)d->N * d->H * d->W * (d->C / 4);
    hipLaunchKernelGGL(se_apply_kernel, dim3(grid_for(total4)), dim3(256), 0, (hipStream_t)s, *d, total4);
    return check_launch();
}

extern "C" int ga_bilinear_up2_bwd(const ga_bilinear_up2_bwd_desc* d, void* s) {
    ga::clear_stale_error();
    if (!d || !d->dhigh || !d->dlow || d->N <= 0 || d->h <= 0 || d->w <= 0 || d->C <= 0) return GA_E_BADARG;
    if (d->C % 4) return GA_E_UNSUPPORTED;
    if (!aligned16(d->dhigh) || !aligned16(d->dlow)) return GA_E_ALIGN;
    const long total4 = (long)d->N * d->h * d->w * (d->C / 4);
    hipLaunchKernelGGL(bilinear_up2_bwd_kernel, dim3(grid_for(total4)), dim3(256), 0, (hipStream_t)s, *d, total4);
    return check_launch();
}

extern "C" int ga_sampler_mix(const ga_sampler_desc* d, void* s) {
    ga::clear_stale_error();
    if (!d || !d->mu_q || !d->eps || d->N <= 0 || d->h <= 0 || d->w <= 0 || d->NL <= 0) return GA_E_BADARG;
    if (d->ldq < d->NL || (d->p && d->ldp < 2 * d->NL)) return GA_E_BADARG;
    if (!d->backward && !d->z) return GA_E_BADARG;
    if (d->backward && (!d->dz || (d->q_rep > 1 ? !d->dmu_q_rows : !d->dmu_q) || (d->p && !d->dp))) return GA_E_BADARG;
    if (d->q_rep > 1 && d->N % d->q_rep) return GA_E_BADARG;
    if (d->act_rep > 1 && (!d->backward || d->N % d->act_rep || (d->q_rep > 1 && (d->N / d->act_rep) % d->q_rep))) return GA_E_BADARG;
    if (d->mode == 1 && (!d->p || d->ldq < 2 * d->NL || d->q_rep > 1 || (d->backward && (!d->dmu_q || !d->dp)))) return GA_E_BADARG;
    if (d->mode != 0 && d->mode != 1) return GA_E_UNSUPPORTED;
    const long total = (long)d->N * d->h * d->w * d->NL;
    hipLaunchKernelGGL(sampler_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)s, *d, total);
    return check_launch();
}

extern "C" int ga_dml_mean(const ga_dml_desc* d, void* s) {
    ga::clear_stale_error();
    if (!d || !d->logits || d->N <= 0 || d->H <= 0 || d->W <= 0 || d->nmix <= 0) return GA_E_BADARG;
    if (d->nmix > DML_MAXMIX) return GA_E_UNSUPPORTED;
    if (d->ld < d->nmix * 10 || (d->ld_img != 0 && d->ld_img < 3)) return GA_E_BADARG;
    if (!d->backward && !d->img_nchw && !d->img_nhwc) return GA_E_BADARG;
    if (d->backward && (!d->dlogits || (!d->dimg_nhwc && !d->dimg_nchw))) return GA_E_BADARG;
    if (d->act_rep > 1 && (!d->backward || d->N % d->act_rep)) return GA_E_BADARG;
    const long npix = (long)d->N * d->H * d->W;
    const size_t lds = (size_t)DML_ROWS * (d->ld + 1) * sizeof(float);
    if (lds > 64 * 1024) return GA_E_UNSUPPORTED;
    long blocks = (npix + DML_ROWS - 1) / DML_ROWS;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(dml_kernel, dim3((unsigned)blocks), dim3(DML_ROWS), lds, (hipStream_t)s, *d, npix);
    return check_launch();
}

extern "C" int ga_maxpool2(const ga_maxpool2_desc* d, void* s) {
    ga::clear_stale_error();
    if (!d || !d->x || d->N <= 0 || d->H <= 0 || d->W <= 0 || d->C <= 0) return GA_E_BADARG;
    if ((d->H | d->W) & 1) return GA_E_UNSUPPORTED;
    if (d->C % 4) return GA_E_UNSUPPORTED;
    if (!d->backward && !d->y) return GA_E_BADARG;
    if (d->backward && (!d->dy || !d->dx)) return GA_E_BADARG;
    if (d->act_rep > 1 && (!d->backward || d->N % d->act_rep)) return GA_E_BADARG;
    const long total4 = (long)d->N * (d->H / 2) * (d->W / 2) * (d->C / 4);
    hipLaunchKernelGGL(maxpool2_kernel, dim3(grid_for(total4)), dim3(256), 0, (hipStream_t)s, *d, total4);
    return check_launch();
}

extern "C" int ga_maxpool3s2(const ga_maxpool3s2_desc* d, void* s) {
    ga::clear_stale_error();
    if (!d || !d->x || d->N <= 0 || d->H <= 0 || d->W <= 0 || d->C <= 0) return GA_E_BADARG;
    if (((d->H | d->W) & 1) || (d->C % 4)) return GA_E_UNSUPPORTED;
    if (!d->backward && !d->y) return GA_E_BADARG;
    if (d->backward && (!d->dy || !d->dx)) return GA_E_BADARG;
    const long total4 = d->backward ? (long)d->N * d->H * d->W * (d->C / 4) : (long)d->N * (d->H / 2) * (d->W / 2) * (d->C / 4);
    hipLaunchKernelGGL(maxpool3s2_kernel, dim3(grid_for(total4)), dim3(256), 0, (hipStream_t)s, *d, total4);
    return check_launch();
}

extern "C" int ga_unary(const ga_unary_desc* d, void* s) {
    ga::clear_stale_error();
    if (!d || !d->x || !d->y || d->n <= 0 || d->mode < 0 || d->mode > 3) return GA_E_BADARG;
    if ((d->mode == 1 || d->mode == 3) && !d->g) return GA_E_BADARG;
    hipLaunchKernelGGL(unary_kernel, dim3(grid_for(d->n)), dim3(256), 0, (hipStream_t)s, *d);
    return check_launch();
}

extern "C" int ga_modout(const ga_modout_desc* d, void* s) {
    ga::clear_stale_error();
    if (!d || (!d->t && !d->t_planes[0]) || d->N <= 0 || d->P <= 0 || d->C <= 0) return GA_E_BADARG;
    if (d->C % 4) return GA_E_UNSUPPORTED;
    if (!d->backward && !d->out) return GA_E_BADARG;
    if (d->backward && (!d->dout || (!d->dt && !d->dt_planes[0]))) return GA_E_BADARG;
    if ((d->backward && d->dt_planes[0]) || d->t_planes[0]) {
        if (d->backward && d->dt_planes[0] && (!d->dt_planes[1] || !d->dt_planes[2] || !d->dt_planes[3])) return GA_E_BADARG;
        if (d->t_planes[0] && (!d->t_planes[1] || !d->t_planes[2] || !d->t_planes[3])) return GA_E_BADARG;
        if (d->W <= 0 || d->W % 2 || d->P % d->W || (d->P / d->W) % 2) return GA_E_BADARG;
        if (d->ld_planes % 4) return GA_E_UNSUPPORTED;
        for (int i = 0; i < 4; ++i)
            if ((d->t_planes[0] && !aligned16(d->t_planes[i])) || (d->backward && d->dt_planes[0] && !aligned16(d->dt_planes[i]))) return GA_E_ALIGN;
    }
    if (d->backward && d->red) {
        if (!d->ws || d->ws_floats < (long)d->N * d->C) return GA_E_BADARG;
        if (!aligned16(d->ws) || (d->t && !aligned16(d->t)) || !aligned16(d->dout) || (d->dt && !aligned16(d->dt))) return GA_E_ALIGN;
        const int nchunks = (d->C + 63) / 64;
        long S = d->ws_floats / ((long)d->N * d->C);
        const long want = (2048 + (long)d->N * nchunks - 1) / ((long)d->N * nchunks);     // ~8 workgroups per CU
        if (S > want) S = want;
        if (S > (d->P + 255) / 256) S = (d->P + 255) / 256;                               // >= 256 pixels per segment
        if (S < 1) S = 1;
        const int q = d->C >= 64 ? 16 : d->C / 4;
        int ql = 1;
        while (ql < q) ql <<= 1;
        const int seg_len = (int)((d->P + S - 1) / S);
        hipLaunchKernelGGL(modout_bwd_reduce_kernel, dim3((unsigned)(d->N * nchunks * S)), dim3(256), 0, (hipStream_t)s, *d, nchunks, ql,
                           (int)S, seg_len);
        ga_rowchan_reduce_desc r = {};
        r.out = d->red; r.N = d->N; r.P = d->P; r.C = d->C; r.scale = 1.0f; r.ws = d->ws; r.ws_floats = d->ws_floats;
        const long total = (long)d->N * d->C;
        hipLaunchKernelGGL(rowchan_reduce_final_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)s, r, (int)S);
        return check_launch();
    }
    const long total4 = (long)d->N * d->P * (d->C / 4);
    hipLaunchKernelGGL(modout_kernel, dim3(grid_for(total4)), dim3(256), 0, (hipStream_t)s, *d, total4);
    return check_launch();
}

extern "C" int ga_up2_blur(const ga_up2_blur_desc* d, void* s) {
    ga::clear_stale_error();
    if (!d || d->N <= 0 || d->H <= 0 || d->W <= 0 || d->C <= 0) return GA_E_BADARG;
    if (d->C % 4) return GA_E_UNSUPPORTED;
    if (!d->backward && (!d->lo_in || !d->hi)) return GA_E_BADARG;
    if (d->backward && (!d->hi_in || !d->lo)) return GA_E_BADARG;
    const long total4 = (long)d->N * d->H * d->W * (d->C / 4) * (d->backward ? 1 : 4);
    hipLaunchKernelGGL(up2_blur_kernel, dim3(grid_for(total4)), dim3(256), 0, (hipStream_t)s, *d, total4);
    return check_launch();
}

extern "C" int ga_pixelnorm(const float* x, float* y, long rows, int C, void* s) {
    ga::clear_stale_error();
    if (!x || !y || rows <= 0 || C <= 0) return GA_E_BADARG;
    const long blocks = (rows + 3) / 4;
    hipLaunchKernelGGL(pixelnorm_kernel, dim3((unsigned)(blocks > 65535 ? 65535 : blocks)), dim3(256), 0, (hipStream_t)s, x, y, rows, C);
    return check_launch();
}

extern "C" int ga_latent_mix(const ga_latent_mix_desc* d, void* s) {
    ga::clear_stale_error();
    if (!d || !d->alpha || d->R <= 0 || d->J <= 0 || d->D <= 0) return GA_E_BADARG;
    if (d->D % 4) return GA_E_UNSUPPORTED;
    if (!d->backward && (!d->codes || !d->styles || !d->out)) return GA_E_BADARG;
    if (d->backward && (!d->dout || !d->dcodes)) return GA_E_BADARG;
    if (d->rep > 1 && d->R % d->rep) return GA_E_BADARG;
    const long total4 = (long)(d->backward && d->rep > 1 ? d->R / d->rep : d->R) * d->J * (d->D / 4);
    hipLaunchKernelGGL(latent_mix_kernel, dim3(grid_for(total4)), dim3(256), 0, (hipStream_t)s, *d, total4);
    return check_launch();
}

extern "C" int ga_pool_denorm(const ga_pool_denorm_desc* d, void* s) {
    ga::clear_stale_error();
    if (!d || d->N <= 0 || d->H <= 0 || d->W <= 0 || d->k <= 0 || d->ld < 4) return GA_E_BADARG;
    if ((d->H | d->W) & 1 || d->ld % 4) return GA_E_UNSUPPORTED;
    if (!d->backward && (!d->x || !d->y)) return GA_E_BADARG;
    if (d->backward && (!d->dy || !d->dx)) return GA_E_BADARG;
    const long total = (long)d->N * d->H * d->W * (d->backward ? d->k * d->k : 1);
    hipLaunchKernelGGL(pool_denorm_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)s, *d, total);
    return check_launch();
}

extern "C" int ga_prelu(const ga_prelu_desc* d, void* s) {
    ga::clear_stale_error();
    if (!d || !d->x || !d->slope || d->rows <= 0 || d->C <= 0) return GA_E_BADARG;
    if (d->C % 4) return GA_E_UNSUPPORTED;
    if (!d->backward && !d->y) return GA_E_BADARG;
    if (d->backward && (!d->dy || !d->dx)) return GA_E_BADARG;
    const long total4 = d->rows * (d->C / 4);
    hipLaunchKernelGGL(prelu_kernel, dim3(grid_for(total4)), dim3(256), 0, (hipStream_t)s, *d, total4);
    return check_launch();
}

extern "C" int ga_gconv(const ga_gconv_desc* d, void* s) {
    ga::clear_stale_error();
    if (!d || !d->x || !d->w || !d->y || d->N <= 0 || d->Hi <= 0 || d->Wi <= 0 || d->Ho <= 0 || d->Wo <= 0) return GA_E_BADARG;
    if (d->C <= 0 || d->cg <= 0 || d->C % d->cg || d->KH <= 0 || d->KW <= 0 || d->stride <= 0 || d->pad < 0) return GA_E_BADARG;
    if (d->cg % 4) return GA_E_UNSUPPORTED;
    if (!aligned16(d->x) || !aligned16(d->w) || !aligned16(d->y) || (d->bias && !aligned16(d->bias)) ||
        (d->dact_x && !aligned16(d->dact_x))) return GA_E_ALIGN;
    const long npix = (long)d->N * d->Ho * d->Wo;
    const size_t lds = (size_t)d->cg * d->KH * d->KW * d->cg * sizeof(float);
    const dim3 grid((unsigned)((npix + 255) / 256), (unsigned)(d->C / d->cg));
    if (npix < 0x7fffffffL * 256 && lds <= 64 * 1024) {         // one pixel x one whole group per thread, weights in LDS
        switch (d->cg) {
            case 4:  hipLaunchKernelGGL(gconv_group_kernel<4>, grid, dim3(256), lds, (hipStream_t)s, *d, npix); return check_launch();
            case 8:  hipLaunchKernelGGL(gconv_group_kernel<8>, grid, dim3(256), lds, (hipStream_t)s, *d, npix); return check_launch();
            case 16: hipLaunchKernelGGL(gconv_group_kernel<16>, grid, dim3(256), lds, (hipStream_t)s, *d, npix); return check_launch();
            case 32: hipLaunchKernelGGL(gconv_group_kernel<32>, grid, dim3(256), lds, (hipStream_t)s, *d, npix); return check_launch();
            default: break;
        }
    }
    const long total4 = npix * (d->C / 4);
    hipLaunchKernelGGL(gconv_kernel, dim3(grid_for(total4)), dim3(256), 0, (hipStream_t)s, *d, total4);
    return check_launch();
}

extern "C" int ga_avgpool_act(const ga_avgpool_act_desc* d, void* s) {
    ga::clear_stale_error();
    if (!d || !d->x || d->N <= 0 || d->P <= 0 || d->C <= 0) return GA_E_BADARG;
    if (d->C % 4) return GA_E_UNSUPPORTED;
    if (!d->backward && !d->y) return GA_E_BADARG;
    if (d->backward && (!d->dy || !d->dx)) return GA_E_BADARG;
    const int nchunks = (d->C + 63) / 64;
    hipLaunchKernelGGL(avgpool_act_kernel, dim3(d->N * nchunks), dim3(256), 0, (hipStream_t)s, *d, nchunks);
    return check_launch();
}

extern "C" int ga_image_io(const ga_image_io_desc* d, void* s) {
    ga::clear_stale_error();
    if (!d || !d->x_nchw || d->N <= 0 || d->C <= 0 || d->H <= 0 || d->W <= 0 || d->rep <= 0 || d->N % d->rep) return GA_E_BADARG;
    if ((d->noise_nchw == nullptr) != (d->noise_coef == nullptr)) return GA_E_BADARG;
    if (!d->backward && !d->y_nhwc) return GA_E_BADARG;
    if (d->backward && (!d->dy_nhwc || !d->dx_nchw)) return GA_E_BADARG;
    if (d->ld != 0 && d->ld < d->C) return GA_E_BADARG;
    if (d->s2d && ((d->H | d->W) & 1)) return GA_E_BADARG;
    if (d->cot_rep > 1 && (!d->backward || d->N % (d->rep * d->cot_rep))) return GA_E_BADARG;
    const long total = (long)d->N * d->C * d->H * d->W;
    hipLaunchKernelGGL(image_io_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)s, *d, total);
    return check_launch();
}

extern "C" int ga_gauss_blur(const ga_blur_desc* d, void* s) {
    ga::clear_stale_error();
    if (!d || !d->x || !d->y || !d->taps || d->planes <= 0 || d->H <= 0 || d->W <= 0 || d->k <= 0) return GA_E_BADARG;
    if (!(d->k & 1) || d->k / 2 >= d->H || d->k / 2 >= d->W || d->radius < 0) return GA_E_UNSUPPORTED;
    const int r = (d->radius > 0 && d->radius < d->k / 2) ? d->radius : d->k / 2;
    const size_t lds = ((size_t)2 * d->H * d->W + d->k) * sizeof(float);
    if (lds <= 64 * 1024) {                                  // the whole plane (both passes) in LDS
        hipLaunchKernelGGL(gauss_blur_kernel, dim3(d->planes), dim3(256), lds, (hipStream_t)s, *d, r);
        return check_launch();
    }
    if (!d->tmp || d->tmp == d->x || d->tmp == d->y) return GA_E_BADARG;      // two passes through the caller's intermediate planes
    if (d->H > 4096 || d->W > 4096) return GA_E_UNSUPPORTED;
    const int RT = 8;
    int CT = 32;
    while (CT > 4 && (size_t)d->H * CT * sizeof(float) > 48 * 1024) CT >>= 1;
    const size_t kpad = ((size_t)d->k + 3) & ~(size_t)3;
    const size_t lds_x = (kpad + (size_t)RT * d->W) * sizeof(float), lds_y = (kpad + (size_t)d->H * CT) * sizeof(float);
    if (lds_x > 64 * 1024 || lds_y > 64 * 1024) return GA_E_UNSUPPORTED;
    const int tx = (d->H + RT - 1) / RT, ty = (d->W + CT - 1) / CT;
    // forward: along x, then along y (kornia); adjoint: the transposed passes in reverse order
    if (!d->backward) {
        hipLaunchKernelGGL(gauss_blur_pass_kernel, dim3(d->planes * tx), dim3(256), lds_x, (hipStream_t)s, d->x, d->tmp, d->taps, d->H, d->W, d->k, r, 0, 0, RT, tx);
        hipLaunchKernelGGL(gauss_blur_pass_kernel, dim3(d->planes * ty), dim3(256), lds_y, (hipStream_t)s, (const float*)d->tmp, d->y, d->taps, d->H, d->W, d->k, r, 1, 0, CT, ty);
    } else {
        hipLaunchKernelGGL(gauss_blur_pass_kernel, dim3(d->planes * ty), dim3(256), lds_y, (hipStream_t)s, d->x, d->tmp, d->taps, d->H, d->W, d->k, r, 1, 1, CT, ty);
        hipLaunchKernelGGL(gauss_blur_pass_kernel, dim3(d->planes * tx), dim3(256), lds_x, (hipStream_t)s, (const float*)d->tmp, d->y, d->taps, d->H, d->W, d->k, r, 0, 1, RT, tx);
    }
    return check_launch();
}

extern "C" int ga_interleave2(const ga_interleave2_desc* d, void* s) {
    ga::clear_stale_error();
    if (!d || !d->y || d->N <= 0 || d->H <= 0 || d->W <= 0 || d->C <= 0) return GA_E_BADARG;
    if (((d->H | d->W) & 1) || (d->C % 4)) return GA_E_UNSUPPORTED;
    if ((d->dact_scale == nullptr) != (d->dact_shift == nullptr) && !d->dact_prelu) return GA_E_BADARG;
    if (d->dact_prelu && (!d->dact_x || !d->dact_scale)) return GA_E_BADARG;
    const long total4 = (long)d->N * d->H * d->W * (d->C / 4);
    hipLaunchKernelGGL(interleave2_kernel, dim3(grid_for(total4)), dim3(256), 0, (hipStream_t)s, *d, total4);
    return check_launch();
}

extern "C" int ga_rep_sum(const float* x, float* y, long rows, long inner, int rep, int accumulate, void* s) {
    ga::clear_stale_error();
    if (!x || !y || rows <= 0 || inner <= 0 || rep <= 0 || rows % rep) return GA_E_BADARG;
    const long total = rows / rep * inner;
    hipLaunchKernelGGL(rep_sum_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)s, x, y, total, inner, rep, accumulate);
    return check_launch();
}

extern "C" int ga_axpby(const float* x, float* y, long n, float alpha, float beta, void* s) {
    ga::clear_stale_error();
    if (!x || !y || n <= 0) return GA_E_BADARG;
    hipLaunchKernelGGL(axpby_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)s, x, y, n, alpha, beta);
    return check_launch();
}
