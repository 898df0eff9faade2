// Shared epilogue of the implicit-GEMM convolution kernels (fp32-MFMA and split-bf16 variants): the accumulator
// tile is transposed through LDS so that each lane owns 4 consecutive channels of one pixel, then bias / act' /
// addends / store run as 16-B accesses on contiguous rows; scalar fallback for channel counts not multiple of 4.
#pragma once
#include "ga_common.h"

namespace ga {

// row of dact_x for output pixel m: with K cotangents per saved activation (dact_rep = K > 1) output row n reads row n / K
__device__ __forceinline__ size_t dact_row(const ga_conv_desc& d, const size_t m, const int HoWo) {
    if (d.dact_rep <= 1) return m;
    const size_t n = m / (size_t)HoWo;
    return (n / (size_t)d.dact_rep) * (size_t)HoWo + (m - n * (size_t)HoWo);
}

// epilogue on 4 consecutive channels of one output pixel (vector form)
__device__ __forceinline__ void epilogue4(const ga_conv_desc& d, const int m, const int co, floatx4 v, const int HoWo) {
    if (d.bias) v += *reinterpret_cast<const floatx4*>(d.bias + co);
    if ((d.flags & GA_CONV_ADDEND_PRE_DACT) && d.addend) v += *reinterpret_cast<const floatx4*>(d.addend + (size_t)m * d.ldadd + co);
    if (d.dact_x) {
        floatx4 u = *reinterpret_cast<const floatx4*>(d.dact_x + dact_row(d, (size_t)m, HoWo) * d.lddact + co);
        floatx4 ds = {1.f, 1.f, 1.f, 1.f};
        if (d.flags & GA_CONV_DACT_PRELU) {
            ds = *reinterpret_cast<const floatx4*>(d.dact_scale + co);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] *= u[e] > 0.f ? 1.f : ds[e];
        } else {
            if (d.dact_scale) {
                ds = *reinterpret_cast<const floatx4*>(d.dact_scale + co);
                u = u * ds + *reinterpret_cast<const floatx4*>(d.dact_shift + co);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] *= act_bwd_fast(u[e], d.dact_act) * ds[e];
        }
    }
    if (d.addend && !(d.flags & GA_CONV_ADDEND_PRE_DACT)) {
        size_t am = d.addend_bcast_n ? (size_t)(m % HoWo) : (size_t)m;
        if (d.addend_rep > 1) am = (size_t)((m / HoWo) / d.addend_rep) * HoWo + (m % HoWo);
        floatx4 a = *reinterpret_cast<const floatx4*>(d.addend + am * d.ldadd + co);
        if (d.flags & GA_CONV_ADDEND_RELU) { for (int e = 0; e < 4; ++e) a[e] = fmaxf(a[e], 0.f); }
        v += a;
    }
    if (d.addend2) v += *reinterpret_cast<const floatx4*>(d.addend2 + (size_t)m * d.ldadd2 + co);
    *reinterpret_cast<floatx4*>(d.y + (size_t)m * d.ldy + co) = v;
}

__device__ __forceinline__ void epilogue1(const ga_conv_desc& d, const int m, const int co, float v, const int HoWo) {
    if (d.bias) v += d.bias[co];
    if ((d.flags & GA_CONV_ADDEND_PRE_DACT) && d.addend) v += d.addend[(size_t)m * d.ldadd + co];
    if (d.dact_x) {
        float ds = 1.f, db = 0.f;
        if (d.dact_scale) { ds = d.dact_scale[co]; db = d.dact_shift[co]; }
        if (d.flags & GA_CONV_DACT_PRELU) {
            v *= d.dact_x[dact_row(d, (size_t)m, HoWo) * d.lddact + co] > 0.f ? 1.f : ds;
        } else {
            const float u = d.dact_x[dact_row(d, (size_t)m, HoWo) * d.lddact + co] * ds + db;
            v *= act_bwd_fast(u, d.dact_act) * ds;
        }
    }
    if (d.addend && !(d.flags & GA_CONV_ADDEND_PRE_DACT)) {
        size_t am = d.addend_bcast_n ? (size_t)(m % HoWo) : (size_t)m;
        if (d.addend_rep > 1) am = (size_t)((m / HoWo) / d.addend_rep) * HoWo + (m % HoWo);
        const float a = d.addend[am * d.ldadd + co];
        v += (d.flags & GA_CONV_ADDEND_RELU) ? fmaxf(a, 0.f) : a;
    }
    if (d.addend2) v += d.addend2[(size_t)m * d.ldadd2 + co];
    d.y[(size_t)m * d.ldy + co] = v;
}


// acc: per-wave TM x TN accumulator tiles in the 32x32 MFMA C/D layout; smem: >= BM*(BN+4) floats, free to overwrite
template <int WM, int WN, int TM, int TN>
__device__ __forceinline__ void conv_epilogue(const ga_conv_desc& d, floatx16 (&acc)[TM][TN], float* smem, const int m0,
                                              const int n0, const int M, const int vec_out, const int splits, const int split) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    constexpr int LDC = BN + 4;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int lrow = lane & 31, lh = lane >> 5;
    const int HoWo = d.Ho * d.Wo;
    float* ws = splits > 1 ? d.ws + (size_t)split * M * d.Cout : nullptr;
    if (vec_out) {
        float* Cs = smem;                                   // [BM][LDC], the K loop's buffers are free now
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    Cs[(wm * TM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * LDC + wn * TN * 32 + j * 32 + lrow] = acc[i][j][r];
        __syncthreads();
        constexpr int QL = BN / 4;                          // channel-quads per row
        constexpr int ROWS = 256 / QL;                      // rows per pass
        const int q = tid % QL, rr = tid / QL;
        const int co = n0 + 4 * q;
        if (co < d.Cout) {
            constexpr int NB = (BM / ROWS) < 4 ? (BM / ROWS) : 4;      // rows handled together: loads first, math after
            floatx4 bias4 = {0.f, 0.f, 0.f, 0.f}, ds4 = {1.f, 1.f, 1.f, 1.f}, dt4 = {0.f, 0.f, 0.f, 0.f};
            if (!ws) {
                if (d.bias) bias4 = *reinterpret_cast<const floatx4*>(d.bias + co);
                if (d.dact_x && d.dact_scale) {
                    ds4 = *reinterpret_cast<const floatx4*>(d.dact_scale + co);
                    dt4 = *reinterpret_cast<const floatx4*>(d.dact_shift + co);
                }
            }
            for (int rb0 = rr; rb0 < BM; rb0 += ROWS * NB) {
                floatx4 v[NB], u[NB], a1[NB], a2[NB];
                bool ok[NB];
#pragma unroll
                for (int k = 0; k < NB; ++k) {
                    const int r = rb0 + k * ROWS;
                    ok[k] = (r < BM) && (m0 + r < M);
                    v[k] = *reinterpret_cast<const floatx4*>(Cs + (r < BM ? r : 0) * LDC + 4 * q);
                }
                if (ws) {
#pragma unroll
                    for (int k = 0; k < NB; ++k)
                        if (ok[k]) *reinterpret_cast<floatx4*>(ws + (size_t)(m0 + rb0 + k * ROWS) * d.Cout + co) = v[k];
                    continue;
                }
                if (d.dact_x) {
#pragma unroll
                    for (int k = 0; k < NB; ++k) {
                        const size_t m = ok[k] ? dact_row(d, (size_t)(m0 + rb0 + k * ROWS), HoWo) : 0;
                        u[k] = *reinterpret_cast<const floatx4*>(d.dact_x + m * d.lddact + co);
                    }
                }
                if (d.addend) {
#pragma unroll
                    for (int k = 0; k < NB; ++k) {
                        size_t m = ok[k] ? (size_t)(m0 + rb0 + k * ROWS) : 0;
                        if (d.addend_bcast_n) m = m % HoWo;
                        if (d.addend_rep > 1) m = ((m / HoWo) / d.addend_rep) * HoWo + (m % HoWo);
                        a1[k] = *reinterpret_cast<const floatx4*>(d.addend + m * d.ldadd + co);
                    }
                }
                if (d.addend2) {
#pragma unroll
                    for (int k = 0; k < NB; ++k) {
                        const size_t m = ok[k] ? (size_t)(m0 + rb0 + k * ROWS) : 0;
                        a2[k] = *reinterpret_cast<const floatx4*>(d.addend2 + m * d.ldadd2 + co);
                    }
                }
#pragma unroll
                for (int k = 0; k < NB; ++k) {
                    floatx4 o = v[k] + bias4;
                    const bool pre = (d.flags & GA_CONV_ADDEND_PRE_DACT) != 0;
                    if (pre && d.addend) o += a1[k];
                    if (d.dact_x && (d.flags & GA_CONV_DACT_PRELU)) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] *= u[k][e] > 0.f ? 1.f : ds4[e];
                    } else if (d.dact_x) {
                        const floatx4 uu = u[k] * ds4 + dt4;
                        if (d.dact_act == GA_ACT_SILU) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) { const float sg = fast_sigmoid(uu[e]); o[e] *= sg * (1.0f + uu[e] * (1.0f - sg)) * ds4[e]; }
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e) o[e] *= act_bwd_fast(uu[e], d.dact_act) * ds4[e];
                        }
                    }
                    if (d.addend && !pre) {
                        if (d.flags & GA_CONV_ADDEND_RELU) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) o[e] += fmaxf(a1[k][e], 0.f);
                        } else {
                            o += a1[k];
                        }
                    }
                    if (d.addend2) o += a2[k];
                    if (ok[k]) *reinterpret_cast<floatx4*>(d.y + (size_t)(m0 + rb0 + k * ROWS) * d.ldy + co) = o;
                }
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int co = n0 + wn * TN * 32 + j * 32 + lrow;
            if (co >= d.Cout) continue;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + wm * TM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (m >= M) continue;
                    if (ws) ws[(size_t)m * d.Cout + co] = acc[i][j][r];
                    else epilogue1(d, m, co, acc[i][j][r], HoWo);
                }
            }
        }
    }
}

}  // namespace ga
