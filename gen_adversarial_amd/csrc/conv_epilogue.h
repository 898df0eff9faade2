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


// The accumulators of one wave: TM x TN blocks of 32 x 32 outputs, held either as one v_mfma_f32_32x32x16 C/D tile each
// (floatx16: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)) or as 2 x 2 v_mfma_f32_16x16x32 tiles each
// (floatx4 [2 TM][2 TN]: col = lane & 15, row = 4 (lane >> 4) + r).  acc_at() walks either form as (row, col, value) of the wave's
// TM*32 x TN*32 sub-tile.
template <int TM, int TN, class F>
__device__ __forceinline__ void acc_walk(floatx16 (&acc)[TM][TN], const int lane, F&& f) {
    const int lrow = lane & 31, lh = lane >> 5;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) f(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh, j * 32 + lrow, acc[i][j][r]);
}
template <int TM2, int TN2, class F>
__device__ __forceinline__ void acc_walk(floatx4 (&acc)[TM2][TN2], const int lane, F&& f) {
    const int lc = lane & 15, lq = lane >> 4;
#pragma unroll
    for (int i = 0; i < TM2; ++i)
#pragma unroll
        for (int j = 0; j < TN2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) f(i * 16 + 4 * lq + r, j * 16 + lc, acc[i][j][r]);
}

// acc: per-wave accumulators (either form above); smem: >= BM*(BN+4) floats, free to overwrite.
// Tile rows -> output pixels: row r of the tile is pixel m0 + r (tiles that are runs of the [N, Ho, Wo] pixel index: the default,
// tw_shift = 31) or, for 2-D tiles of 2^tw_shift columns (conv_thin3: 8 x 16 pixel tiles of a wide image), pixel
// m0 + (r >> tw_shift) * row_stride + (r & (2^tw_shift - 1)) with m0 the tile's top-left pixel and row_stride = Wo.
template <int WM, int WN, int TM, int TN, class ACC>
__device__ __forceinline__ void conv_epilogue(const ga_conv_desc& d, ACC& acc, float* smem, const int m0,
                                              const int n0, const int M, const int vec_out, const int splits, const int split,
                                              const int tw_shift = 31, const int row_stride = 0) {
    const int tw_mask = (int)((1u << tw_shift) - 1u);
    auto mrow = [&](const int r) __attribute__((always_inline)) { return m0 + (r >> tw_shift) * row_stride + (r & tw_mask); };
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    constexpr int LDC = BN + 4;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int HoWo = d.Ho * d.Wo;
    float* ws = splits > 1 ? d.ws + (size_t)split * M * d.Cout : nullptr;
    if (vec_out) {
        float* Cs = smem;                                   // [BM][LDC], the K loop's buffers are free now
        constexpr int QL = BN / 4;                          // channel-quads per row
        constexpr int ROWS = 256 / QL;                      // rows per pass
        constexpr int NB = (BM / ROWS) < 4 ? (BM / ROWS) : 4;      // rows handled together: loads first, math after
        constexpr int NIT = BM / (ROWS * NB);               // batches of NB rows per thread
        static_assert(NIT * ROWS * NB == BM, "whole batches");
        const int q = tid % QL, rr = tid / QL;
        const int co = n0 + 4 * q;
        const bool mine = co < d.Cout;
        // Each batch of NB rows requests its global operands (act' input, addends), then does its arithmetic and stores.  (r04: a
        // software-pipelined form — batch b + 1's operands requested before batch b's arithmetic, batch 0's before the LDS
        // transposition, two register sets, the batch loop unrolled — measured 0 - 5 % SLOWER per launch on forward and backward
        // 3x3 shapes in interleaved A/B runs on one box, gpurun_out/r04_ab_epi.log: the co-resident workgroup already covers the
        // round trips, the unrolled code only costs registers and instruction cache.  Not kept.)
        floatx4 u[1][NB], a1[1][NB], a2[1][NB];
        auto issue = [&](const int set, const int rb0) __attribute__((always_inline)) {
#pragma unroll
            for (int k = 0; k < NB; ++k) {
                const int r = rb0 + k * ROWS;
                const int mr = mrow(r);
                const bool ok = mr < M;
                if (d.dact_x) {
                    const size_t m = ok ? dact_row(d, (size_t)mr, HoWo) : 0;
                    u[set][k] = *reinterpret_cast<const floatx4*>(d.dact_x + m * d.lddact + co);
                }
                if (d.addend) {
                    size_t m = ok ? (size_t)mr : 0;
                    if (d.addend_bcast_n) m = m % HoWo;
                    if (d.addend_rep > 1) m = ((m / HoWo) / d.addend_rep) * HoWo + (m % HoWo);
                    a1[set][k] = *reinterpret_cast<const floatx4*>(d.addend + m * d.ldadd + co);
                }
                if (d.addend2) {
                    const size_t m = ok ? (size_t)mr : 0;
                    a2[set][k] = *reinterpret_cast<const floatx4*>(d.addend2 + m * d.ldadd2 + co);
                }
            }
        };
        floatx4 bias4 = {0.f, 0.f, 0.f, 0.f}, ds4 = {1.f, 1.f, 1.f, 1.f}, dt4 = {0.f, 0.f, 0.f, 0.f};
        if (mine && !ws) {
            if (d.bias) bias4 = *reinterpret_cast<const floatx4*>(d.bias + co);
            if (d.dact_x && d.dact_scale) {
                ds4 = *reinterpret_cast<const floatx4*>(d.dact_scale + co);
                dt4 = *reinterpret_cast<const floatx4*>(d.dact_shift + co);
            }
        }
        acc_walk(acc, lane, [&](const int row, const int col, const float v) {
            Cs[(wm * TM * 32 + row) * LDC + wn * TN * 32 + col] = v;
        });
        __syncthreads();
        if (mine) {
#pragma unroll 1
            for (int it = 0; it < NIT; ++it) {
                const int rb0 = rr + it * ROWS * NB, set = 0;
                floatx4 v[NB];
                bool ok[NB];
                if (!ws) issue(0, rb0);
#pragma unroll
                for (int k = 0; k < NB; ++k) {
                    const int r = rb0 + k * ROWS;
                    ok[k] = mrow(r) < M;
                    v[k] = *reinterpret_cast<const floatx4*>(Cs + r * LDC + 4 * q);
                }
                if (ws) {
#pragma unroll
                    for (int k = 0; k < NB; ++k)
                        if (ok[k]) *reinterpret_cast<floatx4*>(ws + (size_t)mrow(rb0 + k * ROWS) * d.Cout + co) = v[k];
                    continue;
                }
#pragma unroll
                for (int k = 0; k < NB; ++k) {
                    floatx4 o = v[k] + bias4;
                    const bool pre = (d.flags & GA_CONV_ADDEND_PRE_DACT) != 0;
                    if (pre && d.addend) o += a1[set][k];
                    if (d.dact_x && (d.flags & GA_CONV_DACT_PRELU)) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] *= u[set][k][e] > 0.f ? 1.f : ds4[e];
                    } else if (d.dact_x) {
                        const floatx4 uu = u[set][k] * ds4 + dt4;
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
                            for (int e = 0; e < 4; ++e) o[e] += fmaxf(a1[set][k][e], 0.f);
                        } else {
                            o += a1[set][k];
                        }
                    }
                    if (d.addend2) o += a2[set][k];
                    if (ok[k]) *reinterpret_cast<floatx4*>(d.y + (size_t)mrow(rb0 + k * ROWS) * d.ldy + co) = o;
                }
            }
        }
    } else {
        acc_walk(acc, lane, [&](const int row, const int col, const float v) {
            const int co = n0 + wn * TN * 32 + col, m = mrow(wm * TM * 32 + row);
            if (co < d.Cout && m < M) {
                if (ws) ws[(size_t)m * d.Cout + co] = v;
                else epilogue1(d, m, co, v, HoWo);
            }
        });
    }
}

}  // namespace ga
