// Device helpers shared by the fused decoder-cell kernels (dec_cell.hip: whole-image workgroups; dec_cell_halo.hip: tiles with a
// recomputed halo): split-bf16 fragments, the depthwise loops over an fp32 LDS plane, weight-chunk staging, SiLU.
#pragma once
#include "ga_common.h"

namespace ga {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int DC_CH = 32;       // hidden channels per chunk
constexpr int DC_PS = 40;       // floats per pixel of an fp32 LDS plane (32 + 8 pad)
constexpr int DC_LDB = 40;      // bf16 per pixel of a bf16 LDS plane (80 B rows: conflict-free 16-B fragment reads)

// -DGA_DC_TRACE (make dctrace -> libga_ops_dctrace.so, tools/dec_cell_trace.py): per-phase shader-clock sums of the chunk
// loop, lane 0 of every wave of workgroup 0
#ifdef GA_DC_TRACE
__device__ unsigned long long ga_dc_trace_buf[8 * 16];
#define DC_T0 unsigned long long tsum[12] = {}; unsigned long long tprev = __builtin_amdgcn_s_memtime();
#define DC_T(i) { const unsigned long long tn = __builtin_amdgcn_s_memtime(); tsum[i] += tn - tprev; tprev = tn; }
#define DC_TEND if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) { for (int i = 0; i < 12; ++i) ga_dc_trace_buf[(threadIdx.x >> 6) * 16 + i] = tsum[i]; }
#else
#define DC_T0
#define DC_T(i)
#define DC_TEND
#endif

struct dc_geom { int lw, lhw, W, HW, PW, PH; };      // log2 W, log2 (H W); padded plane = (H + 4) x (W + 4)

__device__ __forceinline__ void split8(const floatx4 a, const floatx4 b, bf16x8& hi, bf16x8& lo) {
    const bf16x4 ha = __builtin_convertvector(a, bf16x4), hb = __builtin_convertvector(b, bf16x4);
    const bf16x4 la = __builtin_convertvector(a - __builtin_convertvector(ha, floatx4), bf16x4);
    const bf16x4 lb = __builtin_convertvector(b - __builtin_convertvector(hb, floatx4), bf16x4);
    hi = __builtin_shufflevector(ha, hb, 0, 1, 2, 3, 4, 5, 6, 7);
    lo = __builtin_shufflevector(la, lb, 0, 1, 2, 3, 4, 5, 6, 7);
}

// index of workgroup-local pixel p in the framed plane
__device__ __forceinline__ int plane_idx(const int p, const dc_geom& g) {
    const int ni = p >> g.lhw, rem = p & (g.HW - 1);
    return (ni * g.PH + (rem >> g.lw) + 2) * g.PW + (rem & (g.W - 1)) + 2;
}

// First pixel of strip s (SW adjacent pixels of one image row).  Consecutive strips are vertically adjacent rows, not
// horizontal neighbours: a plane row is (W + 4) * 160 B = 128 B mod 256 B for W = 8, 16, ..., so the two strips whose
// 8 channel-quad lanes share a 16-lane LDS access phase cover all 64 banks (horizontal neighbours, SW * 160 B = 0 mod 256 B
// apart, met on the same 32).
template <int SW>
__device__ __forceinline__ int strip_pixel(const int s, const dc_geom& g) {
    const int H = g.HW >> g.lw, spr = g.W / SW;                       // strips per image row
    const int h = s & (H - 1), rest = s >> (g.lhw - g.lw);            // rest = (image, segment)
    const int seg = rest % spr, ni = rest / spr;
    return ni * g.HW + h * g.W + seg * SW;
}

// acc[i] += A[i] . B for the resident A-fragments of TMW 32-pixel tiles over K = 16 KS; B rows at wh / wl (row = this lane's
// output column, 8 lh already added), contiguous along K
template <int TMW, int KS>
__device__ __forceinline__ void gemm_resident(floatx16 (&acc)[TMW], const bf16x8 (&ah)[TMW][KS], const bf16x8 (&al)[TMW][KS],
                                              const __bf16* wh, const __bf16* wl) {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const bf16x8 bh = *reinterpret_cast<const bf16x8*>(wh + ks * 16);
        const bf16x8 bl = *reinterpret_cast<const bf16x8*>(wl + ks * 16);
#pragma unroll
        for (int i = 0; i < TMW; ++i) {
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i][ks], bh, acc[i], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i][ks], bl, acc[i], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i][ks], bh, acc[i], 0, 0, 0);
        }
    }
}

// depthwise 5x5 of SW adjacent pixels x 4 channels from a framed plane; base = top-left of the (5 x (SW + 4)) window
template <int SW>
__device__ __forceinline__ void dw_strip(floatx4 (&acc)[SW], const float* base, const float* taps, const int PW) {
    const floatx4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < SW; ++j) acc[j] = zero;
#pragma unroll 1
    for (int kh = 0; kh < 5; ++kh) {
        floatx4 w5[5], in[SW + 4];
#pragma unroll
        for (int kw = 0; kw < 5; ++kw) w5[kw] = *reinterpret_cast<const floatx4*>(taps + (kh * 5 + kw) * DC_CH);
#pragma unroll
        for (int j = 0; j < SW + 4; ++j) in[j] = *reinterpret_cast<const floatx4*>(base + (kh * PW + j) * DC_PS);
#pragma unroll
        for (int j = 0; j < SW; ++j)
#pragma unroll
            for (int kw = 0; kw < 5; ++kw) acc[j] += in[j + kw] * w5[kw];
    }
}

// The SW outputs of a thread as a block of BH rows x BW columns.  At 128 channels (SW = 8) a 2 x 4 block reads a 6 x 8 window
// (48 16-byte LDS reads) where the 1 x 8 strip read 5 x 12 (60): the depthwise phases are LDS-read bound.  Each output still sums
// its 25 taps in the order kh, kw: bitwise the strip's result.  Horizontally adjacent blocks (4 pixels = 640 B = 128 B mod 256 B) share
// a 16-lane LDS phase without bank conflicts; at 256 channels (SW = 4) the 1 x 4 strip of vertically adjacent rows stays.
template <int BH, int BW>
__device__ __forceinline__ void dw_tile(floatx4 (&acc)[BH * BW], const float* base, const float* taps, const int PW) {
    if constexpr (BH == 1) {
        dw_strip<BW>(acc, base, taps, PW);
    } else {
        static_assert(BH == 2, "blocks of one or two rows");
        const floatx4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 2 * BW; ++j) acc[j] = zero;
        floatx4 wprev[5];
#pragma unroll
        for (int r = 0; r < 6; ++r) {                               // input row r feeds output row 0 (tap row r) and row 1 (tap row r - 1)
            floatx4 in[BW + 4], wcur[5];
#pragma unroll
            for (int j = 0; j < BW + 4; ++j) in[j] = *reinterpret_cast<const floatx4*>(base + (r * PW + j) * DC_PS);
            if (r < 5) {
#pragma unroll
                for (int kw = 0; kw < 5; ++kw) wcur[kw] = *reinterpret_cast<const floatx4*>(taps + (r * 5 + kw) * DC_CH);
#pragma unroll
                for (int j = 0; j < BW; ++j)
#pragma unroll
                    for (int kw = 0; kw < 5; ++kw) acc[j] += in[j + kw] * wcur[kw];
            }
            if (r > 0) {
#pragma unroll
                for (int j = 0; j < BW; ++j)
#pragma unroll
                    for (int kw = 0; kw < 5; ++kw) acc[BW + j] += in[j + kw] * wprev[kw];
            }
            if (r < 5) {
#pragma unroll
                for (int kw = 0; kw < 5; ++kw) wprev[kw] = wcur[kw];
            }
        }
    }
}

// first pixel of block s (BH x BW outputs); blocks are numbered row-major inside an image
template <int BH, int BW>
__device__ __forceinline__ int tile_pixel(const int s, const dc_geom& g) {
    if constexpr (BH == 1) {
        return strip_pixel<BW>(s, g);
    } else {
        const int H = g.HW >> g.lw, bpr = g.W / BW, nblk = (H / BH) * bpr;
        const int ni = s / nblk, rem = s - ni * nblk;
        return ni * g.HW + (rem / bpr) * BH * g.W + (rem % bpr) * BW;
    }
}

typedef unsigned uintx4 __attribute__((ext_vector_type(4)));

// One chunk of a [rows][K]-major split-bf16 weight matrix through LDS.  ROWMAJOR_K (the expand conv W1 and, backward, W2^T):
// 32 rows (the chunk's hidden channels) x C columns, LDS pitch C + 8 elements.  Otherwise (the project conv W2, [C][Hd]):
// C rows x the chunk's 32 columns, LDS pitch 40.  Both pitches are 16 B mod 256 B: the 16-B fragment reads of 16 lanes
// with consecutive rows fall on 16 distinct bank slots.  256 threads move 16-B pieces; NP pieces per thread and array.
template <int C, bool ROWS32, int NTHR = 256>
struct w_chunk {
    static constexpr int NP = 4 * C / NTHR;                          // 4 C 16-byte pieces per array, NTHR threads
    static constexpr int PITCH = ROWS32 ? C + 8 : 40;
    static constexpr int ELEMS = (ROWS32 ? 32 : C) * PITCH;          // bf16 elements of one LDS copy (hi or lo)
    uintx4 hi[NP], lo[NP];
    __device__ __forceinline__ void issue(const __bf16* gh, const __bf16* gl, const int ld, const int h0, const int tid) {
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            const int q = tid + NTHR * k;
            const size_t o = ROWS32 ? (size_t)(h0 + q / (C / 8)) * ld + (q % (C / 8)) * 8 : (size_t)(q >> 2) * ld + h0 + (q & 3) * 8;
            hi[k] = *reinterpret_cast<const uintx4*>(gh + o);
            lo[k] = *reinterpret_cast<const uintx4*>(gl + o);
        }
    }
    __device__ __forceinline__ void store(__bf16* sh, __bf16* sl, const int tid) const {
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            const int q = tid + NTHR * k;
            const int o = ROWS32 ? (q / (C / 8)) * PITCH + (q % (C / 8)) * 8 : (q >> 2) * PITCH + (q & 3) * 8;
            *reinterpret_cast<uintx4*>(sh + o) = hi[k];
            *reinterpret_cast<uintx4*>(sl + o) = lo[k];
        }
    }
};

__device__ __forceinline__ float silu_f(const float v) { return v * fast_sigmoid(v); }
__device__ __forceinline__ float dsilu_f(const float v) { const float s = fast_sigmoid(v); return s * (1.0f + v * (1.0f - s)); }

// interleave hint for one basic block holding NM MFMAs and ~NV VALU instructions of an independent chain: MFMA, a slice of
// the VALU work, one LDS access, repeated (the matrix pipe takes 32 clocks per MFMA, the wave issues beside it)
template <int NM, int NV>
__device__ __forceinline__ void interleave_mfma_valu() {
    constexpr int VPM = (NV + NM - 1) / NM;
#pragma unroll
    for (int k = 0; k < NM; ++k) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, VPM, 0);
    }
}


}  // namespace ga
