// ga_dec_cell — the residual branch of NVAE's ResidualCellDecoder (reference: NVAE/modules/architecture.py:139-186,
// BN -> 1x1 (C -> 6C) -> BN -> SiLU -> depthwise 5x5 -> BN -> SiLU -> 1x1 (6C -> C) -> BN, BatchNorms folded) in ONE launch.
//
// Unfused, the two 6C-wide tensors cross HBM ten times per cell (forward + backward) and bound the decoder; here they
// never leave the CU.  A workgroup (4 waves) owns M = 128 TMW pixels = whole images (no halo to exchange) and walks the
// hidden width in chunks of 32 channels:
//   GEMM1  t1c [M x 32] = x [M x C] . W1c            x stays resident in registers as split-bf16 MFMA A-fragments
//   SiLU   -> LDS plane P1 (fp32, [pixel][32 + 8 pad], each image framed by a 2-pixel zero border)
//   dw5    t2c = taps * P1 + bd                      thread = (channel quad, strip of SW adjacent pixels), as ga_dwconv5
//   SiLU   -> split-bf16 -> LDS planes P2 (hi, lo)   the A operand of
//   GEMM2  acc [M x C] += s2c [M x 32] . W2c         accumulators resident in registers (the other half of the budget)
// The weights of a chunk (32 rows of W1, 32 columns of W2: 16 / 32 KB as split bf16) go HBM / L2 -> registers a whole chunk
// ahead and registers -> LDS at the top of their chunk: the B-fragments are then 16-B LDS reads (read straight from global
// memory, one exposed L2 round trip per k step made the kernel 2-3x slower: one wave per SIMD hides nothing).
// Contractions are the three-MFMA split-bf16 products of conv_bf3 (same operand split, same k order), the depthwise
// part is the fp32 loop of dwconv5 in the same order: the fused result is, bit for bit, that of the three launches it replaces
// (tests/test_dec_cell_gpu.py, tests/test_fullsize_gpu.py).
//
// Backward (d loss / d t1 from d loss / d t3; d x is then ONE 1x1 ga_conv2d of it): x AND dt3 = dout * ps[n] + pb[n]
// resident as A-fragments; per chunk t1c and t2c are recomputed (GEMM1 + dw5), then
//   GEMM3  g [M x 32] = dt3 [M x C] . W2c^T ;  g *= SiLU'(t2c) ;  dw5^T through the same LDS plane ;  *= SiLU'(t1c) -> HBM.
// Register budget per lane (one workgroup per CU, 512 registers): forward 128 (x) + 128 (acc), backward 128 + 128 (dt3).
#include "ga_common.h"
#include "dec_cell_common.h"

namespace ga {

// Forward.  Software pipeline over the chunks (two barriers per chunk, every MFMA phase beside an independent VALU phase):
//   A  depthwise(ch)                          P1, taps -> registers                              (LDS-read bound)
//   B  GEMM1(ch+1)  ||  SiLU + split(ch) -> P2
//   -- barrier 1 --   W1(ch+2), taps(ch+1): registers -> LDS
//   C  GEMM2(ch)    ||  SiLU(t1(ch+1)) -> P1
//   -- barrier 2 --   W2(ch+1): registers -> LDS
// The global loads behind the register -> LDS stores are issued at the top of the iteration.  Chunk indices past the end
// are clamped: the last iteration's GEMM1 / P1 write are wasted work beside phases that run anyway.
// NW waves per workgroup (4: one per SIMD, the whole register file for x + accumulators; 8: two per SIMD at half the rows per wave —
// the vector-ALU phases of one wave then run beside the other's MFMAs, and a SIMD issues a vector instruction every 2 clocks
// instead of every 4 (one wave alone cannot: MI355X_MICROARCH.md, 'vector-instruction ISSUE cost'))
template <int C, int TMW, int NW>
__global__ void __launch_bounds__(64 * NW, NW / 4) dec_cell_fwd_kernel(const ga_dec_cell_desc d, const dc_geom gm) {
    constexpr int NTHR = 64 * NW;
    constexpr int M = 32 * TMW * NW, KS = C / 16, NT = C / 32, SW = M / (NTHR / 8);
    constexpr int BH = TMW == 2 ? 2 : 1, BW = SW / BH;      // this thread's SW depthwise outputs: BH rows x BW columns
    using WA = w_chunk<C, true, NTHR>;
    using WB = w_chunk<C, false, NTHR>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* wS = smem;                                                   // [25][32] taps of the chunk
    float* P1 = smem + 25 * DC_CH;                                      // framed fp32 plane
    const int plane_px = (M >> gm.lhw) * gm.PH * gm.PW;
    __bf16* P2h = reinterpret_cast<__bf16*>(P1 + plane_px * DC_PS);     // [M][DC_LDB] bf16, hi then lo
    __bf16* P2l = P2h + M * DC_LDB;
    __bf16* W1h = P2l + M * DC_LDB;                                     // 32 rows of W1, then 32 columns of W2
    __bf16* W1l = W1h + WA::ELEMS;
    __bf16* W2h = W1l + WA::ELEMS;
    __bf16* W2l = W2h + WB::ELEMS;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lrow = lane & 31, lh = lane >> 5;
    const int c4 = tid & 7, strip = tid >> 3;
    const size_t pix0 = (size_t)blockIdx.x * M;
    const int wb = wave * 32 * TMW;
    const floatx4 zero = {0.f, 0.f, 0.f, 0.f};
    const __bf16* g1h = reinterpret_cast<const __bf16*>(d.w1_hi);
    const __bf16* g1l = reinterpret_cast<const __bf16*>(d.w1_lo);
    const __bf16* g2h = reinterpret_cast<const __bf16*>(d.w2_hi);
    const __bf16* g2l = reinterpret_cast<const __bf16*>(d.w2_lo);
#ifdef GA_DC_PRIO
    // two waves per SIMD: the second-dispatched half loses the vector-issue arbitration on every phase (priority, then age) and
    // arrives last at every barrier; one static priority raise evens them out (MI355X_MICROARCH.md, two waves per SIMD, item 4)
    if (NW == 8 && __builtin_amdgcn_readfirstlane(threadIdx.x) >= 256) __builtin_amdgcn_s_setprio(1);
#endif
    const int nch = d.Hd / DC_CH;
    const int last = (nch - 1) * DC_CH;
    const int tap_o = (tid < 200 ? (tid >> 3) : 0) * d.Hd + 4 * c4;     // threads >= 200 load tap 0 again and drop it

    WA wa;
    WB wq;
    floatx4 taps;
    wa.issue(g1h, g1l, C, 0, tid);
    wq.issue(g2h, g2l, d.Hd, 0, tid);
    taps = *reinterpret_cast<const floatx4*>(d.wd + tap_o);

    for (int i = tid; i < plane_px * (DC_PS / 4); i += NTHR) reinterpret_cast<floatx4*>(P1)[i] = zero;

    bf16x8 xh[TMW][KS], xl[TMW][KS];
#pragma unroll
    for (int i = 0; i < TMW; ++i)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const float* p = d.x + (pix0 + wb + i * 32 + lrow) * C + ks * 16 + 8 * lh;
            split8(*reinterpret_cast<const floatx4*>(p), *reinterpret_cast<const floatx4*>(p + 4), xh[i][ks], xl[i][ks]);
        }
    floatx16 acc[TMW][NT];
#pragma unroll
    for (int i = 0; i < TMW; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // this thread's strip of SW pixels (one image row segment) and its window in the framed plane
    const int p0 = tile_pixel<BH, BW>(strip, gm);
    const int win = plane_idx(p0, gm) - 2 * gm.PW - 2;
    int prow[TMW][4];                                   // plane index of accumulator rows 8 q (+ 0..3) of each 32-pixel tile
#pragma unroll
    for (int i = 0; i < TMW; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) prow[i][q] = plane_idx(wb + i * 32 + 8 * q + 4 * lh, gm) * DC_PS + lrow;
    const __bf16* w1h = W1h + lrow * WA::PITCH + 8 * lh;
    const __bf16* w1l = W1l + lrow * WA::PITCH + 8 * lh;

    // ---- prologue: chunk 0's weights -> LDS, GEMM1(0), SiLU -> P1, then W1(1) -> LDS
    wa.store(W1h, W1l, tid);
    wq.store(W2h, W2l, tid);
    if (tid < 200) *reinterpret_cast<floatx4*>(wS + (tid >> 3) * DC_CH + 4 * c4) = taps;
    __syncthreads();
    wa.issue(g1h, g1l, C, min(DC_CH, last), tid);
    floatx16 t1[TMW];
#pragma unroll
    for (int i = 0; i < TMW; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) t1[i][r] = 0.f;
    gemm_resident<TMW, KS>(t1, xh, xl, w1h, w1l);
    {
        const float b1v = d.b1[lrow];
#pragma unroll
        for (int i = 0; i < TMW; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) P1[prow[i][r >> 2] + (r & 3) * DC_PS] = silu_f(t1[i][r] + b1v);
    }
    __syncthreads();
    wa.store(W1h, W1l, tid);
    __syncthreads();

    DC_T0
#pragma unroll 1
    for (int ch = 0; ch < nch; ++ch) {
        const int h0 = ch * DC_CH;
        const int h1 = min(h0 + DC_CH, last), h2 = min(h0 + 2 * DC_CH, last);
        // the small loads first: a wait on them must not wait on the weight prefetch behind them (vmcnt retires in order)
        const float b1v = d.b1[h1 + lrow];
        const floatx4 bd4 = *reinterpret_cast<const floatx4*>(d.bd + h0 + 4 * c4);
        __builtin_amdgcn_sched_barrier(0);
        wa.issue(g1h, g1l, C, h2, tid);
        wq.issue(g2h, g2l, d.Hd, h1, tid);
        // ---- A: depthwise 5x5 of chunk ch
        floatx4 a[SW];
        dw_tile<BH, BW>(a, P1 + win * DC_PS + 4 * c4, wS + 4 * c4, gm.PW);
        DC_T(0)
        // ---- B: GEMM1 of chunk ch + 1  ||  SiLU + split of chunk ch -> P2.  One scheduling region per strip pixel: its share of
        // the k steps (6 / 12 MFMAs, the next region's B-fragments already on their way) beside its ~40 VALU instructions.
#pragma unroll
        for (int i = 0; i < TMW; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) t1[i][r] = 0.f;
        {
            constexpr int KPJ = KS / SW;                // k steps per strip pixel
            bf16x8 bh[KPJ], bl[KPJ];
#pragma unroll
            for (int k = 0; k < KPJ; ++k) {
                bh[k] = *reinterpret_cast<const bf16x8*>(w1h + k * 16);
                bl[k] = *reinterpret_cast<const bf16x8*>(w1l + k * 16);
            }
#pragma unroll
            for (int j = 0; j < SW; ++j) {
                bf16x8 nh[KPJ], nl[KPJ];
                if (j + 1 < SW) {
#pragma unroll
                    for (int k = 0; k < KPJ; ++k) {
                        nh[k] = *reinterpret_cast<const bf16x8*>(w1h + ((j + 1) * KPJ + k) * 16);
                        nl[k] = *reinterpret_cast<const bf16x8*>(w1l + ((j + 1) * KPJ + k) * 16);
                    }
                }
#pragma unroll
                for (int k = 0; k < KPJ; ++k)
#pragma unroll
                    for (int i = 0; i < TMW; ++i) {
                        t1[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xl[i][j * KPJ + k], bh[k], t1[i], 0, 0, 0);
                        t1[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh[i][j * KPJ + k], bl[k], t1[i], 0, 0, 0);
                        t1[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh[i][j * KPJ + k], bh[k], t1[i], 0, 0, 0);
                    }
                floatx4 v = a[j] + bd4;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = silu_f(v[e]);
                const bf16x4 hi = __builtin_convertvector(v, bf16x4);
                const bf16x4 lo = __builtin_convertvector(v - __builtin_convertvector(hi, floatx4), bf16x4);
                const int pj = p0 + (j / BW) * gm.W + (j % BW);                 // pixel of output j of the block
                *reinterpret_cast<bf16x4*>(P2h + pj * DC_LDB + 4 * c4) = hi;
                *reinterpret_cast<bf16x4*>(P2l + pj * DC_LDB + 4 * c4) = lo;
                interleave_mfma_valu<KPJ * TMW * 3, 42>();
                __builtin_amdgcn_sched_barrier(0);
                if (j + 1 < SW) {
#pragma unroll
                    for (int k = 0; k < KPJ; ++k) { bh[k] = nh[k]; bl[k] = nl[k]; }
                }
            }
        }
        DC_T(1)
        __syncthreads();
        wa.store(W1h, W1l, tid);
        // the next chunk's taps travel during phase C only (every depthwise read of this chunk's is behind the barrier above):
        // four registers that are free in phases A and B, where the kernel is at its register limit
        taps = *reinterpret_cast<const floatx4*>(d.wd + tap_o + h1);
        DC_T(2)
        // ---- C: GEMM2 of chunk ch  ||  SiLU(t1 of chunk ch + 1) -> P1.  One scheduling region per (output tile, k step): 6 / 3 MFMAs
        // beside the SiLU of 4 / 1 accumulator elements.
        {
            bf16x8 ah[TMW][2], al[TMW][2];
#pragma unroll
            for (int i = 0; i < TMW; ++i)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const int o = (wb + i * 32 + lrow) * DC_LDB + ks * 16 + 8 * lh;
                    ah[i][ks] = *reinterpret_cast<const bf16x8*>(P2h + o);
                    al[i][ks] = *reinterpret_cast<const bf16x8*>(P2l + o);
                }
            const __bf16* w2h = W2h + lrow * WB::PITCH + 8 * lh;
            const __bf16* w2l = W2l + lrow * WB::PITCH + 8 * lh;
            constexpr int NG = 2 * NT, EPG = TMW * 16 / NG;     // regions, accumulator elements per region
            bf16x8 bh = *reinterpret_cast<const bf16x8*>(w2h), bl = *reinterpret_cast<const bf16x8*>(w2l);
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const int j = g >> 1, ks = g & 1;
                bf16x8 nh, nl;
                if (g + 1 < NG) {
                    nh = *reinterpret_cast<const bf16x8*>(w2h + ((g + 1) >> 1) * 32 * WB::PITCH + ((g + 1) & 1) * 16);
                    nl = *reinterpret_cast<const bf16x8*>(w2l + ((g + 1) >> 1) * 32 * WB::PITCH + ((g + 1) & 1) * 16);
                }
#pragma unroll
                for (int i = 0; i < TMW; ++i) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i][ks], bh, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i][ks], bl, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i][ks], bh, acc[i][j], 0, 0, 0);
                }
#pragma unroll
                for (int e = 0; e < EPG; ++e) {
                    const int idx = g * EPG + e, i = idx >> 4, r = idx & 15;
                    P1[prow[i][r >> 2] + (r & 3) * DC_PS] = silu_f(t1[i][r] + b1v);
                }
                interleave_mfma_valu<TMW * 3, EPG * 9>();
                __builtin_amdgcn_sched_barrier(0);
                if (g + 1 < NG) { bh = nh; bl = nl; }
            }
        }
        DC_T(3)
        if (tid < 200) *reinterpret_cast<floatx4*>(wS + (tid >> 3) * DC_CH + 4 * c4) = taps;
        __syncthreads();
        wq.store(W2h, W2l, tid);
        DC_T(4)
    }
    DC_TEND
    // ---- t3 = acc + b2
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const float b2v = d.b2[j * 32 + lrow];
#pragma unroll
        for (int i = 0; i < TMW; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wb + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                d.y[(pix0 + row) * C + j * 32 + lrow] = acc[i][j][r] + b2v;
            }
    }
}

// Backward, per chunk: (a) recompute t1c (GEMM1): SiLU' -> P4, SiLU -> P1 | (b) depthwise -> SiLU'(t2c) in registers | (c) GEMM3 =
// dt3 . W2c^T | (d) -> P1 | (e) own strip *= SiLU'(t2c) | (f) depthwise^T, * P4 -> HBM; a barrier after each.  GEMM3 runs beside
// SiLU'(t2c) in small scheduling regions.  (A two-GEMMs-beside-VALU software pipeline like the forward kernel's measured 7.75 + 5.29 ms
// against 7.12 + 5.69 ms per 512 rows for this form: no gain for twice the bookkeeping.)
template <int C, int TMW, int NW>
__global__ void __launch_bounds__(64 * NW, NW / 4) dec_cell_bwd_kernel(const ga_dec_cell_desc d, const dc_geom gm) {
    constexpr int NTHR = 64 * NW;
    constexpr int M = 32 * TMW * NW, KS = C / 16, SW = M / (NTHR / 8);
    constexpr int BH = TMW == 2 ? 2 : 1, BW = SW / BH;
    using WA = w_chunk<C, true, NTHR>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* wS = smem;                                                   // [25][32] forward taps of the chunk
    float* wT = smem + 25 * DC_CH;                                      // [25][32] flipped taps
    float* P1 = smem + 50 * DC_CH;                                      // framed fp32 plane: silu(t1c), later dt2c
    const int plane_px = (M >> gm.lhw) * gm.PH * gm.PW;
    float* P4 = P1 + plane_px * DC_PS;                                  // [M][DC_PS] silu'(t1c)
    __bf16* W1h = reinterpret_cast<__bf16*>(P4 + M * DC_PS);            // the chunk's rows of W1, then of W2^T
    __bf16* W1l = W1h + WA::ELEMS;
    __bf16* W2h = W1l + WA::ELEMS;
    __bf16* W2l = W2h + WA::ELEMS;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lrow = lane & 31, lh = lane >> 5;
    const int c4 = tid & 7, strip = tid >> 3;
    const size_t pix0 = (size_t)blockIdx.x * M;
    const int wb = wave * 32 * TMW;
    const floatx4 zero = {0.f, 0.f, 0.f, 0.f};
    const __bf16* g1h = reinterpret_cast<const __bf16*>(d.w1_hi);
    const __bf16* g1l = reinterpret_cast<const __bf16*>(d.w1_lo);
    const __bf16* g2h = reinterpret_cast<const __bf16*>(d.w2_hi);
    const __bf16* g2l = reinterpret_cast<const __bf16*>(d.w2_lo);

    WA wa, wq;
    wa.issue(g1h, g1l, C, 0, tid);
    wq.issue(g2h, g2l, C, 0, tid);
    const size_t tap_o = (size_t)(tid < 200 ? (tid >> 3) : 0) * d.Hd + 4 * c4;     // threads >= 200 load tap 0 again and drop it

    for (int i = tid; i < plane_px * (DC_PS / 4); i += NTHR) reinterpret_cast<floatx4*>(P1)[i] = zero;

    bf16x8 xh[TMW][KS], xl[TMW][KS], gh[TMW][KS], gl[TMW][KS];
#pragma unroll
    for (int i = 0; i < TMW; ++i) {
        const size_t gp = pix0 + wb + i * 32 + lrow;
        const size_t n = gp >> gm.lhw;
        // K cotangents per forward row (act_rep = K): the cell is recomputed from the forward's x, row n / K
        const size_t xp = d.act_rep > 1 ? (((n / (size_t)d.act_rep) << gm.lhw) + (gp & (size_t)(gm.HW - 1))) : gp;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int k = ks * 16 + 8 * lh;
            const float* p = d.x + xp * C + k;
            split8(*reinterpret_cast<const floatx4*>(p), *reinterpret_cast<const floatx4*>(p + 4), xh[i][ks], xl[i][ks]);
            const float* q = d.dout + gp * C + k;
            const float* s = d.pro_scale + n * C + k;
            const float* t = d.pro_shift + n * C + k;
            const floatx4 a = *reinterpret_cast<const floatx4*>(q) * *reinterpret_cast<const floatx4*>(s) + *reinterpret_cast<const floatx4*>(t);
            const floatx4 b = *reinterpret_cast<const floatx4*>(q + 4) * *reinterpret_cast<const floatx4*>(s + 4) + *reinterpret_cast<const floatx4*>(t + 4);
            split8(a, b, gh[i][ks], gl[i][ks]);
        }
    }
    const int p0 = tile_pixel<BH, BW>(strip, gm);
    const int ctr = plane_idx(p0, gm);
    const int win = ctr - 2 * gm.PW - 2;
    int prow[TMW][4];
#pragma unroll
    for (int i = 0; i < TMW; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) prow[i][q] = plane_idx(wb + i * 32 + 8 * q + 4 * lh, gm) * DC_PS + lrow;

    const int nch = d.Hd / DC_CH;
    DC_T0
#pragma unroll 1
    for (int ch = 0; ch < nch; ++ch) {
        const int h0 = ch * DC_CH;
        wa.store(W1h, W1l, tid);
        wq.store(W2h, W2l, tid);
        __syncthreads();
        DC_T(0)
        // the small loads first: a wait on them must not wait on the weight prefetch behind them (vmcnt retires in order).  The
        // chunk's depthwise taps travel during phase (a) only and reach LDS in front of its closing barrier: eight registers
        // that are free in phases (b) .. (f), where the kernel is at its register limit
        const float b1v = d.b1[h0 + lrow];
        const floatx4 bd4 = *reinterpret_cast<const floatx4*>(d.bd + h0 + 4 * c4);
        const floatx4 taps = *reinterpret_cast<const floatx4*>(d.wd + tap_o + h0);
        const floatx4 tapsT = *reinterpret_cast<const floatx4*>(d.wd_bwd + tap_o + h0);
        __builtin_amdgcn_sched_barrier(0);
        // ---- (a) recompute t1c -> P4, silu(t1c) -> P1
        floatx16 t1[TMW];
#pragma unroll
        for (int i = 0; i < TMW; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) t1[i][r] = 0.f;
        gemm_resident<TMW, KS>(t1, xh, xl, W1h + lrow * WA::PITCH + 8 * lh, W1l + lrow * WA::PITCH + 8 * lh);
#pragma unroll
        for (int i = 0; i < TMW; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wb + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                const float v = t1[i][r] + b1v;
                const float sg = fast_sigmoid(v);                       // one sigmoid serves SiLU (now) and SiLU' (step f)
                P4[row * DC_PS + lrow] = sg * (1.0f + v * (1.0f - sg));
                P1[prow[i][r >> 2] + (r & 3) * DC_PS] = v * sg;
            }
        if (tid < 200) {
            *reinterpret_cast<floatx4*>(wS + (tid >> 3) * DC_CH + 4 * c4) = taps;
            *reinterpret_cast<floatx4*>(wT + (tid >> 3) * DC_CH + 4 * c4) = tapsT;
        }
        DC_T(1)
        __syncthreads();
        DC_T(2)
        // ---- (b) t2c = dw5(silu(t1c)) + bd ;  (c) silu'(t2c) -> registers  ||  g = dt3 . W2c^T, one scheduling region per strip pixel
        floatx4 g2[SW];
        dw_tile<BH, BW>(g2, P1 + win * DC_PS + 4 * c4, wS + 4 * c4, gm.PW);
        DC_T(3)
        floatx16 g[TMW];
#pragma unroll
        for (int i = 0; i < TMW; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) g[i][r] = 0.f;
        {
            const __bf16* w2h = W2h + lrow * WA::PITCH + 8 * lh;
            const __bf16* w2l = W2l + lrow * WA::PITCH + 8 * lh;
            constexpr int KPJ = KS / SW;
            bf16x8 bh[KPJ], bl[KPJ];
#pragma unroll
            for (int k = 0; k < KPJ; ++k) {
                bh[k] = *reinterpret_cast<const bf16x8*>(w2h + k * 16);
                bl[k] = *reinterpret_cast<const bf16x8*>(w2l + k * 16);
            }
#pragma unroll
            for (int j = 0; j < SW; ++j) {
                bf16x8 nh[KPJ], nl[KPJ];
                if (j + 1 < SW) {
#pragma unroll
                    for (int k = 0; k < KPJ; ++k) {
                        nh[k] = *reinterpret_cast<const bf16x8*>(w2h + ((j + 1) * KPJ + k) * 16);
                        nl[k] = *reinterpret_cast<const bf16x8*>(w2l + ((j + 1) * KPJ + k) * 16);
                    }
                }
#pragma unroll
                for (int k = 0; k < KPJ; ++k)
#pragma unroll
                    for (int i = 0; i < TMW; ++i) {
                        g[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gl[i][j * KPJ + k], bh[k], g[i], 0, 0, 0);
                        g[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gh[i][j * KPJ + k], bl[k], g[i], 0, 0, 0);
                        g[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gh[i][j * KPJ + k], bh[k], g[i], 0, 0, 0);
                    }
#pragma unroll
                for (int e = 0; e < 4; ++e) g2[j][e] = dsilu_f(g2[j][e] + bd4[e]);
                interleave_mfma_valu<KPJ * TMW * 3, 44>();
                __builtin_amdgcn_sched_barrier(0);
                if (j + 1 < SW) {
#pragma unroll
                    for (int k = 0; k < KPJ; ++k) { bh[k] = nh[k]; bl[k] = nl[k]; }
                }
            }
        }
        DC_T(4)
        __syncthreads();                // every strip is done reading silu(t1c)
        DC_T(5)
        {   // the next chunk's weights fly during phases (d) .. (f) and land in LDS at the top of the next chunk: their staging
            // registers are free in (a) .. (c), where the kernel is at its register limit.  Straight-line code (the last chunk
            // prefetches itself again): behind a branch the compiler can no longer count the loads in flight
            const int h1 = min(h0 + DC_CH, (nch - 1) * DC_CH);
            wa.issue(g1h, g1l, C, h1, tid);
            wq.issue(g2h, g2l, C, h1, tid);
        }
        // ---- (d) the plane now carries W2c^T dt3 ...
#pragma unroll
        for (int i = 0; i < TMW; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) P1[prow[i][r >> 2] + (r & 3) * DC_PS] = g[i][r];
        __syncthreads();
        DC_T(6)
        // ---- (e) ... times silu'(t2c), each thread on its own strip
#pragma unroll
        for (int j = 0; j < SW; ++j) {
            floatx4* q = reinterpret_cast<floatx4*>(P1 + (ctr + (j / BW) * gm.PW + (j % BW)) * DC_PS + 4 * c4);
            *q = *q * g2[j];
        }
        __syncthreads();
        DC_T(7)
        // ---- (f) dw5^T, times silu'(t1c) -> dt1
        {
            floatx4 a[SW];
            dw_tile<BH, BW>(a, P1 + win * DC_PS + 4 * c4, wT + 4 * c4, gm.PW);
#pragma unroll
            for (int j = 0; j < SW; ++j) {
                const int pj = p0 + (j / BW) * gm.W + (j % BW);
                const floatx4 u = *reinterpret_cast<const floatx4*>(P4 + pj * DC_PS + 4 * c4);
                *reinterpret_cast<floatx4*>(d.y + (pix0 + pj) * d.Hd + h0 + 4 * c4) = a[j] * u;
            }
        }
        DC_T(8)
        __syncthreads();                // P1, P4, the weight buffers and the taps are rewritten by the next chunk
        DC_T(9)
    }
    DC_TEND
}

static int log2_exact(int v) {
    int l = 0;
    while ((1 << l) < v) ++l;
    return (1 << l) == v ? l : -1;
}

constexpr size_t DC_LDS_MAX = 160 * 1024;

// M pixels per workgroup for channel count C (0: no kernel)
static int dc_tile_pixels(int C) { return C == 128 ? 256 : C == 256 ? 128 : 0; }

static size_t dc_lds_bytes(int C, int M, int H, int W, bool bwd) {
    const size_t plane_px = (size_t)(M / (H * W)) * (H + 4) * (W + 4);
    const size_t rows32 = (size_t)2 * 32 * (C + 8) * 2, cols32 = (size_t)2 * C * 40 * 2;      // one staged weight chunk, hi + lo
    return bwd ? (size_t)(50 * DC_CH + plane_px * DC_PS + (size_t)M * DC_PS) * 4 + 2 * rows32
               : (size_t)(25 * DC_CH + plane_px * DC_PS) * 4 + (size_t)2 * M * DC_LDB * 2 + rows32 + cols32;
}

template <int C, int TMW, int NW>
static int launch_dec_cell(const ga_dec_cell_desc& d, const dc_geom& gm, hipStream_t stream) {
    constexpr int M = 32 * TMW * NW;
    const size_t lds = dc_lds_bytes(C, M, d.H, d.W, d.backward != 0);
    const dim3 grid((unsigned)((size_t)d.N * d.H * d.W / M));
    if (d.backward) {
        static dyn_lds_cache attr;
        if (!ensure_dyn_lds(attr, reinterpret_cast<const void*>(&dec_cell_bwd_kernel<C, TMW, NW>), lds)) return GA_E_LAUNCH;
        hipLaunchKernelGGL((dec_cell_bwd_kernel<C, TMW, NW>), grid, dim3(64 * NW), lds, stream, d, gm);
    } else {
        static dyn_lds_cache attr;
        if (!ensure_dyn_lds(attr, reinterpret_cast<const void*>(&dec_cell_fwd_kernel<C, TMW, NW>), lds)) return GA_E_LAUNCH;
        hipLaunchKernelGGL((dec_cell_fwd_kernel<C, TMW, NW>), grid, dim3(64 * NW), lds, stream, d, gm);
    }
    return check_launch();
}

}  // namespace ga

#ifdef GA_DC_TRACE
extern "C" int ga_debug_dc_trace_read(unsigned long long* out, int n) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(ga::ga_dc_trace_buf), (size_t)n * 8) == hipSuccess ? GA_OK : GA_E_LAUNCH;
}
#endif

extern "C" int ga_dec_cell_supported(int N, int H, int W, int C, int Hd) {
    using namespace ga;
    const int M = dc_tile_pixels(C);
    if (!M || N <= 0 || H <= 0 || W <= 0 || Hd <= 0 || Hd % DC_CH) return 0;
    if (log2_exact(W) < 0 || log2_exact(H) < 0) return 0;
    const long HW = (long)H * W;
    if (HW > M || M % HW || W % (M / 32)) return 0;          // whole images per workgroup, strips inside one image row
    if (C == 128 && H % 2) return 0;                         // the 128-channel kernel's depthwise outputs are 2 x 4 blocks (4-wave form)
    if (((long)N * HW) % M) return 0;
    return dc_lds_bytes(C, M, H, W, true) <= DC_LDS_MAX && dc_lds_bytes(C, M, H, W, false) <= DC_LDS_MAX;
}

extern "C" int ga_dec_cell(const ga_dec_cell_desc* dp, void* stream_) {
    ga::clear_stale_error();
    using namespace ga;
    if (!dp) return GA_E_BADARG;
    const ga_dec_cell_desc& d = *dp;
    if (!d.x || !d.w1_hi || !d.w1_lo || !d.b1 || !d.wd || !d.bd || !d.w2_hi || !d.w2_lo || !d.y) return GA_E_BADARG;
    if (d.backward ? (!d.dout || !d.pro_scale || !d.pro_shift || !d.wd_bwd) : !d.b2) return GA_E_BADARG;
    if (!ga_dec_cell_supported(d.N, d.H, d.W, d.C, d.Hd)) return GA_E_UNSUPPORTED;
    if (d.act_rep > 1 && (!d.backward || d.N % d.act_rep)) return GA_E_BADARG;
    const void* ptrs[] = {d.x, d.w1_hi, d.w1_lo, d.wd, d.wd_bwd, d.bd, d.w2_hi, d.w2_lo, d.dout, d.pro_scale, d.pro_shift, d.y};
    for (const void* p : ptrs) if (p && !aligned16(p)) return GA_E_ALIGN;
    dc_geom gm;
    gm.W = d.W; gm.HW = d.H * d.W; gm.lw = log2_exact(d.W); gm.lhw = log2_exact(gm.HW); gm.PW = d.W + 4; gm.PH = d.H + 4;
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    if (d.C == 128) return d.variant == 1 ? launch_dec_cell<128, 1, 8>(d, gm, stream) : launch_dec_cell<128, 2, 4>(d, gm, stream);
    if (d.C == 256) return launch_dec_cell<256, 1, 4>(d, gm, stream);
    return GA_E_UNSUPPORTED;
}
