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
