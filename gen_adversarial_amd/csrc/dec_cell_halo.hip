// ga_dec_cell_halo — the residual branch of NVAE's ResidualCellDecoder (NVAE/modules/architecture.py:139-186, BatchNorms folded:
// 1x1 (C -> Hd) -> SiLU -> depthwise 5x5 -> SiLU -> 1x1 (Hd -> C)) for the FEW-CHANNEL cells whose images are larger than a
// workgroup (the post-processing cells: 32 channels at 64 x 64, 64 at 32 x 32).  Unfused they are the purest HBM work of the plan:
// the two Hd-wide tensors are 3 - 6x the cell's input and cross HBM ten times per cell.  Here a workgroup owns an 8 x 16 pixel
// tile and RECOMPUTES the expand conv on the tile's halo (K = C is tiny): the Hd-wide tensors never leave the CU.
//
// Forward, per 32-channel chunk of the hidden width:
//   GEMM1  t1c on the 12 x 20 window (tile + 2-pixel ring; 240 of 256 MFMA rows)  x resident as split-bf16 A-fragments
//   SiLU   -> fp32 LDS plane [12 x 20][32 + 8], pixels outside the image written as 0 (the depthwise conv's zero padding)
//   dw5    on the 8 x 16 tile, thread = (channel quad, strip of 4 pixels) -> + bd -> SiLU -> split-bf16 LDS planes
//   GEMM2  acc [128 x C] += s2c [128 x 32] . W2c
// Backward (d x from d t3 in ONE launch: with few channels the d x accumulator fits beside the operands): t1c is recomputed on the
// 16 x 24 window (tile + 4), t2c and dt2c = (dt3 . W2c^T) * SiLU'(t2c) on the 12 x 20 window, dw5^T on the tile, * SiLU'(t1c),
// and dx [128 x C] += dt1c [128 x 32] . W1c;  dx (+ addends) is written once.
// Contractions are the three-MFMA split-bf16 products of conv_bf3 / ga_dec_cell (same operand split), the depthwise part the fp32
// loop of dwconv5 in the same tap order.
#include "ga_common.h"
#include "dec_cell_common.h"

namespace ga {

constexpr int HT_H = 8, HT_W = 16;                  // output tile
constexpr int HW1_H = HT_H + 4, HW1_W = HT_W + 4;   // window of ring 1 (12 x 20)
constexpr int HW2_H = HT_H + 8, HW2_W = HT_W + 8;   // window of ring 2 (16 x 24), backward only

// one 32-channel chunk of a split-bf16 weight matrix through registers into LDS, for any small C (pieces of 16 B; 4 C per array).
// ROWS32: rows h0 .. h0+31 of a [Hd][C] matrix, LDS pitch C + 8;  otherwise columns h0 .. h0+31 of a [C][Hd] matrix, pitch 40.
template <int C, bool ROWS32>
struct w_small {
    static constexpr int NPC = 4 * C, NP = (NPC + 255) / 256;
    static constexpr int PITCH = ROWS32 ? C + 8 : 40;
    static constexpr int ELEMS = (ROWS32 ? 32 : C) * PITCH;
    uintx4 hi[NP], lo[NP];
    __device__ __forceinline__ void issue(const __bf16* gh, const __bf16* gl, const int ld, const int h0, const int tid) {
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            const int q = min(tid + 256 * k, NPC - 1);
            const size_t o = ROWS32 ? (size_t)(h0 + q / (C / 8)) * ld + (q % (C / 8)) * 8 : (size_t)(q >> 2) * ld + h0 + (q & 3) * 8;
            hi[k] = *reinterpret_cast<const uintx4*>(gh + o);
            lo[k] = *reinterpret_cast<const uintx4*>(gl + o);
        }
    }
    __device__ __forceinline__ void store(__bf16* sh, __bf16* sl, const int tid) const {
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            const int q = tid + 256 * k;
            if (q < NPC) {
                const int o = ROWS32 ? (q / (C / 8)) * PITCH + (q % (C / 8)) * 8 : (q >> 2) * PITCH + (q & 3) * 8;
                *reinterpret_cast<uintx4*>(sh + o) = hi[k];
                *reinterpret_cast<uintx4*>(sl + o) = lo[k];
            }
        }
    }
};

struct hc_geom { int tiles_x, tiles_per_img; };

// -DGA_HC_TRACE (make hctrace -> libga_ops_hctrace.so, tools/dec_cell_halo_trace.py): shader-clock sums per phase, lane 0 of every
// wave of workgroup 0; slot 15 = the whole kernel
#ifdef GA_HC_TRACE
__device__ unsigned long long ga_hc_trace_buf[4 * 16];
#define HC_T0 unsigned long long tsum[16] = {}; unsigned long long tprev = __builtin_amdgcn_s_memtime(); const unsigned long long tstart = tprev;
#define HC_T(i) { const unsigned long long tn = __builtin_amdgcn_s_memtime(); tsum[i] += tn - tprev; tprev = tn; }
#define HC_TEND { tsum[15] = __builtin_amdgcn_s_memtime() - tstart; if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) { for (int i = 0; i < 16; ++i) ga_hc_trace_buf[(threadIdx.x >> 6) * 16 + i] = tsum[i]; } }
#else
#define HC_T0
#define HC_T(i)
#define HC_TEND
#endif

// ---------------------------------------------------------------------------------------------------------------------
template <int C>
__global__ void __launch_bounds__(256, 2) dec_cell_halo_fwd_kernel(const ga_dec_cell_halo_desc d, const hc_geom gm) {
    constexpr int KS = C / 16, NT = C / 32, SW = 4;
    using WA = w_small<C, true>;
    using WB = w_small<C, false>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* wS = smem;                                                   // [25][32] taps of the chunk
    float* P1 = smem + 25 * DC_CH;                                      // fp32 plane of the 12 x 20 window
    __bf16* P2h = reinterpret_cast<__bf16*>(P1 + HW1_H * HW1_W * DC_PS);    // [128][DC_LDB] bf16, hi then lo
    __bf16* P2l = P2h + 128 * DC_LDB;
    __bf16* W1h = P2l + 128 * DC_LDB;
    __bf16* W1l = W1h + WA::ELEMS;
    __bf16* W2h = W1l + WA::ELEMS;
    __bf16* W2l = W2h + WB::ELEMS;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lrow = lane & 31, lh = lane >> 5;
    const int c4 = tid & 7, strip = tid >> 3;
    const int n = blockIdx.x / gm.tiles_per_img, tl = blockIdx.x - n * gm.tiles_per_img;
    const int y0 = (tl / gm.tiles_x) * HT_H, x0 = (tl % gm.tiles_x) * HT_W;
    const float* xn = d.x + (size_t)n * d.H * d.W * C;
    const __bf16* g1h = reinterpret_cast<const __bf16*>(d.w1_hi);
    const __bf16* g1l = reinterpret_cast<const __bf16*>(d.w1_lo);
    const __bf16* g2h = reinterpret_cast<const __bf16*>(d.w2_hi);
    const __bf16* g2l = reinterpret_cast<const __bf16*>(d.w2_lo);
    const floatx4 zero = {0.f, 0.f, 0.f, 0.f};
    const int nch = d.Hd / DC_CH;
    const int tap_o = (tid < 200 ? (tid >> 3) : 0) * d.Hd + 4 * c4;

    WA wa;
    WB wq;
    wa.issue(g1h, g1l, C, 0, tid);
    wq.issue(g2h, g2l, d.Hd, 0, tid);
    floatx4 taps = *reinterpret_cast<const floatx4*>(d.wd + tap_o);

    // x of the window as resident A-fragments: wave w owns window pixels 64 w .. 64 w + 63 (two 32-row MFMA tiles)
    bf16x8 xh[2][KS], xl[2][KS];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int p = wave * 64 + i * 32 + lrow;
        const int wy = p / HW1_W, wx = p - wy * HW1_W;
        const int gy = y0 - 2 + wy, gx = x0 - 2 + wx;
        const bool ok = p < HW1_H * HW1_W && gy >= 0 && gy < d.H && gx >= 0 && gx < d.W;
        const float* px = xn + ((size_t)(ok ? gy : 0) * d.W + (ok ? gx : 0)) * C + 8 * lh;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const floatx4 a = ok ? *reinterpret_cast<const floatx4*>(px + ks * 16) : zero;
            const floatx4 b = ok ? *reinterpret_cast<const floatx4*>(px + ks * 16 + 4) : zero;
            split8(a, b, xh[i][ks], xl[i][ks]);
        }
    }
    // accumulator rows of the two tiles (row = 8 q + 4 lh + j): bit (4 q + j) of live[i] = the window pixel is inside the image;
    // pixels outside hold 0 in the plane (the depthwise conv pads SiLU(t1) with zeros), rows >= 240 are not written
    unsigned live[2] = {0u, 0u}, inwin[2] = {0u, 0u};
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int p = wave * 64 + i * 32 + 8 * (r >> 2) + 4 * lh + (r & 3);
            const int wy = p / HW1_W, wx = p - wy * HW1_W;
            const int gy = y0 - 2 + wy, gx = x0 - 2 + wx;
            if (p < HW1_H * HW1_W) {
                inwin[i] |= 1u << r;
                if (gy >= 0 && gy < d.H && gx >= 0 && gx < d.W) live[i] |= 1u << r;
            }
        }

    floatx16 acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    // this thread's strip of 4 tile pixels: consecutive strips are vertically adjacent rows (dec_cell_common.h, strip_pixel)
    const int oy = strip & 7, ox = (strip >> 3) * SW;
    const float* win = P1 + (oy * HW1_W + ox) * DC_PS + 4 * c4;             // top-left of the strip's 5 x 8 window
    const __bf16* w1h = W1h + lrow * WA::PITCH + 8 * lh;
    const __bf16* w1l = W1l + lrow * WA::PITCH + 8 * lh;
    const __bf16* w2h = W2h + lrow * WB::PITCH + 8 * lh;
    const __bf16* w2l = W2l + lrow * WB::PITCH + 8 * lh;

#pragma unroll 1
    for (int ch = 0; ch < nch; ++ch) {
        const int h0 = ch * DC_CH;
        wa.store(W1h, W1l, tid);
        wq.store(W2h, W2l, tid);
        if (tid < 200) *reinterpret_cast<floatx4*>(wS + (tid >> 3) * DC_CH + 4 * c4) = taps;
        __syncthreads();
        const float b1v = d.b1[h0 + lrow];
        const floatx4 bd4 = *reinterpret_cast<const floatx4*>(d.bd + h0 + 4 * c4);
        {   // next chunk's weights (the last chunk prefetches itself again: straight-line code)
            const int h1 = min(h0 + DC_CH, (nch - 1) * DC_CH);
            wa.issue(g1h, g1l, C, h1, tid);
            wq.issue(g2h, g2l, d.Hd, h1, tid);
            taps = *reinterpret_cast<const floatx4*>(d.wd + tap_o + h1);
        }
        // ---- GEMM1 on the window, SiLU -> plane
        floatx16 t1[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) t1[i][r] = 0.f;
        gemm_resident<2, KS>(t1, xh, xl, w1h, w1l);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int p = wave * 64 + i * 32 + 8 * (r >> 2) + 4 * lh + (r & 3);
                if ((inwin[i] >> r) & 1u) P1[p * DC_PS + lrow] = ((live[i] >> r) & 1u) ? silu_f(t1[i][r] + b1v) : 0.f;
            }
        __syncthreads();
        // ---- depthwise 5x5 on the tile, SiLU, split -> P2
        {
            floatx4 a[SW];
            dw_strip<SW>(a, win, wS + 4 * c4, HW1_W);
#pragma unroll
            for (int j = 0; j < SW; ++j) {
                floatx4 v = a[j] + bd4;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = silu_f(v[e]);
                const bf16x4 hi = __builtin_convertvector(v, bf16x4);
                const bf16x4 lo = __builtin_convertvector(v - __builtin_convertvector(hi, floatx4), bf16x4);
                const int ip = oy * HT_W + ox + j;
                *reinterpret_cast<bf16x4*>(P2h + ip * DC_LDB + 4 * c4) = hi;
                *reinterpret_cast<bf16x4*>(P2l + ip * DC_LDB + 4 * c4) = lo;
            }
        }
        __syncthreads();
        // ---- GEMM2: this wave's 32 tile pixels x all C output channels
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int o = (wave * 32 + lrow) * DC_LDB + ks * 16 + 8 * lh;
            const bf16x8 ah = *reinterpret_cast<const bf16x8*>(P2h + o);
            const bf16x8 al = *reinterpret_cast<const bf16x8*>(P2l + o);
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const bf16x8 bh = *reinterpret_cast<const bf16x8*>(w2h + j * 32 * WB::PITCH + ks * 16);
                const bf16x8 bl = *reinterpret_cast<const bf16x8*>(w2l + j * 32 * WB::PITCH + ks * 16);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[j], 0, 0, 0);
            }
        }
        __syncthreads();                // the weight buffers, the taps and both planes are rewritten by the next chunk
    }
    // ---- t3 = acc + b2
    float* yn = d.y + (size_t)n * d.H * d.W * C;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const float b2v = d.b2[j * 32 + lrow];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ip = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            const int gy = y0 + (ip >> 4), gx = x0 + (ip & 15);
            yn[((size_t)gy * d.W + gx) * C + j * 32 + lrow] = acc[j][r] + b2v;
        }
    }
}

static size_t hc_lds_fwd(int C) {
    return (size_t)(25 * DC_CH + HW1_H * HW1_W * DC_PS) * 4 + (size_t)2 * 128 * DC_LDB * 2 + (size_t)2 * 32 * (C + 8) * 2 + (size_t)2 * C * 40 * 2;
}

template <int C>
static int launch_hc_fwd(const ga_dec_cell_halo_desc& d, const hc_geom& gm, hipStream_t stream) {
    const size_t lds = hc_lds_fwd(C);
    static dyn_lds_cache attr;
    (void)ensure_dyn_lds(attr, reinterpret_cast<const void*>(&dec_cell_halo_fwd_kernel<C>), lds);
    hipLaunchKernelGGL((dec_cell_halo_fwd_kernel<C>), dim3((unsigned)(d.N * gm.tiles_per_img)), dim3(256), lds, stream, d, gm);
    return check_launch();
}

// ---------------------------------------------------------------------------------------------------------------------
// Backward.  Rings: the tile (8 x 16), ring 1 = tile + 2 (12 x 20: where t2 / dt2 are needed), ring 2 = tile + 4 (16 x 24: where
// SiLU(t1) is needed).  One fp32 plane PA over ring 2 carries SiLU(t1c), then (ring-1 positions) W2c^T dt3 and dt2c.  Per chunk:
//   (a) GEMM1 on ring 2: SiLU(t1c) -> PA (0 outside the image), SiLU'(t1c) of the tile -> P4
//   (b) t2c = dw5(PA) + bd on ring 1 (two strips of 4 per thread): SiLU'(t2c) in registers (0 outside the image)
//   (c) GEMM3 on ring 1: g = dt3 . W2c^T -> PA          (d) own strips *= SiLU'(t2c)
//   (e) dw5^T on the tile, * P4 -> split-bf16 planes     (f) GEMM4: dx [128 x C] += dt1c . W1c
template <int C>
__global__ void __launch_bounds__(256, 1) dec_cell_halo_bwd_kernel(const ga_dec_cell_halo_desc d, const hc_geom gm) {
    constexpr int KS = C / 16, NT = C / 32, SW = 4;
    constexpr int NP1 = HW1_H * HW1_W, NP2 = HW2_H * HW2_W;            // 240, 384 window pixels
    using WA = w_small<C, true>;
    using WB = w_small<C, false>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* wS = smem;                                                   // [25][32] forward taps of the chunk
    float* wT = smem + 25 * DC_CH;                                      // [25][32] flipped taps
    float* PA = smem + 50 * DC_CH;                                      // fp32 plane over ring 2
    float* P4 = PA + NP2 * DC_PS;                                       // [128][DC_PS] SiLU'(t1c) of the tile
    __bf16* P2h = reinterpret_cast<__bf16*>(P4 + 128 * DC_PS);          // [128][DC_LDB] dt1c, hi then lo
    __bf16* P2l = P2h + 128 * DC_LDB;
    __bf16* W1h = P2l + 128 * DC_LDB;                                   // 32 rows of W1, 32 rows of W2^T, 32 columns of W1^T
    __bf16* W1l = W1h + WA::ELEMS;
    __bf16* W2h = W1l + WA::ELEMS;
    __bf16* W2l = W2h + WA::ELEMS;
    __bf16* W3h = W2l + WA::ELEMS;
    __bf16* W3l = W3h + WB::ELEMS;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lrow = lane & 31, lh = lane >> 5;
    const int c4 = tid & 7, strip = tid >> 3;
    const int n = blockIdx.x / gm.tiles_per_img, tl = blockIdx.x - n * gm.tiles_per_img;
    const int y0 = (tl / gm.tiles_x) * HT_H, x0 = (tl % gm.tiles_x) * HT_W;
    const size_t img = (size_t)n * d.H * d.W;
    const __bf16* g1h = reinterpret_cast<const __bf16*>(d.w1_hi);
    const __bf16* g1l = reinterpret_cast<const __bf16*>(d.w1_lo);
    const __bf16* g2h = reinterpret_cast<const __bf16*>(d.w2_hi);
    const __bf16* g2l = reinterpret_cast<const __bf16*>(d.w2_lo);
    const __bf16* g3h = reinterpret_cast<const __bf16*>(d.w1t_hi);
    const __bf16* g3l = reinterpret_cast<const __bf16*>(d.w1t_lo);
    const floatx4 zero = {0.f, 0.f, 0.f, 0.f};
    const int nch = d.Hd / DC_CH;
    const size_t tap_o = (size_t)(tid < 200 ? (tid >> 3) : 0) * d.Hd + 4 * c4;

    HC_T0
    WA wa, wq;
    WB wr;
    wa.issue(g1h, g1l, C, 0, tid);
    wq.issue(g2h, g2l, C, 0, tid);
    wr.issue(g3h, g3l, d.Hd, 0, tid);
    floatx4 taps = *reinterpret_cast<const floatx4*>(d.wd + tap_o);
    floatx4 tapsT = *reinterpret_cast<const floatx4*>(d.wd_bwd + tap_o);

    auto inside = [&](const int gy, const int gx) { return gy >= 0 && gy < d.H && gx >= 0 && gx < d.W; };

    // resident A-fragments: x on ring 2 (three 32-row tiles per wave), dt3 = dout * ps[n] + pb[n] on ring 1 (two per wave)
    bf16x8 xh[3][KS], xl[3][KS], gh[2][KS], gl[2][KS];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int p = wave * 96 + i * 32 + lrow;
        const int wy = p / HW2_W, wx = p - wy * HW2_W;
        const int gy = y0 - 4 + wy, gx = x0 - 4 + wx;
        const bool ok = inside(gy, gx);
        const float* px = d.x + (img + (size_t)(ok ? gy : 0) * d.W + (ok ? gx : 0)) * C + 8 * lh;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const floatx4 a = ok ? *reinterpret_cast<const floatx4*>(px + ks * 16) : zero;
            const floatx4 b = ok ? *reinterpret_cast<const floatx4*>(px + ks * 16 + 4) : zero;
            split8(a, b, xh[i][ks], xl[i][ks]);
        }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int p = wave * 64 + i * 32 + lrow;
        const int wy = p / HW1_W, wx = p - wy * HW1_W;
        const int gy = y0 - 2 + wy, gx = x0 - 2 + wx;
        const bool ok = p < NP1 && inside(gy, gx);
        const float* pq = d.dout + (img + (size_t)(ok ? gy : 0) * d.W + (ok ? gx : 0)) * C + 8 * lh;
        const float* ps = d.pro_scale + (size_t)n * C + 8 * lh;
        const float* pb = d.pro_shift + (size_t)n * C + 8 * lh;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const floatx4 a = ok ? *reinterpret_cast<const floatx4*>(pq + ks * 16) * *reinterpret_cast<const floatx4*>(ps + ks * 16) +
                                   *reinterpret_cast<const floatx4*>(pb + ks * 16) : zero;
            const floatx4 b = ok ? *reinterpret_cast<const floatx4*>(pq + ks * 16 + 4) * *reinterpret_cast<const floatx4*>(ps + ks * 16 + 4) +
                                   *reinterpret_cast<const floatx4*>(pb + ks * 16 + 4) : zero;
            split8(a, b, gh[i][ks], gl[i][ks]);
        }
    }

    floatx16 acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    // accumulator rows r = 4 q + j of tile i sit at window pixel base + 8 q + 4 lh + j: four consecutive pixels of one window row
    // (both window widths are multiples of 4).  Per group: PA / P4 positions; per row: "inside the image" bits.
    unsigned live2[3] = {0u, 0u, 0u}, live1[2] = {0u, 0u};
    int p4i[3][4], pa1[2][4];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int p = wave * 96 + i * 32 + 8 * q + 4 * lh;
            const int wy = p / HW2_W, wx = p - wy * HW2_W;
            p4i[i][q] = (wy >= 4 && wy < 4 + HT_H && wx >= 4 && wx < 4 + HT_W) ? ((wy - 4) * HT_W + wx - 4) * DC_PS + lrow : -1;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (inside(y0 - 4 + wy, x0 - 4 + wx + j)) live2[i] |= 1u << (4 * q + j);
        }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int p = wave * 64 + i * 32 + 8 * q + 4 * lh;
            const int wy = p / HW1_W, wx = p - wy * HW1_W;
            pa1[i][q] = p < NP1 ? ((wy + 2) * HW2_W + wx + 2) * DC_PS + lrow : -1;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (p < NP1 && inside(y0 - 2 + wy, x0 - 2 + wx + j)) live1[i] |= 1u << (4 * q + j);
        }
    (void)live1;

    // (b) / (d): this thread's two ring-1 strips (strip s: row s / 5, columns 4 (s % 5) ..), (e): its tile strip (row s / 4).  The
    // two strips that share a 16-lane LDS phase are HORIZONTAL neighbours (4 pixels = 640 B = 128 B mod 256 B: all 64 banks); vertical
    // ones would be a whole plane row apart (24 pixels = 0 mod 256 B: an 8-way conflict on every window read)
    int s1[2];
    s1[0] = strip;
    s1[1] = strip + 32;                                                 // 60 strips: the last four thread groups have one
    const int oy = strip >> 2, ox = (strip & 3) * SW;
    const __bf16* w1h = W1h + lrow * WA::PITCH + 8 * lh;
    const __bf16* w1l = W1l + lrow * WA::PITCH + 8 * lh;
    const __bf16* w2h = W2h + lrow * WA::PITCH + 8 * lh;
    const __bf16* w2l = W2l + lrow * WA::PITCH + 8 * lh;
    const __bf16* w3h = W3h + lrow * WB::PITCH + 8 * lh;
    const __bf16* w3l = W3l + lrow * WB::PITCH + 8 * lh;

    HC_T(0)
#pragma unroll 1
    for (int ch = 0; ch < nch; ++ch) {
        const int h0 = ch * DC_CH;
        wa.store(W1h, W1l, tid);
        wq.store(W2h, W2l, tid);
        wr.store(W3h, W3l, tid);
        if (tid < 200) {
            *reinterpret_cast<floatx4*>(wS + (tid >> 3) * DC_CH + 4 * c4) = taps;
            *reinterpret_cast<floatx4*>(wT + (tid >> 3) * DC_CH + 4 * c4) = tapsT;
        }
        __syncthreads();
        HC_T(1)
        const float b1v = d.b1[h0 + lrow];
        const floatx4 bd4 = *reinterpret_cast<const floatx4*>(d.bd + h0 + 4 * c4);
        {
            const int h1 = min(h0 + DC_CH, (nch - 1) * DC_CH);
            wa.issue(g1h, g1l, C, h1, tid);
            wq.issue(g2h, g2l, C, h1, tid);
            wr.issue(g3h, g3l, d.Hd, h1, tid);
            taps = *reinterpret_cast<const floatx4*>(d.wd + tap_o + h1);
            tapsT = *reinterpret_cast<const floatx4*>(d.wd_bwd + tap_o + h1);
        }
        // ---- (a) t1c on ring 2
        {
            floatx16 t1[3];
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) t1[i][r] = 0.f;
            gemm_resident<3, KS>(t1, xh, xl, w1h, w1l);
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int p = wave * 96 + i * 32 + 8 * (r >> 2) + 4 * lh + (r & 3);
                    const float v = t1[i][r] + b1v;
                    const float sg = fast_sigmoid(v);
                    PA[p * DC_PS + lrow] = ((live2[i] >> r) & 1u) ? v * sg : 0.f;
                    if (p4i[i][r >> 2] >= 0) P4[p4i[i][r >> 2] + (r & 3) * DC_PS] = sg * (1.0f + v * (1.0f - sg));
                }
        }
        HC_T(2)
        __syncthreads();
        HC_T(3)
        // ---- (b) SiLU'(t2c) on ring 1, in registers
        floatx4 g2[2][SW];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            if (s1[k] < 60) {
                const int wy = s1[k] / 5, wx0 = (s1[k] % 5) * SW;
                dw_strip<SW>(g2[k], PA + (wy * HW2_W + wx0) * DC_PS + 4 * c4, wS + 4 * c4, HW2_W);
#pragma unroll
                for (int j = 0; j < SW; ++j) {
                    const bool ok = inside(y0 - 2 + wy, x0 - 2 + wx0 + j);
#pragma unroll
                    for (int e = 0; e < 4; ++e) g2[k][j][e] = ok ? dsilu_f(g2[k][j][e] + bd4[e]) : 0.f;
                }
            }
        }
        HC_T(4)
        // ---- (c) g = dt3 . W2c^T on ring 1
        floatx16 gg[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) gg[i][r] = 0.f;
        gemm_resident<2, KS>(gg, gh, gl, w2h, w2l);
        HC_T(5)
        __syncthreads();                // every strip is done reading SiLU(t1c)
        HC_T(6)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (pa1[i][r >> 2] >= 0) PA[pa1[i][r >> 2] + (r & 3) * DC_PS] = gg[i][r];
        HC_T(7)
        __syncthreads();
        HC_T(8)
        // ---- (d) dt2c = g * SiLU'(t2c), each thread on its own strips
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            if (s1[k] < 60) {
                const int wy = s1[k] / 5, wx0 = (s1[k] % 5) * SW;
#pragma unroll
                for (int j = 0; j < SW; ++j) {
                    floatx4* q = reinterpret_cast<floatx4*>(PA + ((wy + 2) * HW2_W + wx0 + j + 2) * DC_PS + 4 * c4);
                    *q = *q * g2[k][j];
                }
            }
        }
        HC_T(9)
        __syncthreads();
        HC_T(10)
        // ---- (e) dt1c = dw5^T(dt2c) * SiLU'(t1c) on the tile -> split planes
        {
            floatx4 a[SW];
            dw_strip<SW>(a, PA + ((oy + 2) * HW2_W + ox + 2) * DC_PS + 4 * c4, wT + 4 * c4, HW2_W);
#pragma unroll
            for (int j = 0; j < SW; ++j) {
                const int ip = oy * HT_W + ox + j;
                const floatx4 v = a[j] * *reinterpret_cast<const floatx4*>(P4 + ip * DC_PS + 4 * c4);
                const bf16x4 hi = __builtin_convertvector(v, bf16x4);
                const bf16x4 lo = __builtin_convertvector(v - __builtin_convertvector(hi, floatx4), bf16x4);
                *reinterpret_cast<bf16x4*>(P2h + ip * DC_LDB + 4 * c4) = hi;
                *reinterpret_cast<bf16x4*>(P2l + ip * DC_LDB + 4 * c4) = lo;
            }
        }
        HC_T(11)
        __syncthreads();
        HC_T(12)
        // ---- (f) dx += dt1c . W1c
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int o = (wave * 32 + lrow) * DC_LDB + ks * 16 + 8 * lh;
            const bf16x8 ah = *reinterpret_cast<const bf16x8*>(P2h + o);
            const bf16x8 al = *reinterpret_cast<const bf16x8*>(P2l + o);
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const bf16x8 bh = *reinterpret_cast<const bf16x8*>(w3h + j * 32 * WB::PITCH + ks * 16);
                const bf16x8 bl = *reinterpret_cast<const bf16x8*>(w3l + j * 32 * WB::PITCH + ks * 16);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[j], 0, 0, 0);
            }
        }
        HC_T(13)
        __syncthreads();                // planes, weight buffers and taps are rewritten by the next chunk
        HC_T(14)
    }
    // ---- dx = addend + addend2 + acc
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ip = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            const size_t o = (img + (size_t)(y0 + (ip >> 4)) * d.W + x0 + (ip & 15)) * C + j * 32 + lrow;
            float v = acc[j][r];
            if (d.addend) v += d.addend[o];
            if (d.addend2) v += d.addend2[o];
            d.y[o] = v;
        }
    HC_TEND
}

static size_t hc_lds_bwd(int C) {
    return (size_t)(50 * DC_CH + HW2_H * HW2_W * DC_PS + 128 * DC_PS) * 4 + (size_t)2 * 128 * DC_LDB * 2 + (size_t)2 * 2 * 32 * (C + 8) * 2 +
           (size_t)2 * C * 40 * 2;
}

template <int C>
static int launch_hc_bwd(const ga_dec_cell_halo_desc& d, const hc_geom& gm, hipStream_t stream) {
    const size_t lds = hc_lds_bwd(C);
    static dyn_lds_cache attr;
    (void)ensure_dyn_lds(attr, reinterpret_cast<const void*>(&dec_cell_halo_bwd_kernel<C>), lds);
    hipLaunchKernelGGL((dec_cell_halo_bwd_kernel<C>), dim3((unsigned)(d.N * gm.tiles_per_img)), dim3(256), lds, stream, d, gm);
    return check_launch();
}

}  // namespace ga

#ifdef GA_HC_TRACE
extern "C" int ga_hc_trace_read(unsigned long long* out, int n) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(ga::ga_hc_trace_buf), (size_t)n * 8) == hipSuccess ? GA_OK : GA_E_LAUNCH;
}
#endif

extern "C" int ga_dec_cell_halo_supported(int N, int H, int W, int C, int Hd) {
    if (N <= 0 || (C != 32 && C != 64) || Hd <= 0 || Hd % 32) return 0;
    if (H < ga::HT_H || W < ga::HT_W || H % ga::HT_H || W % ga::HT_W) return 0;
    if ((long)N * (H / ga::HT_H) * (W / ga::HT_W) > 0x7fffffffL) return 0;
    return 1;
}

extern "C" int ga_dec_cell_halo(const ga_dec_cell_halo_desc* d, void* s) {
    ga::clear_stale_error();
    if (!d || !d->x || !d->w1_hi || !d->w1_lo || !d->b1 || !d->wd || !d->bd || !d->w2_hi || !d->w2_lo || !d->y) return GA_E_BADARG;
    if (d->Cin != d->Cout || d->up) return GA_E_UNSUPPORTED;
    if (!ga_dec_cell_halo_supported(d->N, d->H, d->W, d->Cin, d->Hd)) return GA_E_UNSUPPORTED;
    if (!ga::aligned16(d->x) || !ga::aligned16(d->y) || !ga::aligned16(d->w1_hi) || !ga::aligned16(d->w1_lo) || !ga::aligned16(d->w2_hi) ||
        !ga::aligned16(d->w2_lo) || !ga::aligned16(d->wd) || !ga::aligned16(d->bd)) return GA_E_ALIGN;
    ga::hc_geom gm;
    gm.tiles_x = d->W / ga::HT_W;
    gm.tiles_per_img = gm.tiles_x * (d->H / ga::HT_H);
    hipStream_t stream = (hipStream_t)s;
    if (!d->backward) {
        if (!d->b2) return GA_E_BADARG;
        return d->Cin == 32 ? ga::launch_hc_fwd<32>(*d, gm, stream) : ga::launch_hc_fwd<64>(*d, gm, stream);
    }
    if (!d->wd_bwd || !d->w1t_hi || !d->w1t_lo || !d->dout || !d->pro_scale || !d->pro_shift) return GA_E_BADARG;
    if (!ga::aligned16(d->dout) || !ga::aligned16(d->pro_scale) || !ga::aligned16(d->pro_shift) || !ga::aligned16(d->wd_bwd) ||
        !ga::aligned16(d->w1t_hi) || !ga::aligned16(d->w1t_lo) || (d->addend && !ga::aligned16(d->addend)) ||
        (d->addend2 && !ga::aligned16(d->addend2))) return GA_E_ALIGN;
    return d->Cin == 32 ? ga::launch_hc_bwd<32>(*d, gm, stream) : ga::launch_hc_bwd<64>(*d, gm, stream);
}
