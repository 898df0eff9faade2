// conv_thin3 — 3x3 / stride 1 / pad 1 convolutions with 32 (or 64) input channels on the split-bf16 matrix path (arithmetic: conv_bf3.hip) as a
// PERSISTENT, WEIGHTS-RESIDENT kernel on 2-D tiles (tile code 11 of ga_conv_desc.tile; round 4).
//
// Why: a 32-channel 3x3 layer has ONE 32-channel chunk, i.e. 9 tap steps of matrix work per 128-pixel tile (54 MFMAs per wave, ~1.7 K
// clocks) against ~20 K clocks of per-workgroup setup (slot tables, first patch round trip), weight staging (nine barriers) and staged
// epilogue in conv_halo3 — those layers (NVAE pre / post-processing at 64 x 64, StyleGAN2 at 1024 x 1024) ran at 120 - 150 TFLOP/s and
// ~2 TB/s, neither roofline.  Here a workgroup
//   * keeps ALL weights of its 32 output channels in LDS for its lifetime (9 taps x 32 x 32, split bf16, fragment order: 36 KB),
//   * walks a run of 8 x 16-pixel tiles (2-D: the halo window is 10 x 18 = 1.4x the tile; conv_halo3's row tiles read 2x at 64 px,
//     its row SEGMENTS of wide images 3x), setup paid once,
//   * requests tile t+1's window from memory before the MFMAs and the epilogue of tile t.
// Per tile: prologue + bf16 split of the window into LDS | barrier | 9 taps x 2 k steps x 3 MFMAs, A and B fragments both from LDS at
// immediate offsets | barrier | conv_epilogue through the window's LDS (bias, act', addends; 2-D row map) | barrier.
// Same operand split, k order and MFMA order as conv_halo3 (tap-major inside the chunk): results are BITWISE those of tile 7.
// Cout > 32: grid.y output-channel tiles, each staging the window itself (the layers this is for have Cout <= 32, or few tiles).
// 64 input channels (template parameter CIN): four k steps per tap (108 MFMAs per wave and tile), 74 KB of weight fragments and a
// 55 KB window: ONE workgroup per CU; the next window's 12 loads per thread are in flight under the MFMAs and the epilogue.
#include "ga_common.h"
#include "conv_epilogue.h"

namespace ga {

typedef __bf16 tk_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 tk_bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned tk_uintx4 __attribute__((ext_vector_type(4)));

constexpr int TK_TH = 8, TK_TW = 16;        // tile: 8 rows x 16 columns = 128 output pixels
constexpr int TK_PH = TK_TH + 2, TK_PW = TK_TW + 2, TK_P = TK_PH * TK_PW;      // window: 10 x 18 = 180 pixels
// per input-channel count CIN (32: two workgroups per CU; 64: one — 74 KB of weight fragments + a 55 KB window)
template <int CIN> struct thin_traits {
    static constexpr int KS = CIN / 16;                 // 16-deep k steps per tap
    static constexpr int Q = CIN / 4;                   // channel quads per pixel
    static constexpr int QSH = CIN == 32 ? 3 : 4;       // log2(Q)
    static constexpr int LDH = CIN + 8;                 // bf16 per window pixel (80 / 144 B: conflict-free 16-B fragment reads)
    // bf16 per window row: 18 * LDH rounded up to 0 mod 256 B (a tile row of 16 pixels then continues the bank pattern of the row above,
    // as 16 * LDH * 2 B = 0 mod 256 B does in a linear tile): 720 -> 768, 1296 -> 1408
    static constexpr int RS = CIN == 32 ? 768 : 1408;
    static constexpr int PLANE = TK_PH * RS;            // bf16 elements of one window plane (hi or lo): 15,360 / 28,160 B
    static constexpr int BFR = 9 * KS * 2 * 64 * 8;     // resident weight fragments [tap][k step][hi | lo][lane][8]: 36,864 / 73,728 B
    static constexpr int SLOTS = (TK_P * Q + 255) / 256;    // window float4 slots per thread: 6 / 12
    static constexpr int WGS_PER_CU = CIN == 32 ? 2 : 1;
    static constexpr size_t LDS = (size_t)(BFR + 2 * PLANE) * 2;       // 67,584 / 130,048 B
};

struct thin_geom { int H, W, tiles_x, tpi, ntiles, per; fastdiv fd_tpi, fd_tx; };

template <int CIN, int AFF, int ACT>
__global__ void __launch_bounds__(256, thin_traits<CIN>::WGS_PER_CU)
conv_thin3_kernel(const ga_conv_desc d, const thin_geom g, const int M) {
    using T = thin_traits<CIN>;
    constexpr int TK_C = CIN, TK_LDH = T::LDH, TK_RS = T::RS, TK_PLANE = T::PLANE, TK_BFR = T::BFR, TK_SLOTS = T::SLOTS, KS = T::KS;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __bf16* Bs = reinterpret_cast<__bf16*>(smem);                       // resident weight fragments
    __bf16* Ph = Bs + TK_BFR;                                           // window, hi then lo
    __bf16* Pl = Ph + TK_PLANE;
    float* Cs = reinterpret_cast<float*>(Ph);                           // the epilogue's staging tile aliases the window (128 x 36 floats)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c4 = tid & (T::Q - 1), lrow = lane & 31, lh = lane >> 5;
    const int nt = blockIdx.y, n0 = nt * 32;
    constexpr int INV = 0x7fffffff;
    const __amdgpu_buffer_rsrc_t rsrcX = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.x), 0, d.x_bytes, 0x00020000);

    // ---- once per workgroup: the weight fragments of this output-channel tile -> LDS (lane-linear 1-KB pieces: conflict-free reads)
    {
        const tk_uintx4* wf = reinterpret_cast<const tk_uintx4*>(d.w_frag) + (size_t)nt * (TK_BFR / 8);
#pragma unroll
        for (int i = 0; i < TK_BFR / 8 / 256; ++i) reinterpret_cast<tk_uintx4*>(Bs)[tid + 256 * i] = wf[tid + 256 * i];
    }
    // window slot j of this thread: window pixel (tid + 256 j) / Q = (py, px), channel quad c4
    int slot_yx[TK_SLOTS], slot_lds[TK_SLOTS];
#pragma unroll
    for (int j = 0; j < TK_SLOTS; ++j) {
        const int pp = (tid + 256 * j) >> T::QSH;
        const int py = pp / TK_PW, px = pp - py * TK_PW;
        slot_yx[j] = pp < TK_P ? (py << 8) | px : -1;
        slot_lds[j] = py * TK_RS + px * TK_LDH + 4 * c4;
    }
    floatx4 rs = {1.f, 1.f, 1.f, 1.f}, rt = {0.f, 0.f, 0.f, 0.f};
    if (AFF == 1) {
        rs = *reinterpret_cast<const floatx4*>(d.pro_scale + 4 * c4);
        rt = *reinterpret_cast<const floatx4*>(d.pro_shift + 4 * c4);
    }
    // fragment addresses: tile pixel o = wave * 32 + lrow = (row o >> 4, column o & 15) of the 8 x 16 tile
    const int o = wave * 32 + lrow;
    const __bf16* fa_h = Ph + (o >> 4) * TK_RS + (o & 15) * TK_LDH + 8 * lh;
    const __bf16* fa_l = fa_h + TK_PLANE;
    const __bf16* fb = Bs + lane * 8;

    const int t0 = blockIdx.x * g.per, t1 = min(g.ntiles, t0 + g.per);
    floatx4 rpat[TK_SLOTS], ps = rs, pt = rt;
    unsigned okmask = 0;
    int cur_n = 0, cur_y0 = 0, cur_x0 = 0;
    auto origin = [&](const int t, int& n, int& y0, int& x0) __attribute__((always_inline)) {
        n = fd_div(t, g.fd_tpi);
        const int rem = t - n * g.tpi, ty = fd_div(rem, g.fd_tx);
        y0 = ty * TK_TH;
        x0 = (rem - ty * g.tiles_x) * TK_TW;
    };
    auto issue = [&](const int t) __attribute__((always_inline)) {
        origin(t, cur_n, cur_y0, cur_x0);
        okmask = 0;
#pragma unroll
        for (int j = 0; j < TK_SLOTS; ++j) {
            const int y = cur_y0 - 1 + (slot_yx[j] >> 8), x = cur_x0 - 1 + (slot_yx[j] & 255);
            const bool ok = slot_yx[j] >= 0 && y >= 0 && y < g.H && x >= 0 && x < g.W;
            const int off = ok ? (((cur_n * g.H + y) * g.W + x) * d.ldx + 4 * c4) * 4 : INV;
            okmask |= (ok ? 1u : 0u) << j;
            rpat[j] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(rsrcX, off, 0, 0));
        }
        if (AFF == 2) {         // per-(image, channel) scale / shift (the SE gate's prologue of the backward convs)
            ps = *reinterpret_cast<const floatx4*>(d.pro_scale + (size_t)cur_n * TK_C + 4 * c4);
            pt = *reinterpret_cast<const floatx4*>(d.pro_shift + (size_t)cur_n * TK_C + 4 * c4);
        }
    };
    auto convert = [&]() __attribute__((always_inline)) {
        const floatx4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < TK_SLOTS; ++j) {
            floatx4 v = rpat[j];
            if (AFF == 1) {
                if (d.flags & GA_CONV_PRO_PRELU) {          // uniform: nn.PReLU, the slopes travel in pro_scale
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * rs[e];
                } else {
                    v = v * rs + rt;
                }
            }
            if (AFF == 2) v = v * ps + pt;
            if (ACT == GA_ACT_SILU) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = v[e] * fast_sigmoid(v[e]);
            } else if (ACT == GA_ACT_ELU) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : __expf(v[e]) - 1.f;
            } else if (ACT == GA_ACT_RELU) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
            } else if (ACT == GA_ACT_LRELU) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.01f * v[e];
            }
            if (AFF != 0) v = (okmask >> j) & 1u ? v : zero;            // only a shift un-zeroes the padding
            const tk_bf16x4 hi = __builtin_convertvector(v, tk_bf16x4);
            const tk_bf16x4 lo = __builtin_convertvector(v - __builtin_convertvector(hi, floatx4), tk_bf16x4);
            if (slot_yx[j] >= 0) {
                *reinterpret_cast<tk_bf16x4*>(Ph + slot_lds[j]) = hi;
                *reinterpret_cast<tk_bf16x4*>(Pl + slot_lds[j]) = lo;
            }
        }
    };

    if (t0 < t1) issue(t0);
    for (int t = t0; t < t1; ++t) {
        const int m_base = (cur_n * g.H + cur_y0) * g.W + cur_x0;      // top-left output pixel of tile t (issue(t) set the origin)
        convert();
        __syncthreads();                                    // the window (and, first time round, the weights) are in LDS
        if (t + 1 < t1) issue(t + 1);                       // next window: in flight under the MFMAs and the epilogue
        floatx16 acc[1][1];
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][0][r] = 0.f;
#pragma unroll
        for (int ch = 0; ch < KS / 2; ++ch) {               // 32-channel chunk major, tap major inside it: conv_halo3's summation order
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int tapoff = (tap / 3) * TK_RS + (tap % 3) * TK_LDH;
#pragma unroll
                for (int k2 = 0; k2 < 2; ++k2) {
                    const int ks = 2 * ch + k2;
                    const tk_bf16x8 ah = *reinterpret_cast<const tk_bf16x8*>(fa_h + tapoff + ks * 16);
                    const tk_bf16x8 al = *reinterpret_cast<const tk_bf16x8*>(fa_l + tapoff + ks * 16);
                    const tk_bf16x8 bh = *reinterpret_cast<const tk_bf16x8*>(fb + ((tap * KS + ks) * 2 + 0) * 512);
                    const tk_bf16x8 bl = *reinterpret_cast<const tk_bf16x8*>(fb + ((tap * KS + ks) * 2 + 1) * 512);
                    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[0][0], 0, 0, 0);
                    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[0][0], 0, 0, 0);
                    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[0][0], 0, 0, 0);
                }
            }
        }
        __syncthreads();                                    // every wave has read the window: its LDS becomes the epilogue's staging tile
        // (r04: the epilogue straight from the accumulators — a register of the 32 x 32 tile is two whole 128-byte pixel rows, so no
        // staging tile and two barriers less — measured 249 against 182 us forward and 794 against 245 us with act' + addend: the
        // scalar form pays per-element address arithmetic for 16 x 4-byte accesses per lane; gpurun_out/r04_thin_ab2.log.  Staged form kept.)
        conv_epilogue<4, 1, 1, 1>(d, acc, Cs, m_base, n0, M, 1, 1, 0, 4, g.W);
        __syncthreads();                                    // window (and staging tile) read: the next window may be written
    }
}

template <int CIN, int AFF, int ACT>
static void launch_thin_inst(const ga_conv_desc& d, hipStream_t stream, dim3 grid, const thin_geom& g, int M) {
    static dyn_lds_cache attr;
    const size_t lds = thin_traits<CIN>::LDS;
    (void)ensure_dyn_lds(attr, reinterpret_cast<const void*>(&conv_thin3_kernel<CIN, AFF, ACT>), lds);
    hipLaunchKernelGGL((conv_thin3_kernel<CIN, AFF, ACT>), grid, dim3(256), lds, stream, d, g, M);
}

static inline int thin_mode(const ga_conv_desc& d) {
    return ((d.pro_scale ? (d.pro_per_row ? 2 : 1) : 0) << 4) | d.pro_act;
}

// 1 when tile code 11 takes the descriptor: 3x3 / stride 1 / pad 1, one source of exactly 32 or 64 channels, images of 8 x 16 tiles
int conv_thin3_supports(const ga_conv_desc& d) {
    if (d.KH != 3 || d.KW != 3 || d.sn != 1 || d.sd != 1 || d.pad != 1 || d.C2 != 0 || (d.C1 != 32 && d.C1 != 64)) return 0;
    if (d.Ho != d.Hi || d.Wo != d.Wi || d.Ho % TK_TH || d.Wo % TK_TW || d.Wo > 255 * TK_TW) return 0;
    switch (thin_mode(d)) {
        case 0x00: case 0x01: case 0x02: case 0x03: case 0x04: case 0x10: case 0x11: case 0x20: return 1;
        default: return 0;
    }
}

template <int CIN>
static int thin_dispatch_c(const ga_conv_desc& d, hipStream_t stream, dim3 grid, const thin_geom& g, int M) {
#define GA_THIN(A, C) launch_thin_inst<CIN, A, C>(d, stream, grid, g, M)
    switch (thin_mode(d)) {
        case 0x00: GA_THIN(0, GA_ACT_NONE); break;
        case 0x01: GA_THIN(0, GA_ACT_SILU); break;
        case 0x02: GA_THIN(0, GA_ACT_ELU); break;
        case 0x03: GA_THIN(0, GA_ACT_RELU); break;
        case 0x04: GA_THIN(0, GA_ACT_LRELU); break;
        case 0x10: GA_THIN(1, GA_ACT_NONE); break;
        case 0x11: GA_THIN(1, GA_ACT_SILU); break;
        case 0x20: GA_THIN(2, GA_ACT_NONE); break;
        default: return GA_E_UNSUPPORTED;
    }
#undef GA_THIN
    return check_launch();
}

// called by ga_conv2d after validation (tile 11; vec_out and the split-bf16 operands checked there); needs d.w_frag in the thin order
// (WeightStore.frag_thin: [Cout tile of 32][tap][k step][hi | lo][lane][8], C1 / 16 k steps per tap)
int conv_thin3_dispatch(const ga_conv_desc& d, hipStream_t stream, int vec_out, int splits) {
    if (!vec_out || splits != 1 || !d.w_frag || !aligned16(d.w_frag) || !conv_thin3_supports(d)) return GA_E_UNSUPPORTED;
    const int M = d.N * d.Ho * d.Wo;
    thin_geom g;
    g.H = d.Ho; g.W = d.Wo;
    g.tiles_x = d.Wo / TK_TW;
    g.tpi = g.tiles_x * (d.Ho / TK_TH);
    g.ntiles = d.N * g.tpi;
    const int NT = (d.Cout + 31) / 32;
    int G = (d.C1 == 32 ? 512 : 256) / NT;                  // one wave of workgroups in all: every workgroup walks a long run of tiles
    if (G < 1) G = 1;
    if (G > g.ntiles) G = g.ntiles;
    g.per = (g.ntiles + G - 1) / G;
    G = (g.ntiles + g.per - 1) / g.per;
    g.fd_tpi = make_fastdiv(g.tpi);
    g.fd_tx = make_fastdiv(g.tiles_x);
    const dim3 grid(G, NT);
    return d.C1 == 32 ? thin_dispatch_c<32>(d, stream, grid, g, M) : thin_dispatch_c<64>(d, stream, grid, g, M);
}

}  // namespace ga
