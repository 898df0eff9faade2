// conv_halo3 — 3x3 / stride 1 / pad 1 convolutions on the split-bf16 matrix path (see conv_bf3.hip for the arithmetic)
// with the INPUT WINDOW STAGED ONCE PER CHANNEL CHUNK instead of once per tap.
//
// The generic implicit-GEMM kernel gathers a fresh [128 pixels][32 channels] operand tile for each of the 9 taps: every
// input value is fetched from L2, run through the prologue (affine / SiLU / ...) and split into bf16 hi + lo nine
// times per output-channel tile.  On gfx950 that loader — not the MFMAs — bounds the 3x3 layers (SiLU costs two
// quarter-rate transcendentals per element).  Here a workgroup owns 128 output pixels that form whole image rows
// (128/Wo rows of one image, or 128/(Ho*Wo) whole images), stages their halo window ("patch": (TH+2) x (Wo+2) pixels
// per image, zero padded, 1.4-2.3x the tile instead of 9x) once per 32-channel chunk — prologue and split included —
// and feeds the 9 taps from it: the A fragment of tap (kh, kw) is the same ds_read_b128 at a uniform LDS offset
// (kh*(Wo+2) + kw) rows further.  Only the weight tile changes per tap (pre-split bf16, plain copies, double-buffered).
// Same tile order, epilogue, split-K (over channel chunks) and operand layouts as conv_bf3.
// Wide images (Wo a multiple of 128: the 256^2 .. 1024^2 layers of the e4e encoder and the StyleGAN2 synthesis network): the
// tile is a 128-pixel SEGMENT of one row and its window 3 x 130 pixels (3x the tile instead of 9x); 13 patch slots per thread
// instead of 9, hence a second set of instantiations (template parameter RP).
#include <type_traits>
#include "ga_common.h"
#include "conv_epilogue.h"

namespace ga {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned uintx4 __attribute__((ext_vector_type(4)));

#ifdef GA_TRACE     // see conv_bf3.hip: per-workgroup phase stamps for tools/conv_trace.py
__device__ unsigned long long ga_trace_buf_halo[8 * 8192];
#define GA_HSTAMP(i)                                                                                     \
    if (threadIdx.x == 0 && blockIdx.x < 8192 && blockIdx.y == 0) ga_trace_buf_halo[blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memtime();
// K-loop anatomy (wave 0 of every workgroup): clocks in the MFMA part of the steps, in the chunk-boundary patch staging and in
// the step barrier, summed over the loop into trace slots 5..7
#define GA_HT0 unsigned long long hsum[3] = {0, 0, 0}; unsigned long long hprev = __builtin_amdgcn_s_memtime();
#define GA_HT(i) { __builtin_amdgcn_sched_barrier(0); const unsigned long long tn = __builtin_amdgcn_s_memtime(); hsum[i] += tn - hprev; hprev = tn; __builtin_amdgcn_sched_barrier(0); }
#define GA_HTEND if (threadIdx.x == 0 && blockIdx.x < 8192 && blockIdx.y == 0) { for (int i = 0; i < 3; ++i) ga_trace_buf_halo[blockIdx.x * 8 + 5 + i] = hsum[i]; }
#else
#define GA_HSTAMP(i)
#define GA_HT0
#define GA_HT(i)
#define GA_HTEND
#endif

constexpr int HK = 32;          // channels per chunk
constexpr int LDH = 40;         // bf16 per LDS row (32 + 8 pad = 80 B, conflict-free 16-B fragment reads)
// patch float4 slots per thread (template parameter RP): 9 = ceil(288 * 8 / 256) for tiles of whole rows (at most 288 patch
// pixels: 8 images of 4x4 -> 8 * 6 * 6), 13 for a row segment (3 * 130 = 390 patch pixels)

// LDS image of the patch: pixel (img, py, px) of a plane at img*IS + py*RS + px*LDH elements.  RS and IS are padded so
// that tile row r lands on the bank slot of a linear 80-B pitch (80 r mod 256) although the rows of the window are
// two pixels longer than the tile's: RS = Wo*80 (mod 256) bytes, IS = TH*Wo*80 (mod 256) — the 16-lane groups of the
// fragment ds_read_b128 then stay conflict-free across row and image boundaries, as in conv_bf3's linear tile.
struct halo_geom {
    int TH, NI, PH, PW, P;      // rows per image in the tile, images per tile, patch rows / cols per image, patch pixels
    int TW, wide;               // tile width (= Wo, or 128 for a row segment of a wide image: wide = 1)
    int RS, IS;                 // row / image stride of a patch plane, in bf16 elements
    int ldh;                    // pixel pitch of a patch plane, in bf16 elements (LDH; 48 for the 16x16x32 fragment reads of tile 8)
    fastdiv fd_howo, fd_wo, fd_phpw, fd_pw, fd_thwo, fd_tw, fd_is;
};

template <int WM, int WN, int TM, int TN, int AFF, int ACT, int RPMAX>
__global__ void __launch_bounds__(256, 2)
conv_halo3_kernel(const ga_conv_desc d, const int tilesN, const int M, const int C, const int Ktot, const int nkc,
                  const int vec_out, const halo_geom g) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    static_assert(BM == 128, "the patch geometry is derived for 128-pixel tiles");
    constexpr int RB = BN >= 64 ? BN / 64 : 1;
    constexpr int BSTAGE = 2 * BN * LDH;                    // bf16 elements of one weight stage (hi + lo)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __bf16* Ph = reinterpret_cast<__bf16*>(smem);           // patch, hi then lo: [P][LDH] each
    __bf16* Pl = Ph + g.NI * g.IS;
    __bf16* Bst = Pl + g.NI * g.IS;                         // two weight stages

    GA_HSTAMP(0)
    int bid;
    {
        const int nb = gridDim.x, orig = blockIdx.x;
        const int q = nb >> 3, r = nb & 7, xcd = orig & 7, k = orig >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
    }
    const int m0 = (bid / tilesN) * BM;
    const int n0 = (bid % tilesN) * BN;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int c4 = tid & 7;                                 // this thread's channel quad of the chunk (all its patch slots)
    const int k8 = tid & 3, rb0 = tid >> 2;                 // weight staging: 4 k-octets x 64 rows
    const int lrow = lane & 31, lh = lane >> 5;

    constexpr int INV = 0x7fffffff;
    const int HoWo = d.Ho * d.Wo;
    const int n_first = fd_div(m0, g.fd_howo);
    const int y0 = HoWo > BM ? fd_div(m0 - n_first * HoWo, g.fd_wo) : 0;
    const int x0 = g.wide ? m0 - n_first * HoWo - y0 * d.Wo : 0;       // first column of a row segment
    const int rp = (g.P * 8 + 255) >> 8;                    // patch slots per thread in use (uniform)

    const __amdgpu_buffer_rsrc_t rsrcX = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.x), 0, d.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrcWh = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(d.w_hi), 0, d.w_bytes / 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrcWl = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(d.w_lo), 0, d.w_bytes / 2, 0x00020000);

    // ---- patch slots: slot s = tid + 256 j -> patch pixel s >> 3, channel quad s & 7 (= c4).  Per slot: the global byte offset
    // (INV: padding / outside the batch) and one packed word — LDS element offset (bits 0..19), image within the tile (20..23),
    // "no such patch pixel" (30)
    int pbase[RPMAX];
    unsigned pmeta[RPMAX];
#pragma unroll
    for (int j = 0; j < RPMAX; ++j) {
        const int pp = (tid + 256 * j) >> 3;
        int off = INV;
        pmeta[j] = 1u << 30;
        if (j < rp && pp < g.P) {
            const int img = fd_div(pp, g.fd_phpw);
            const int rem = pp - img * g.PH * g.PW;
            const int py = fd_div(rem, g.fd_pw), px = rem - py * g.PW;
            pmeta[j] = (unsigned)(img * g.IS + py * g.RS + px * LDH + 4 * c4) | ((unsigned)img << 20);
            const int n = n_first + img, hi = y0 - 1 + py, wi = x0 + px - 1;
            if (n < d.N && hi >= 0 && hi < d.Hi && wi >= 0 && wi < d.Wi) off = ((n * d.Hi + hi) * d.Wi + wi) * d.ldx * 4 + c4 * 16;
        }
        pbase[j] = off;
    }
    // ---- A fragment rows: tile row o -> patch pixel of the window's top-left tap
    int fragA[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int o = wm * TM * 32 + i * 32 + lrow;
        const int img = fd_div(o, g.fd_thwo);
        const int rem = o - img * g.TH * g.TW;
        const int y = fd_div(rem, g.fd_tw), x = rem - y * g.TW;
        fragA[i] = img * g.IS + y * g.RS + x * LDH + 8 * lh;
    }
    int baseB[RB];
#pragma unroll
    for (int i = 0; i < RB; ++i) {
        const int row = rb0 + 64 * i;
        const int co = n0 + row;
        baseB[i] = (row < BN && co < d.Cout) ? (co * Ktot + 8 * k8) * 2 : INV;
    }

    floatx4 rpat[RPMAX], rs = {1.f, 1.f, 1.f, 1.f}, rt = {0.f, 0.f, 0.f, 0.f};
    uintx4 rbh[2][RB], rbl[2][RB];       // two weight staging sets: the loads of step s+2 fly during the whole of step s

    auto issue_patch = [&](const int chunk) __attribute__((always_inline)) {
        const int soff = chunk * HK * 4;
#pragma unroll
        for (int j = 0; j < RPMAX; ++j)
            if (j < rp) rpat[j] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(rsrcX, pbase[j], soff, 0));
        if (AFF == 1) {
            rs = *reinterpret_cast<const floatx4*>(d.pro_scale + chunk * HK + 4 * c4);
            rt = *reinterpret_cast<const floatx4*>(d.pro_shift + chunk * HK + 4 * c4);
        }
    };
    int cur_chunk = 0;                                      // chunk whose data sits in rpat (for the per-row affine)
    auto finish_patch = [&]() __attribute__((always_inline)) {
        const floatx4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < RPMAX; ++j) {
            if (j < rp) {
                floatx4 v = rpat[j];
                if (AFF == 1) {
                    if (d.flags & GA_CONV_PRO_PRELU) {      // uniform: nn.PReLU, the slopes travel in pro_scale
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * rs[e];
                    } else {
                        v = v * rs + rt;
                    }
                }
                if (AFF == 2) {     // per-(image, channel) scale / shift: small and L2-resident, fetched as the slot is converted
                    const int prow = min(n_first + (int)((pmeta[j] >> 20) & 15u), d.N - 1) * C + 4 * c4 + cur_chunk * HK;
                    const floatx4 ps = *reinterpret_cast<const floatx4*>(d.pro_scale + prow);
                    const floatx4 pt = *reinterpret_cast<const floatx4*>(d.pro_shift + prow);
                    v = v * ps + pt;
                }
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
                if (AFF != 0) v = pbase[j] != INV ? v : zero;           // only a shift un-zeroes the padding
                const bf16x4 hi = __builtin_convertvector(v, bf16x4);
                const bf16x4 lo = __builtin_convertvector(v - __builtin_convertvector(hi, floatx4), bf16x4);
                if (!(pmeta[j] & (1u << 30))) {
                    *reinterpret_cast<bf16x4*>(Ph + (pmeta[j] & 0xfffffu)) = hi;
                    *reinterpret_cast<bf16x4*>(Pl + (pmeta[j] & 0xfffffu)) = lo;
                }
            }
        }
    };

    // weight tiles advance tap-fastest inside a chunk: k offset (tap * C + chunk * 32)
    int q_tap = 0, q_chunk = 0;
    auto issue_B = [&](const int set) __attribute__((always_inline)) {
        const int soffB = (q_tap * C + q_chunk * HK) * 2;
        if (++q_tap == 9) { q_tap = 0; ++q_chunk; }
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            rbh[set][i] = __builtin_amdgcn_raw_buffer_load_b128(rsrcWh, baseB[i], soffB, 0);
            rbl[set][i] = __builtin_amdgcn_raw_buffer_load_b128(rsrcWl, baseB[i], soffB, 0);
        }
    };
    auto finish_B = [&](const int set, const int buf) __attribute__((always_inline)) {
        __bf16* Bh = Bst + buf * BSTAGE;
        __bf16* Bl = Bh + BN * LDH;
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            const int row = rb0 + 64 * i;
            if (BN >= 64 || row < BN) {
                *reinterpret_cast<uintx4*>(Bh + row * LDH + 8 * k8) = rbh[set][i];
                *reinterpret_cast<uintx4*>(Bl + row * LDH + 8 * k8) = rbl[set][i];
            }
        }
    };

    floatx16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    auto mma_tap = [&](const int buf, const int tapoff) __attribute__((always_inline)) {
        const __bf16* Bh = Bst + buf * BSTAGE + (wn * TN * 32 + lrow) * LDH + 8 * lh;
        const __bf16* Bl = Bh + BN * LDH;
#pragma unroll
        for (int ks = 0; ks < HK / 16; ++ks) {
            bf16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                ah[i] = *reinterpret_cast<const bf16x8*>(Ph + fragA[i] + tapoff + ks * 16);
                al[i] = *reinterpret_cast<const bf16x8*>(Pl + fragA[i] + tapoff + ks * 16);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                bh[j] = *reinterpret_cast<const bf16x8*>(Bh + j * 32 * LDH + ks * 16);
                bl[j] = *reinterpret_cast<const bf16x8*>(Bl + j * 32 * LDH + ks * 16);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                }
        }
    };

    // split-K over channel chunks
    const int splits = gridDim.y, split = blockIdx.y;
    const int cper = (nkc + splits - 1) / splits;
    const int cb = split * cper, ce = min(nkc, cb + cper);
    const int nsteps = (ce - cb) * 9;

    GA_HSTAMP(1)
    if (nsteps > 0) {
        q_chunk = cb;
        issue_patch(cb);
        issue_B(0);
        if (nsteps > 1) issue_B(1);
        cur_chunk = cb;
        finish_patch();
        finish_B(0, 0);
        if (cb + 1 < ce) issue_patch(cb + 1);
    }
    __syncthreads();
    GA_HSTAMP(2)
    int tap = 0, tapoff = 0, chunk = cb;
    GA_HT0
    // step s: weight stage s & 1 feeds the MFMAs; register set s & 1 (drained into that stage one step ago) takes the
    // loads of step s+2; set (s+1) & 1, loaded during step s-1, is written to the other stage behind the MFMAs
    auto step = [&](const int par, const int s) __attribute__((always_inline)) {
#ifndef GA_EXP
#define GA_EXP 0        // trace builds only: bit 0 drops the weight LDS writes (and with them the loads), 2 the barrier, 3 the MFMAs
#endif
        if (s + 2 < nsteps) issue_B(par);
        if (!(GA_EXP & 8)) mma_tap(par, tapoff);
        if (s + 1 < nsteps && !(GA_EXP & 1)) finish_B(par ^ 1, par ^ 1);
        constexpr int NM = TM * TN * 3 * (HK / 16);
        __builtin_amdgcn_sched_group_barrier(0x008, NM - 2 * RB * 2, 0);        // MFMAs first: the weight loads have a full step
#pragma unroll
        for (int k = 0; k < 2 * RB; ++k) {
            __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);                  // one LDS write ...
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                  // ... per two trailing MFMAs
        }
        GA_HT(0)
        ++tap;
        tapoff += LDH;                                      // next column of the window ...
        if (tap == 3 || tap == 6) tapoff += g.RS - 3 * LDH;      // ... or the start of its next row
        if (tap == 9) {
            tap = 0; tapoff = 0; ++chunk;
            if (chunk < ce) {
                __syncthreads();                            // every wave has read the old patch
                cur_chunk = chunk;
                finish_patch();
                if (chunk + 1 < ce) issue_patch(chunk + 1);
            }
            GA_HT(1)
        }
        if (!(GA_EXP & 4)) __syncthreads();
        GA_HT(2)
    };
    for (int s = 0; s < nsteps; s += 2) {
        step(0, s);
        if (s + 1 < nsteps) step(1, s + 1);
    }
    GA_HTEND
    __syncthreads();
    GA_HSTAMP(3)

    conv_epilogue<WM, WN, TM, TN>(d, acc, smem, m0, n0, M, vec_out, splits, split);
    GA_HSTAMP(4)
}

template <int WM, int WN, int TM, int TN, int AFF, int ACT, int RP>
static void launch_halo_inst(const ga_conv_desc& d, hipStream_t stream, dim3 grid, size_t lds, int tilesN, int M, int Ktot,
                             int nkc, int vec_out, const halo_geom& g) {
    static dyn_lds_cache attr;
    (void)ensure_dyn_lds(attr, reinterpret_cast<const void*>(&conv_halo3_kernel<WM, WN, TM, TN, AFF, ACT, RP>), lds);
    hipLaunchKernelGGL((conv_halo3_kernel<WM, WN, TM, TN, AFF, ACT, RP>), grid, dim3(256), lds, stream, d, tilesN, M, d.C1, Ktot,
                       nkc, vec_out, g);
}


// ---------------------------------------------------------------------------------------------------------------------
// conv_halo3_bd — the same convolution with the WEIGHT FRAGMENTS READ STRAIGHT FROM GLOBAL MEMORY (tile code 8).
//
// Anatomy of the kernel above at 512 rows x 16 x 16 x 128 (tools/conv_trace.py, trace build; clocks per tap step of one
// workgroup, two workgroups per CU): MFMA part 1705 (two waves' 2 x 768 on the shared matrix pipe + the weight tile's register ->
// LDS writes), patch staging at the chunk boundaries 250, step barrier 264.  A third of a step goes to handing the next weight
// tile from registers through LDS to the waves — one barrier per 24 MFMAs.  Here a wave owns all 128 pixels of the tile and a
// quarter of the output channels (waves 1 x 4 instead of 2 x 2): its B fragments are then nobody else's, so they need no LDS and no
// barrier.  They are loaded a whole step ahead with fully coalesced 1-KB wave loads from a copy of the weights laid out in
// fragment order at load time (`w_frag`: [N tile][chunk][tap][wave][k step][hi | lo][lane][8 bf16], written by the host once),
// and the patch is double-buffered where it fits: ONE barrier per 32-channel chunk (9 taps, 216 MFMAs per wave) instead of nine.
// Same operand split, k order and accumulation order as conv_halo3_kernel: bitwise the same results.
// ---------------------------------------------------------------------------------------------------------------------
template <int AFF, int ACT, int RPMAX, bool M16>
__global__ void __launch_bounds__(256, 2)
conv_halo3_bd_kernel(const ga_conv_desc d, const int tilesN, const int M, const int C, const int nkc, const int vec_out,
                     const halo_geom g, const int dbuf, const int tab_off) {
    constexpr int WM = 1, WN = 4, TM = 4, TN = 1, BM = 128, BN = 128;
    const int LDHr = g.ldh;                                  // pixel pitch of the patch planes (uniform)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int plane = g.NI * g.IS;                           // bf16 elements of one patch plane (hi or lo)
    __bf16* Pbase = reinterpret_cast<__bf16*>(smem);         // buffer b: hi at b * 2 * plane, lo behind it
    GA_HSTAMP(0)

    int bid;
    {
        const int nb = gridDim.x, orig = blockIdx.x;
        const int q = nb >> 3, r = nb & 7, xcd = orig & 7, k = orig >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
    }
    const int m0 = (bid / tilesN) * BM;
    const int nt = bid % tilesN, n0 = nt * BN;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave;
    const int c4 = tid & 7;
    // fragment lane map: 32x32x16 -> row lane & 31, k octet lane >> 5 (of a 16-deep step); 16x16x32 -> row lane & 15, k octet
    // lane >> 4 (of the 32-deep chunk)
    const int lrow = M16 ? (lane & 15) : (lane & 31), lh = M16 ? (lane >> 4) : (lane >> 5);

    constexpr int INV = 0x7fffffff;
    const int HoWo = d.Ho * d.Wo;
    const int n_first = fd_div(m0, g.fd_howo);
    const int y0 = HoWo > BM ? fd_div(m0 - n_first * HoWo, g.fd_wo) : 0;
    const int x0 = g.wide ? m0 - n_first * HoWo - y0 * d.Wo : 0;
    const int rp = (g.P * 8 + 255) >> 8;

    const __amdgpu_buffer_rsrc_t rsrcX = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.x), 0, d.x_bytes, 0x00020000);

    // patch slot j of this thread (patch pixel (tid + 256 j) >> 3, channel quad c4): its global byte offset (INV: padding or
    // outside the batch) and its LDS element offset / 4 (0xffff: no such patch pixel) are per-thread constants that are needed
    // once per 32-channel chunk; they live in LDS ([rp][256] ints, then [rp][256] shorts, behind the patch buffers) instead
    // of 18 registers — two conflict-free LDS reads per slot and chunk, and the kernel fits its 256 registers without scratch
    int* sl_g = reinterpret_cast<int*>(smem + tab_off);
    unsigned short* sl_p = reinterpret_cast<unsigned short*>(sl_g + rp * 256);
    // split-K over channel chunks (needed here already: the FIRST chunk's patch loads leave from the slot loop below, so that their
    // memory round trip runs under the rest of the setup — prologue table, fragment addresses — instead of after it; r04)
    const int splits = gridDim.y, split = blockIdx.y;
    const int cper = (nkc + splits - 1) / splits;
    const int cb = split * cper, ce = min(nkc, cb + cper);
    floatx4 rpat[RPMAX];
#pragma unroll
    for (int j = 0; j < RPMAX; ++j) {
        if (j < rp) {
            const int pp = (tid + 256 * j) >> 3;
            int off = INV, pl = 0xffff;
            if (pp < g.P) {
                const int img = fd_div(pp, g.fd_phpw);
                const int rem = pp - img * g.PH * g.PW;
                const int py = fd_div(rem, g.fd_pw), px = rem - py * g.PW;
                pl = (img * g.IS + py * g.RS + px * LDHr + 4 * c4) >> 2;
                const int n = n_first + img, hi = y0 - 1 + py, wi = x0 + px - 1;
                if (n < d.N && hi >= 0 && hi < d.Hi && wi >= 0 && wi < d.Wi) off = ((n * d.Hi + hi) * d.Wi + wi) * d.ldx * 4 + c4 * 16;
            }
            if (cb < ce) rpat[j] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(rsrcX, off, cb * HK * 4, 0));
            sl_g[j * 256 + tid] = off;
            sl_p[j * 256 + tid] = (unsigned short)pl;
        }
    }
    // AFF != 0: the prologue's scale / shift table, staged once: [scale | shift][TI][C] floats behind the slot tables, TI = 1
    // (per-channel affine, or the PReLU slopes) or the tile's NI images (per-(image, channel) affine)
    float* ptab = reinterpret_cast<float*>(sl_p + rp * 256);
    const int TI = AFF == 2 ? g.NI : 1;
    if (AFF != 0) {
        const int quads = TI * (C >> 2);
        for (int q = tid; q < quads; q += 256) {
            const int img = q / (C >> 2), cq = q - img * (C >> 2);
            const size_t row = AFF == 2 ? (size_t)min(n_first + img, d.N - 1) * C : 0;
            *reinterpret_cast<floatx4*>(ptab + img * C + 4 * cq) = *reinterpret_cast<const floatx4*>(d.pro_scale + row + 4 * cq);
            *reinterpret_cast<floatx4*>(ptab + (TI + img) * C + 4 * cq) = *reinterpret_cast<const floatx4*>(d.pro_shift + row + 4 * cq);
        }
    }
    // a thread's slot entries are read by itself only; the prologue table is read by everybody in the first finish_patch below
    if (AFF != 0) __syncthreads();
    // 16x16x32: M tiles 4..7 sit 64 tile pixels behind tiles 0..3 — whole rows or whole images further, i.e. at ONE uniform
    // offset (`halfoff`): 4 address registers serve all 8 fragments
    int fragA[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int o = i * (M16 ? 16 : 32) + lrow;
        const int img = fd_div(o, g.fd_thwo);
        const int rem = o - img * g.TH * g.TW;
        const int y = fd_div(rem, g.fd_tw), x = rem - y * g.TW;
        fragA[i] = img * g.IS + y * g.RS + x * LDHr + 8 * lh;
    }
    int halfoff = 0;
    if (M16) {
        const int img = fd_div(64, g.fd_thwo);
        const int rem = 64 - img * g.TH * g.TW;
        const int y = fd_div(rem, g.fd_tw), x = rem - y * g.TW;
        halfoff = img * g.IS + y * g.RS + x * LDHr;
    }

    auto issue_patch = [&](const int chunk) __attribute__((always_inline)) {
        const int soff = chunk * HK * 4;
#pragma unroll
        for (int j = 0; j < RPMAX; ++j)
            if (j < rp) rpat[j] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(rsrcX, sl_g[j * 256 + tid], soff, 0));
    };
    auto finish_patch = [&](const int chunk, const int buf) __attribute__((always_inline)) {
        const floatx4 zero = {0.f, 0.f, 0.f, 0.f};
        __bf16* Ph = Pbase + buf * 2 * plane;
        __bf16* Pl = Ph + plane;
        floatx4 rs = {1.f, 1.f, 1.f, 1.f}, rt = {0.f, 0.f, 0.f, 0.f};
        if (AFF == 1) {
            rs = *reinterpret_cast<const floatx4*>(ptab + chunk * HK + 4 * c4);
            rt = *reinterpret_cast<const floatx4*>(ptab + C + chunk * HK + 4 * c4);
        }
#pragma unroll
        for (int j = 0; j < RPMAX; ++j) {
            if (j < rp) {
                floatx4 v = rpat[j];
                const int pl = sl_p[j * 256 + tid];
                if (AFF == 1) {
                    if (d.flags & GA_CONV_PRO_PRELU) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * rs[e];
                    } else {
                        v = v * rs + rt;
                    }
                }
                if (AFF == 2) {
                    const int img = pl != 0xffff ? fd_div(4 * pl, g.fd_is) : 0;        // (no such patch pixel: any table row, the value is dropped)
                    const float* tp = ptab + img * C + chunk * HK + 4 * c4;
                    v = v * *reinterpret_cast<const floatx4*>(tp) + *reinterpret_cast<const floatx4*>(tp + TI * C);
                }
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
                if (AFF != 0) v = sl_g[j * 256 + tid] != INV ? v : zero;        // only a shift un-zeroes the padding
                const bf16x4 hi = __builtin_convertvector(v, bf16x4);
                const bf16x4 lo = __builtin_convertvector(v - __builtin_convertvector(hi, floatx4), bf16x4);
                if (pl != 0xffff) {
                    *reinterpret_cast<bf16x4*>(Ph + 4 * pl) = hi;
                    *reinterpret_cast<bf16x4*>(Pl + 4 * pl) = lo;
                }
            }
        }
    };

    // weight fragments of (chunk, tap): 4 x 16 bytes per lane (k step 0 / 1 x hi / lo), lane-linear 1-KB pieces
    const uintx4* wf = reinterpret_cast<const uintx4*>(d.w_frag) + ((size_t)nt * nkc * 9 * 4 + wn) * 4 * 64 + lane;
    uintx4 bcur[4], bnxt[4];
    auto load_B = [&](uintx4 (&b)[4], const int chunk, const int tap) __attribute__((always_inline)) {
        const uintx4* p = wf + (size_t)(chunk * 9 + tap) * (4 * 4 * 64);
#pragma unroll
        for (int q = 0; q < 4; ++q) b[q] = p[q * 64];
    };

    // accumulators: 4 tiles of 32 x 32 (floatx16), or 8 x 2 tiles of 16 x 16 (floatx4) — 64 registers either way
    typedef typename std::conditional<M16, floatx4[8][2], floatx16[TM][TN]>::type acc_t;
    acc_t acc;
    if constexpr (M16) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
    } else {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][0][r] = 0.f;
    }

    // A fragments of a group = 8 ds_read_b128 feeding 12 MFMAs of 32 cycles (32x32x16: group = (tap, k step), 4 M tiles x hi | lo)
    // or 24 MFMAs of 16 cycles (16x16x32: group = (tap, half of the 8 M tiles), the whole 32-deep chunk per read); two register
    // sets: the reads of group g + 1 are issued BEFORE the MFMAs of group g (the waits the compiler inserts are then counted ones:
    // lgkmcnt(8) instead of a drain), so that a wave's instruction stream is MFMAs back to back with its operand reads riding in
    // the gaps
    bf16x8 ahs[M16 ? 1 : 2][4], als[M16 ? 1 : 2][4];
    auto load_A = [&](const int set, const int buf, const int off, const int half) __attribute__((always_inline)) {
        const __bf16* Ph = Pbase + buf * 2 * plane;
        const __bf16* Pl = Ph + plane;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ahs[set][i] = *reinterpret_cast<const bf16x8*>(Ph + fragA[i] + off + (M16 ? half * halfoff : 0));
            als[set][i] = *reinterpret_cast<const bf16x8*>(Pl + fragA[i] + off + (M16 ? half * halfoff : 0));
        }
    };
    // weight fragments of one tap in b[4]: 32x32x16 -> [k step][hi | lo]; 16x16x32 -> [16-channel half of the wave's 32][hi | lo]
    auto mma_group = [&](const int set, const uintx4 (&b)[4], const int sel) __attribute__((always_inline)) {
        if constexpr (M16) {
            const bf16x8 bh0 = __builtin_bit_cast(bf16x8, b[0]), bl0 = __builtin_bit_cast(bf16x8, b[1]);
            const bf16x8 bh1 = __builtin_bit_cast(bf16x8, b[2]), bl1 = __builtin_bit_cast(bf16x8, b[3]);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                floatx4& c0 = acc[4 * sel + i][0];
                floatx4& c1 = acc[4 * sel + i][1];
                c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(als[set][i], bh0, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(als[set][i], bh1, c1, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ahs[set][i], bl0, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ahs[set][i], bl1, c1, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ahs[set][i], bh0, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ahs[set][i], bh1, c1, 0, 0, 0);
            }
        } else {
            const bf16x8 bh = __builtin_bit_cast(bf16x8, b[2 * sel]), bl = __builtin_bit_cast(bf16x8, b[2 * sel + 1]);
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(als[set][i], bh, acc[i][0], 0, 0, 0);
                acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahs[set][i], bl, acc[i][0], 0, 0, 0);
                acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahs[set][i], bh, acc[i][0], 0, 0, 0);
            }
        }
    };

    GA_HSTAMP(1)
    if (cb < ce) {
        load_B(bcur, cb, 0);                                 // (the first chunk's patch loads are already in flight: slot loop above)
        finish_patch(cb, 0);
        if (cb + 1 < ce) issue_patch(cb + 1);
    }
    __syncthreads();
    GA_HSTAMP(2)
    int buf = 0;
    const int rowskip = g.RS - 3 * LDHr;
    for (int chunk = cb; chunk < ce; ++chunk) {
        load_A(0, buf, 0, 0);
        if constexpr (M16) {
            // 16x16x32: ONE fragment register set, refilled tile by tile: the 6 MFMAs of M tile i (3 per 16-channel half of the
            // wave's outputs) are followed by the two reads of tile i of the NEXT group into the registers they have just consumed —
            // 18 MFMAs (288 cycles) before their first use
            const __bf16* Ph = Pbase + buf * 2 * plane;
            const __bf16* Pl = Ph + plane;
#pragma unroll
            for (int gi = 0; gi < 18; ++gi) {
                const int tap = gi >> 1, half = gi & 1;
                if (half == 0) {
                    const int ntap = tap == 8 ? 0 : tap + 1;
                    const int nchunk = tap == 8 ? min(chunk + 1, ce - 1) : chunk;
                    load_B(bnxt, nchunk, ntap);
                }
                const int nt2 = (gi + 1) >> 1;
                const int noff = nt2 * LDHr + (nt2 / 3) * rowskip + ((gi + 1) & 1) * halfoff;
                const bf16x8 bh0 = __builtin_bit_cast(bf16x8, bcur[0]), bl0 = __builtin_bit_cast(bf16x8, bcur[1]);
                const bf16x8 bh1 = __builtin_bit_cast(bf16x8, bcur[2]), bl1 = __builtin_bit_cast(bf16x8, bcur[3]);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    floatx4& c0 = acc[4 * half + i][0];
                    floatx4& c1 = acc[4 * half + i][1];
                    c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(als[0][i], bh0, c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(als[0][i], bh1, c1, 0, 0, 0);
                    c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ahs[0][i], bl0, c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ahs[0][i], bl1, c1, 0, 0, 0);
                    c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ahs[0][i], bh0, c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ahs[0][i], bh1, c1, 0, 0, 0);
                    if (gi + 1 < 18) {
                        ahs[0][i] = *reinterpret_cast<const bf16x8*>(Ph + fragA[i] + noff);
                        als[0][i] = *reinterpret_cast<const bf16x8*>(Pl + fragA[i] + noff);
                    }
                }
                __builtin_amdgcn_sched_group_barrier(0x020, 4, 0);       // (the global loads, when this group has them)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (dbuf && gi == 9 && chunk + 1 < ce) {
                    finish_patch(chunk + 1, buf ^ 1);
                    if (chunk + 2 < ce) issue_patch(chunk + 2);
                }
                if (half == 1) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) bcur[q] = bnxt[q];
                }
            }
        } else
#pragma unroll
        for (int gi = 0; gi < 18; ++gi) {
            const int tap = gi >> 1, ks = gi & 1;
            if (ks == 0) {
                // the next tap's weight fragments fly during this tap's 24 MFMAs (the last tap of the last chunk re-reads its own)
                const int ntap = tap == 8 ? 0 : tap + 1;
                const int nchunk = tap == 8 ? min(chunk + 1, ce - 1) : chunk;
                load_B(bnxt, nchunk, ntap);
            }
            if (gi + 1 < 18) {
                const int nt2 = (gi + 1) >> 1;
                const int noff = nt2 * LDHr + (nt2 / 3) * rowskip + (M16 ? 0 : ((gi + 1) & 1) * 16);
                load_A((gi + 1) & 1, buf, noff, (gi + 1) & 1);
            }
            __builtin_amdgcn_sched_group_barrier(0x020, 4, 0);       // (the global loads, when this group has them)
            __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);       // next group's 8 fragment reads first ...
            mma_group(gi & 1, bcur, ks);
            __builtin_amdgcn_sched_group_barrier(0x008, M16 ? 24 : 12, 0);      // ... then this group's MFMAs
            __builtin_amdgcn_sched_barrier(0);
            if (dbuf && gi == 9 && chunk + 1 < ce) {        // the next chunk's patch into the other buffer, behind the MFMAs
                finish_patch(chunk + 1, buf ^ 1);
                if (chunk + 2 < ce) issue_patch(chunk + 2);
            }
            if (ks == 1) {
#pragma unroll
                for (int q = 0; q < 4; ++q) bcur[q] = bnxt[q];
            }
        }
        if (chunk + 1 < ce) {
            if (dbuf) {
                __syncthreads();                            // the other buffer is complete, nobody reads this one any more
                buf ^= 1;
            } else {
                __syncthreads();                            // every wave has read the patch
                finish_patch(chunk + 1, 0);
                if (chunk + 2 < ce) issue_patch(chunk + 2);
                __syncthreads();
            }
        }
    }
    __syncthreads();
    GA_HSTAMP(3)
    conv_epilogue<WM, WN, TM, TN>(d, acc, smem, m0, n0, M, vec_out, splits, split);
    GA_HSTAMP(4)
}

template <int AFF, int ACT, int RP, bool M16>
static void launch_halo_bd_inst(const ga_conv_desc& d, hipStream_t stream, dim3 grid, size_t lds, int tilesN, int M, int nkc, int vec_out,
                                const halo_geom& g, int dbuf, int tab_off) {
    static dyn_lds_cache attr;
    (void)ensure_dyn_lds(attr, reinterpret_cast<const void*>(&conv_halo3_bd_kernel<AFF, ACT, RP, M16>), lds);
    hipLaunchKernelGGL((conv_halo3_bd_kernel<AFF, ACT, RP, M16>), grid, dim3(256), lds, stream, d, tilesN, M, d.C1, nkc, vec_out, g, dbuf,
                       tab_off);
}

static inline int halo_mode(const ga_conv_desc& d) {
    return ((d.pro_scale ? (d.pro_per_row ? 2 : 1) : 0) << 4) | d.pro_act;
}

// 1 when the descriptor is a 3x3 / stride 1 / pad 1 single-source convolution whose 128-pixel tiles are whole image rows
// (or whole small images) or 128-pixel segments of one row
int conv_halo3_supports(const ga_conv_desc& d) {
    if (d.KH != 3 || d.KW != 3 || d.sn != 1 || d.sd != 1 || d.pad != 1 || d.C2 != 0) return 0;
    if (d.Ho != d.Hi || d.Wo != d.Wi || d.C1 % HK != 0) return 0;
    const int HoWo = d.Ho * d.Wo;
    const bool rows = d.Wo < 128 && 128 % d.Wo == 0 && (HoWo % 128 == 0 || 128 % HoWo == 0);   // tiles of whole rows / images
    const bool segs = d.Wo % 128 == 0;                                                          // tiles = row segments
    if (!rows && !segs) return 0;
    switch (halo_mode(d)) {
        case 0x00: case 0x01: case 0x02: case 0x03: case 0x04: case 0x10: case 0x11: case 0x20: return 1;
        default: return 0;
    }
}

static int halo_geometry(const ga_conv_desc& d, const int BM, halo_geom& g, const int ldh = LDH) {
    g.ldh = ldh;
    const int HoWo = d.Ho * d.Wo;
    g.wide = d.Wo >= BM ? 1 : 0;
    g.TW = g.wide ? BM : d.Wo;
    g.TH = g.wide ? 1 : (HoWo >= BM ? BM / d.Wo : d.Ho);
    g.NI = HoWo >= BM ? 1 : BM / HoWo;
    g.PH = g.TH + 2;
    g.PW = g.TW + 2;
    g.P = g.NI * g.PH * g.PW;
    if (g.P > 13 * 32) return GA_E_UNSUPPORTED;
    auto pad_to = [](int bytes, int want_mod256) { return bytes + ((want_mod256 - bytes) % 256 + 256) % 256; };
    const int rs_bytes = pad_to(g.PW * ldh * 2, (g.TW * ldh * 2) % 256);
    const int is_bytes = pad_to(g.PH * rs_bytes, (g.TH * g.TW * ldh * 2) % 256);
    g.RS = rs_bytes / 2;
    g.IS = is_bytes / 2;
    g.fd_howo = make_fastdiv(HoWo);
    g.fd_wo = make_fastdiv(d.Wo);
    g.fd_phpw = make_fastdiv(g.PH * g.PW);
    g.fd_pw = make_fastdiv(g.PW);
    g.fd_thwo = make_fastdiv(g.TH * g.TW);
    g.fd_tw = make_fastdiv(g.TW);
    g.fd_is = make_fastdiv(g.IS);
    return GA_OK;
}

template <int WM, int WN, int TM, int TN>
static int launch_halo(const ga_conv_desc& d, hipStream_t stream, int vec_out, int splits) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    const int M = d.N * d.Ho * d.Wo;
    const int Ktot = 9 * d.C1;
    const int nkc = d.C1 / HK;
    halo_geom g;
    { const int rc = halo_geometry(d, BM, g); if (rc != GA_OK) return rc; }
    if (splits > nkc) return GA_E_UNSUPPORTED;
    const int tilesM = (M + BM - 1) / BM, tilesN = (d.Cout + BN - 1) / BN;
    size_t lds = ((size_t)2 * g.NI * g.IS + (size_t)2 * 2 * BN * LDH) * 2;
    if (lds > 160 * 1024) return GA_E_UNSUPPORTED;
    const size_t lds_c = (size_t)BM * (BN + 4) * sizeof(float);
    if (lds_c > lds) lds = lds_c;
    const dim3 grid(tilesM * tilesN, splits);
    // row-segment tiles (13 patch slots) are instantiated for the prologues the wide layers use: none (backward convs), the
    // per-channel affine / PReLU (e4e IR units at 256^2 and 128^2) and the per-row style scale (StyleGAN2 modulated convs)
#define GA_HALO(A, C) launch_halo_inst<WM, WN, TM, TN, A, C, 9>(d, stream, grid, lds, tilesN, M, Ktot, nkc, vec_out, g)
#define GA_HALO_W(A, C)                                                                                               \
    if (g.P > 9 * 32) launch_halo_inst<WM, WN, TM, TN, A, C, 13>(d, stream, grid, lds, tilesN, M, Ktot, nkc, vec_out, g);  \
    else GA_HALO(A, C)
    if (g.P > 9 * 32 && !(halo_mode(d) == 0x00 || halo_mode(d) == 0x10 || halo_mode(d) == 0x20)) return GA_E_UNSUPPORTED;
    switch (halo_mode(d)) {
        case 0x00: GA_HALO_W(0, GA_ACT_NONE); break;
        case 0x01: GA_HALO(0, GA_ACT_SILU); break;
        case 0x02: GA_HALO(0, GA_ACT_ELU); break;
        case 0x03: GA_HALO(0, GA_ACT_RELU); break;
        case 0x04: GA_HALO(0, GA_ACT_LRELU); break;
        case 0x10: GA_HALO_W(1, GA_ACT_NONE); break;
        case 0x11: GA_HALO(0 + 1, GA_ACT_SILU); break;
        case 0x20: GA_HALO_W(2, GA_ACT_NONE); break;
        default: return GA_E_UNSUPPORTED;
    }
#undef GA_HALO_W
#undef GA_HALO
    return check_launch();
}

template <bool M16>
static int launch_halo_bd(const ga_conv_desc& d, hipStream_t stream, int vec_out, int splits, const int ldh) {
    constexpr int BM = 128, BN = 128;
    if (!d.w_frag || !aligned16(d.w_frag)) return GA_E_UNSUPPORTED;
    const int M = d.N * d.Ho * d.Wo;
    const int nkc = d.C1 / HK;
    halo_geom g;
    { const int rc = halo_geometry(d, BM, g, ldh); if (rc != GA_OK) return rc; }
    if (splits > nkc) return GA_E_UNSUPPORTED;
    const int tilesM = (M + BM - 1) / BM, tilesN = (d.Cout + BN - 1) / BN;
    const size_t patch = (size_t)2 * g.NI * g.IS * 2;                   // hi + lo planes of one buffer, bytes
    const int mode = halo_mode(d);
    const int rp = (g.P * 8 + 255) >> 8;                                // patch slots per thread
    const size_t tab = (size_t)rp * 256 * 6                             // slot tables: global offsets (int) + LDS offsets (short)
                       + (mode >= 0x10 ? (size_t)2 * (mode == 0x20 ? g.NI : 1) * d.C1 * sizeof(float) : 0);      // prologue table
    const int dbuf = 2 * patch + tab <= 80 * 1024 ? 1 : 0;              // two buffers when two workgroups per CU still fit
    const int tab_off = (int)(((dbuf ? 2 : 1) * patch + 15) / 16 * 4);  // floats, 16-byte aligned
    size_t lds = (size_t)tab_off * 4 + tab;
    if (lds > 80 * 1024) return GA_E_UNSUPPORTED;                       // (two workgroups per CU are the kernel's launch bounds)
    const size_t lds_c = (size_t)BM * (BN + 4) * sizeof(float);
    if (lds_c > lds) lds = lds_c;
    // (r04 experiment, dropped: asking for > 80 KB of LDS so that a launch takes ONE workgroup slot per CU and the second slot goes to
    // the other stream's kernel — out of phase, so that one's setup / epilogue would overlap the other's K loop: 5,579 against 5,949 -
    // 5,967 rows/s on the same box, gpurun_out/r04_ab_epi.log: a lone conv launch then runs at half occupancy)
    const dim3 grid(tilesM * tilesN, splits);
#define GA_HBD(A, C) launch_halo_bd_inst<A, C, 9, M16>(d, stream, grid, lds, tilesN, M, nkc, vec_out, g, dbuf, tab_off)
    if (g.P > 9 * 32) return GA_E_UNSUPPORTED;                          // row-segment tiles of wide images: the LDS-staged kernel
#ifdef GA_HALO_EXP_ONLY     // quick experiment builds (make hexp): two prologues only
    switch (halo_mode(d)) {
        case 0x00: GA_HBD(0, GA_ACT_NONE); break;
        case 0x01: GA_HBD(0, GA_ACT_SILU); break;
        default: return GA_E_UNSUPPORTED;
    }
#else
    switch (halo_mode(d)) {
        case 0x00: GA_HBD(0, GA_ACT_NONE); break;
        case 0x01: GA_HBD(0, GA_ACT_SILU); break;
        case 0x02: GA_HBD(0, GA_ACT_ELU); break;
        case 0x03: GA_HBD(0, GA_ACT_RELU); break;
        case 0x04: GA_HBD(0, GA_ACT_LRELU); break;
        case 0x10: GA_HBD(1, GA_ACT_NONE); break;
        case 0x11: GA_HBD(1, GA_ACT_SILU); break;
        case 0x20: GA_HBD(2, GA_ACT_NONE); break;
        default: return GA_E_UNSUPPORTED;
    }
#endif
#undef GA_HBD
    return check_launch();
}

// tile codes 5 (128 x 128), 6 (128 x 64) and 7 (128 x 32) of ga_conv_desc.tile, 8 = 128 x 128 with the weight fragments read from
// global memory (needs w_frag); called by ga_conv2d after validation
int conv_halo3_dispatch(const ga_conv_desc& d, hipStream_t stream, int tile, int vec_out, int splits) {
    switch (tile) {
        case 8: return launch_halo_bd<false>(d, stream, vec_out, splits, LDH);
#ifdef GA_HALO_EXP_ONLY
        // experiment build only (make hexp, tools/conv_ab.py): the same kernel on v_mfma_f32_16x16x32_bf16 (M16), w_frag in the m16
        // order (WeightStore.frag3(w, m16=True)); 9 = 80-B pixel pitch (2-way fragment reads), 10 = 96-B pitch (conflict-free).
        // Measured r04 (gpurun_out/r04_ab_m16.log): the bare MFMA loop runs 15 % faster on this shape, the kernel
        // 2 - 7 % (16x16x128: 111.4 -> 105.9 us, 8x8x256: 102.2 -> 100.3, 32x32x64: 213.8 -> 199.8) and 4 % SLOWER at 4x4x512
        // (115.3 -> 120.3): not worth a second set of instantiations (+5 minutes of build) and a second weight copy — not shipped.
        case 9: return launch_halo_bd<true>(d, stream, vec_out, splits, 40);
        case 10: return launch_halo_bd<true>(d, stream, vec_out, splits, 48);
#else
        case 5: return launch_halo<2, 2, 2, 2>(d, stream, vec_out, splits);
        case 6: return launch_halo<4, 1, 1, 2>(d, stream, vec_out, splits);
        case 7: return launch_halo<4, 1, 1, 1>(d, stream, vec_out, splits);
#endif
        default: return GA_E_UNSUPPORTED;
    }
}

}  // namespace ga

#ifdef GA_TRACE
extern "C" int ga_debug_trace_read_halo(unsigned long long* out, int n) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(ga::ga_trace_buf_halo), (size_t)n * 8) == hipSuccess ? GA_OK : GA_E_LAUNCH;
}
#endif
