// conv_bf3 — the implicit-GEMM convolution of conv_mfma.hip on the bf16 matrix cores with fp32-class accuracy.
//
// On gfx950 the f32-input MFMA runs at the vector-ALU rate (64 FLOP/clk/SIMD); the bf16 MFMA runs 16x faster.  Every
// fp32 operand x is split as x = hi + lo + r with hi = bf16(x), lo = bf16(x - hi), |r| <= 2^-16 |x|, and
//     a*b  ~=  a_lo*b_hi + a_hi*b_lo + a_hi*b_hi        (three v_mfma_f32_32x32x16_bf16, fp32 accumulation)
// which keeps ~16 mantissa bits per product (relative error ~2e-5 per product, averaging down over K) — two orders
// of magnitude inside the path's 1e-3 parity bar — at 16/3 of the f32-MFMA rate.
//   * activations: split in registers by the loader, after the prologue (affine / SiLU / ELU / ReLU), right before
//     the LDS write: v_cvt_pk_bf16_f32 x2, shift back, subtract, v_cvt_pk again (12 VALU per float4; the bf16 matrix
//     pipe runs beside the VALU, unlike the f32 MFMA);
//   * weights: split once at load time on the host into two bf16 [Cout][K] arrays (w_hi, w_lo);
//   * LDS: four bf16 tiles per stage (A_hi, A_lo, B_hi, B_lo), rows of 32 k + 8 pad = 80 B: the 16-B fragment reads
//     of a 16-lane group fall on 16 distinct slots of the 256-B bank row (80*r mod 256 is a bijection on r < 16);
//   * fragments: lane (r = lane&31, h = lane>>5) reads the 8 bf16 A[r][8h..8h+7] / B[8h..8h+7][r] of a 16-deep k step
//     as ONE ds_read_b128 — exactly the operand layout of v_mfma_f32_32x32x16_bf16;
//   * everything else (gather by buffer loads with per-row offsets and tap masks, scalar K cursor, register
//     prefetch across the MFMAs, double-buffered LDS, LDS-transposed vector epilogue, split-K) is shared with the
//     fp32 kernel.  Shapes this kernel does not take (stride-2 transposes, channel counts not multiple of 8, the
//     3-channel image) run on the exact fp32-MFMA kernel.
#include "ga_common.h"
#include "conv_epilogue.h"

namespace ga {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned uintx4 __attribute__((ext_vector_type(4)));

// -DGA_TRACE (make trace -> libga_ops_trace.so, tools/conv_trace.py): shader-clock stamps of the phases of each workgroup
#ifdef GA_TRACE
__device__ unsigned long long ga_trace_buf[8 * 8192];
#define GA_STAMP(i)                                                                                      \
    if (threadIdx.x == 0 && blockIdx.x < 8192 && blockIdx.y == 0) {                                      \
        ga_trace_buf[blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memtime();                              \
        if ((i) == 0) ga_trace_buf[blockIdx.x * 8 + 7] = __builtin_amdgcn_s_getreg((31 << 11) | 4);     \
    }
#else
#define GA_STAMP(i)
#endif

constexpr int BK3 = 32;     // k per LDS stage
constexpr int LDB = 40;     // bf16 elements per LDS row (32 + 8 pad = 80 B)

// AFF: 0 no affine, 1 per-channel scale/shift, 2 per-(row,channel) scale/shift.  ACT: the GA_ACT_* prologue activation.
// DUAL: a second K source (x2) exists (only without a prologue).  All three are compile-time so that the steady-state
// loop body is ONE basic block with no calls: the scheduler can then place the loader's VALU work (prologue + bf16
// split of tile t+1) BETWEEN the MFMAs of tile t (sched_group_barrier).  The lambdas are always_inline for the same
// reason — left to its heuristics the compiler outlined finish_tile in the larger instances, which put the staging
// registers in scratch and turned the LDS accesses into flat ones.
#ifndef GA_MINWG
#define GA_MINWG 2      // two workgroups per CU: the register budget is 256 per lane
#endif
template <int WM, int WN, int TM, int TN, int AFF, int ACT, bool DUAL>
__global__ void __launch_bounds__(256, GA_MINWG)
conv_bf3_kernel(const ga_conv_desc d, const int tilesN, const int M, const int Ctot, const int Ktot, const int nkc,
                const int vec_out, const fastdiv fd_howo, const fastdiv fd_wo) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    constexpr int RA = BM / 32;
    constexpr int RB = BN >= 64 ? BN / 64 : 1;
    constexpr int STAGE = (2 * BM + 2 * BN) * LDB;          // bf16 elements per stage
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __bf16* lds = reinterpret_cast<__bf16*>(smem);

    GA_STAMP(0)
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
    const int c4 = tid & 7, r0 = tid >> 3;                  // A staging: 8 channel-quads x 32 rows
    const int k8 = tid & 3, rb0 = tid >> 2;                 // B staging: 4 k-octets x 64 rows

    const int HoWo = d.Ho * d.Wo;
    constexpr int INV = 0x7fffffff;
    int a_n[RA], baseA[RA], baseA2[RA], baseB[RB];
    unsigned maskA[RA];
    const __amdgpu_buffer_rsrc_t rsrcX = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.x), 0, d.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrcX2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.C2 > 0 ? d.x2 : d.x), 0,
                                                                             d.C2 > 0 ? d.x2_bytes : d.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrcWh = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(d.w_hi), 0, d.w_bytes / 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrcWl = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(d.w_lo), 0, d.w_bytes / 2, 0x00020000);
#pragma unroll
    for (int i = 0; i < RA; ++i) {
        const int m = m0 + r0 + 32 * i;
        unsigned mk = 0;
        int n = -1, h0 = 0, w0 = 0;
        if (m < M) {
            n = fd_div(m, fd_howo);
            const int rem = m - n * HoWo, ho = fd_div(rem, fd_wo), wo = rem - ho * d.Wo;
            h0 = ho * d.sn - d.pad; w0 = wo * d.sn - d.pad;
            mk = tap_mask(h0, w0, d.Hi, d.Wi, d.KH, d.KW);
        }
        a_n[i] = n;
        maskA[i] = mk;
        const int pixb = (n * d.Hi + h0) * d.Wi + w0;
        baseA[i] = pixb * d.ldx * 4 + c4 * 16;
        baseA2[i] = pixb * d.ldx2 * 4 + c4 * 16;
    }
#pragma unroll
    for (int i = 0; i < RB; ++i) {
        const int row = rb0 + 64 * i;
        const int co = n0 + row;
        baseB[i] = (row < BN && co < d.Cout) ? (co * Ktot + 8 * k8) * 2 : INV;
    }

    static_assert(!DUAL || (AFF == 0 && ACT == GA_ACT_NONE), "a dual-source conv has no prologue");
    // staging registers: NSET sets.  With two, the loads of tile t+2 are issued at the START of iteration t (into the set
    // iteration t-1 drained) and consumed in iteration t+1: a whole iteration of MFMAs hides the L2 latency.  The
    // per-row-affine variant keeps one set (its scale/shift staging would not fit 256 registers twice).
    constexpr int NSET = AFF == 2 ? 1 : 2;
    floatx4 ra[NSET][RA], rs[NSET][AFF == 2 ? RA : 1], rt[NSET][AFF == 2 ? RA : 1];
    uintx4 rbh[NSET][RB], rbl[NSET][RB];
    unsigned okmask[NSET] = {};

    int q_tap = 0, q_chunk = 0, q_kh = 0, q_kw = 0;
    auto seek_tile = [&](const int t) __attribute__((always_inline)) {
        q_tap = __builtin_amdgcn_readfirstlane(t / nkc);
        q_chunk = __builtin_amdgcn_readfirstlane(t - q_tap * nkc);
        q_kh = __builtin_amdgcn_readfirstlane(q_tap / d.KW);
        q_kw = q_tap - q_kh * d.KW;
    };

    auto issue_tile = [&](const int set) __attribute__((always_inline)) {
        const int tap = q_tap, c0 = q_chunk * BK3;
        const int kh = q_kh, kw = q_kw;
        if (++q_chunk == nkc) { q_chunk = 0; ++q_tap; if (++q_kw == d.KW) { q_kw = 0; ++q_kh; } }
        const int c = c0 + 4 * c4;
        okmask[set] = 0;
        const bool in_x = DUAL ? c0 < d.C1 : true;
        const int lim = (in_x ? d.C1 : Ctot) - c0;
        const bool cval = 4 * c4 < lim;
        const int delta = (kh * d.Wi + kw) * (in_x ? d.ldx : d.ldx2) * 4;
        const int soffA = (in_x ? c0 : c0 - d.C1) * 4;
        const unsigned bit = 1u << tap;
        if (!DUAL || in_x) {
#pragma unroll
            for (int i = 0; i < RA; ++i) {
                const bool valid = cval & ((maskA[i] & bit) != 0);
                const int off = valid ? baseA[i] + delta : INV;
                okmask[set] |= (valid ? 1u : 0u) << i;
                ra[set][i] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(rsrcX, off, soffA, 0));
            }
        } else {
#pragma unroll
            for (int i = 0; i < RA; ++i) {
                const bool valid = cval & ((maskA[i] & bit) != 0);
                const int off = valid ? baseA2[i] + delta : INV;
                okmask[set] |= (valid ? 1u : 0u) << i;
                ra[set][i] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(rsrcX2, off, soffA, 0));
            }
        }
        const int soffB = (tap * Ctot + c0) * 2;
        const bool bval = c0 + 8 * k8 < Ctot;
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            const int off = bval ? baseB[i] : INV;
            rbh[set][i] = __builtin_amdgcn_raw_buffer_load_b128(rsrcWh, off, soffB, 0);
            rbl[set][i] = __builtin_amdgcn_raw_buffer_load_b128(rsrcWl, off, soffB, 0);
        }
        if (AFF == 1) {
            const int pc = cval ? c : 0;
            rs[set][0] = *reinterpret_cast<const floatx4*>(d.pro_scale + pc);
            rt[set][0] = *reinterpret_cast<const floatx4*>(d.pro_shift + pc);
        } else if (AFF == 2) {
#pragma unroll
            for (int i = 0; i < RA; ++i) {
                const size_t po = ((okmask[set] >> i) & 1u) ? (size_t)a_n[i] * d.C1 + c : 0;
                rs[set][i] = *reinterpret_cast<const floatx4*>(d.pro_scale + po);
                rt[set][i] = *reinterpret_cast<const floatx4*>(d.pro_shift + po);
            }
        }
    };

    auto finish_tile = [&](const int set, const int buf) __attribute__((always_inline)) {
        __bf16* Ah = lds + buf * STAGE;
        __bf16* Al = Ah + BM * LDB;
        __bf16* Bh = Al + BM * LDB;
        __bf16* Bl = Bh + BN * LDB;
        const floatx4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < RA; ++i) {
            floatx4 v = ra[set][i];
            if (AFF == 2) v = v * rs[set][i] + rt[set][i];
            else if (AFF == 1) {
                const floatx4 aff = v * rs[set][0] + rt[set][0];
                if (d.flags & GA_CONV_PRO_PRELU) {          // uniform: nn.PReLU, the slopes travel in pro_scale
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * rs[set][0][e];
                } else {
                    v = aff;
                }
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
            if (AFF != 0) v = (okmask[set] >> i) & 1u ? v : zero;     // act(0) = 0 for all three: only a shift un-zeroes padding
            const bf16x4 hi = __builtin_convertvector(v, bf16x4);
            const bf16x4 lo = __builtin_convertvector(v - __builtin_convertvector(hi, floatx4), bf16x4);
            const int o = (r0 + 32 * i) * LDB + 4 * c4;
            *reinterpret_cast<bf16x4*>(Ah + o) = hi;
            *reinterpret_cast<bf16x4*>(Al + o) = lo;
        }
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            const int row = rb0 + 64 * i;
            if (BN >= 64 || row < BN) {
                *reinterpret_cast<uintx4*>(Bh + row * LDB + 8 * k8) = rbh[set][i];
                *reinterpret_cast<uintx4*>(Bl + row * LDB + 8 * k8) = rbl[set][i];
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

    const int T = d.KH * d.KW * nkc;
    const int splits = gridDim.y, split = blockIdx.y;
    const int tper = (T + splits - 1) / splits;
    const int t_begin = split * tper;
    const int t_end = min(T, t_begin + tper);

    const int lrow = lane & 31, lh = lane >> 5;
    GA_STAMP(1)
    if (t_begin < t_end) {
        seek_tile(t_begin);
        issue_tile(0);
        if (NSET == 2 && t_begin + 1 < t_end) issue_tile(1);
        finish_tile(0, 0);
    }
    __syncthreads();
    GA_STAMP(2)
    auto mma_tile = [&](const int buf) __attribute__((always_inline)) {
        const __bf16* Ah = lds + buf * STAGE + (wm * TM * 32 + lrow) * LDB + 8 * lh;
        const __bf16* Al = Ah + BM * LDB;
        const __bf16* Bh = lds + buf * STAGE + 2 * BM * LDB + (wn * TN * 32 + lrow) * LDB + 8 * lh;
        const __bf16* Bl = Bh + BN * LDB;
#pragma unroll
        for (int ks = 0; ks < BK3 / 16; ++ks) {
            bf16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                ah[i] = *reinterpret_cast<const bf16x8*>(Ah + i * 32 * LDB + ks * 16);
                al[i] = *reinterpret_cast<const bf16x8*>(Al + i * 32 * LDB + ks * 16);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                bh[j] = *reinterpret_cast<const bf16x8*>(Bh + j * 32 * LDB + ks * 16);
                bl[j] = *reinterpret_cast<const bf16x8*>(Bl + j * 32 * LDB + ks * 16);
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

    // steady state: a next tile exists -> no branch in the body; the staging math of tile t+1 is spread between the
    // MFMAs of tile t (1 MFMA : 4 VALU : 1 DS, the matrix pipe takes 32 cycles per MFMA, a VALU op 4-8)
    constexpr int NMFMA = TM * TN * 3 * (BK3 / 16);
    // VALU ops of one staged tile: per float4 ~12 for the split, 4 affine, 20 SiLU/ELU, 4 ReLU, 4 padding select
    constexpr int VOPS = RA * (12 + (AFF ? 8 : 0) + (ACT == GA_ACT_SILU || ACT == GA_ACT_ELU ? 20 : ACT == GA_ACT_RELU ? 4 : ACT == GA_ACT_LRELU ? 8 : 0)) + 8;
    constexpr int VPM = (VOPS + NMFMA - 1) / NMFMA;
#ifndef GA_EXP
#define GA_EXP 0        // trace builds only: bit 0 drops the split + LDS write, 1 the global loads, 2 the barrier, 3 the MFMAs
#endif
    // one K step: [issue tile t+2 into the drained set] ; MFMAs of LDS stage `buf` || split + LDS write of tile t+1
    // (registers of set `cons`) into the other stage ; barrier.  `cons`/`prod` are literals at every call site.
    auto step = [&](const int buf, const int cons, const int prod, const bool more2) __attribute__((always_inline)) {
        if (NSET == 2 && more2 && !(GA_EXP & 2)) issue_tile(prod);
        if (!(GA_EXP & 8)) mma_tile(buf);
        if (!(GA_EXP & 1)) finish_tile(cons, buf ^ 1);
#pragma unroll
        for (int k = 0; k < NMFMA; ++k) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // MFMA
            __builtin_amdgcn_sched_group_barrier(0x002, VPM, 0);    // VALU
            __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);      // DS write
        }
        if (NSET == 1 && more2 && !(GA_EXP & 2)) issue_tile(prod);
        if (!(GA_EXP & 4)) __syncthreads();
    };
    // iteration t: LDS stage (t - t_begin) & 1 holds tile t; with two register sets, set (t + 1 - t_begin) & 1 holds
    // tile t+1 and tile t+2 is loaded into set (t - t_begin) & 1; with one set, tile t+2 is loaded after the split.
    int t = t_begin;
    if (NSET == 1 && t + 1 < t_end) issue_tile(0);      // tile t_begin+1
    for (; t + 1 < t_end; t += 2) {
        step(0, NSET == 2 ? 1 : 0, 0, t + 2 < t_end);
        if (t + 2 >= t_end) break;
        step(1, 0, NSET == 2 ? 1 : 0, t + 3 < t_end);
    }
    const int buf = (t_end - 1 - t_begin) & 1;          // the last tile's stage
    if (t_begin < t_end) mma_tile(buf);
    __syncthreads();        // every wave is done reading the operand tiles before the epilogue reuses the LDS
    GA_STAMP(3)

    conv_epilogue<WM, WN, TM, TN>(d, acc, smem, m0, n0, M, vec_out, splits, split);
    GA_STAMP(4)
}

template <int WM, int WN, int TM, int TN, int AFF, int ACT, bool DUAL>
static void launch_bf3_inst(const ga_conv_desc& d, hipStream_t stream, dim3 grid, size_t lds, int tilesN, int M, int Ctot,
                            int Ktot, int nkc, int vec_out) {
    const fastdiv fd_howo = make_fastdiv(d.Ho * d.Wo), fd_wo = make_fastdiv(d.Wo);
    static dyn_lds_cache attr;
    (void)ensure_dyn_lds(attr, reinterpret_cast<const void*>(&conv_bf3_kernel<WM, WN, TM, TN, AFF, ACT, DUAL>), lds);
    hipLaunchKernelGGL((conv_bf3_kernel<WM, WN, TM, TN, AFF, ACT, DUAL>), grid, dim3(256), lds, stream, d, tilesN, M, Ctot, Ktot, nkc, vec_out,
                       fd_howo, fd_wo);
}

// (dual << 8) | (affine kind << 4) | activation — the key of the instantiated prologue variants
static inline int conv_bf3_mode(const ga_conv_desc& d) {
    return ((d.C2 > 0 ? 1 : 0) << 8) | ((d.pro_scale ? (d.pro_per_row ? 2 : 1) : 0) << 4) | d.pro_act;
}

// 1 when conv_bf3 has a kernel for this descriptor's prologue (ga_conv2d falls back to the fp32 kernel otherwise)
int conv_bf3_supports(const ga_conv_desc& d) {
    switch (conv_bf3_mode(d)) {
        case 0x000: case 0x100: case 0x001: case 0x002: case 0x003: case 0x004: case 0x010: case 0x011: case 0x020: case 0x021: return 1;
        default: return 0;
    }
}

template <int WM, int WN, int TM, int TN>
static int launch_bf3(const ga_conv_desc& d, hipStream_t stream, int vec_out, int splits) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    const int M = d.N * d.Ho * d.Wo;
    const int Ctot = d.C1 + d.C2;
    const int Ktot = d.KH * d.KW * Ctot;
    const int nkc = (Ctot + BK3 - 1) / BK3;
    const int tilesM = (M + BM - 1) / BM, tilesN = (d.Cout + BN - 1) / BN;
    size_t lds = (size_t)2 * (2 * BM + 2 * BN) * LDB * 2;
    const size_t lds_c = (size_t)BM * (BN + 4) * sizeof(float);
    if (lds_c > lds) lds = lds_c;
    const dim3 grid(tilesM * tilesN, splits);
#define GA_BF3(A, C, D) launch_bf3_inst<WM, WN, TM, TN, A, C, D>(d, stream, grid, lds, tilesN, M, Ctot, Ktot, nkc, vec_out)
    switch (conv_bf3_mode(d)) {         // the prologue combinations the purification / classifier plans contain
        case 0x000: GA_BF3(0, GA_ACT_NONE, false); break;
        case 0x100: GA_BF3(0, GA_ACT_NONE, true); break;
        case 0x001: GA_BF3(0, GA_ACT_SILU, false); break;
        case 0x002: GA_BF3(0, GA_ACT_ELU, false); break;
        case 0x003: GA_BF3(0, GA_ACT_RELU, false); break;
        case 0x004: GA_BF3(0, GA_ACT_LRELU, false); break;
        case 0x010: GA_BF3(1, GA_ACT_NONE, false); break;
        case 0x011: GA_BF3(1, GA_ACT_SILU, false); break;
        case 0x020: GA_BF3(2, GA_ACT_NONE, false); break;
        case 0x021: GA_BF3(2, GA_ACT_SILU, false); break;
        default: return GA_E_UNSUPPORTED;
    }
#undef GA_BF3
    return check_launch();
}

// called by ga_conv2d (conv_mfma.hip) once the descriptor has been validated and the buffer extents filled in
int conv_bf3_dispatch(const ga_conv_desc& d, hipStream_t stream, int tile, int vec_out, int splits) {
    switch (tile) {
        case 1: return launch_bf3<2, 2, 2, 2>(d, stream, vec_out, splits);
#ifdef GA_TRACE_TILE1_ONLY      // quick experiment builds
        default: return GA_E_UNSUPPORTED;
    }
    switch (tile) {
#endif
        case 2: return launch_bf3<4, 1, 1, 2>(d, stream, vec_out, splits);
        case 3: return launch_bf3<2, 2, 1, 1>(d, stream, vec_out, splits);
        case 4: return launch_bf3<4, 1, 1, 1>(d, stream, vec_out, splits);
        default: return GA_E_UNSUPPORTED;
    }
}

// host helper exported for weight preparation: w (fp32, n elements) -> hi, lo (bf16 bit patterns)
__global__ void split_bf16_kernel(const float* w, __bf16* hi, __bf16* lo, const long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float v = w[i];
        const __bf16 h = (__bf16)v;
        hi[i] = h;
        lo[i] = (__bf16)(v - (float)h);
    }
}

}  // namespace ga

#ifdef GA_TRACE
extern "C" int ga_debug_trace_read(unsigned long long* out, int n) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(ga::ga_trace_buf), (size_t)n * 8) == hipSuccess ? GA_OK : GA_E_LAUNCH;
}
#endif

extern "C" int ga_split_bf16(const float* w, void* hi, void* lo, long n, void* stream) {
    ga::clear_stale_error();
    if (!w || !hi || !lo || n <= 0) return GA_E_BADARG;
    long blocks = (n + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(ga::split_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, w,
                       (__bf16*)hi, (__bf16*)lo, n);
    return ga::check_launch();
}
