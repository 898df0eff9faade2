// ga_conv2d — fp32 implicit-GEMM convolution on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32).
//
// GEMM view: C[M = N*Ho*Wo pixels][Cout] = A[M][K = taps*(C1+C2)] * B[K][Cout].
//   * A is gathered on the fly from the NHWC activation (im2col never materialised); the prologue
//     (per-channel / per-(row,channel) affine + SiLU/ELU/ReLU) is applied in registers between the global load
//     and the LDS write, so activations are stored once, pre-activation, and re-used by the backward pass.
//   * B (weights) is stored [Cout][K] so that both operands are "row x contiguous-k" in LDS.
//   * LDS rows are BK=32 floats + 4 pad (144 B): a ds_read_b128 of 4 consecutive k for 16 different rows hits 16
//     distinct 16-B slots (9*i mod 16 is a bijection) -> conflict-free for every ds_read_b128 lane group.
//   * One ds_read_b128 per operand feeds FOUR MFMAs: within an 8-wide k block, lane half h (= lane>>5) owns
//     k = 4h..4h+3 and MFMA step s uses k = 4h+s for A and B alike (the order of the fma chain inside K is free).
//   * 256 threads = 4 waves arranged WM x WN, each wave owns TM x TN tiles of 32x32; accumulators stay in registers
//     for the whole K loop; global->register prefetch of tile t+1 overlaps the MFMAs of tile t (double-buffered
//     LDS, one barrier per K tile).
//   * Epilogue: the accumulator tile is transposed through the (now free) LDS so that every lane handles 4
//     consecutive channels of one pixel: bias / act' (reads the saved forward input) / addends / store are all 16-B
//     accesses on 512-B contiguous rows.  A scalar epilogue remains for channel counts that are not multiples of 4.
//   * Split-K (small M x Cout with a long K: samplers, the classifier head): grid.y = splits, raw partial tiles go to
//     a caller-provided workspace and a second kernel sums them in a fixed order and applies the epilogue.
//   * blockIdx is remapped so that consecutive logical tiles (which share the A panel) run on one XCD's L2.
#include "ga_common.h"
#include "conv_epilogue.h"

namespace ga {

constexpr int BK = 32;
constexpr int LDK = 36;

// PRO: 0 = no prologue, 1 = per-channel affine and/or activation, 2 = per-(row,channel) affine (+ activation)
// FAST: stride-1 gathers through buffer loads — per output row a byte offset and a tap-validity bitmask are computed
//       ONCE per workgroup; per K tile a row costs one add + one select (out-of-range lanes get an offset beyond the
//       buffer and the hardware returns zeros), the tap/channel displacement rides in the scalar offset.  On gfx950
//       the f32 MFMA runs at the vector-ALU rate, so every VALU instruction saved in the K loop is MFMA time won.
template <int WM, int WN, int TM, int TN, bool VEC, int PRO, bool FAST>
__global__ void __launch_bounds__(256)
conv_mfma_kernel(const ga_conv_desc d, const int tilesN, const int M, const int Ctot, const int Ktot, const int nkc,
                 const int vec_out, const fastdiv fd_howo, const fastdiv fd_wo) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    constexpr int RA = BM / 32, RB = BN / 32;
    constexpr int LDC = BN + 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;
    float* Bs = smem + 2 * BM * LDK;

    // ---- XCD-aware tile order: blocks b, b+8, b+16.. share an XCD (round-robin dispatch); give each XCD a
    //      contiguous range of logical tiles.  Bijective for any grid size.
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
    const int c4 = tid & 7, r0 = tid >> 3;

    const int HoWo = d.Ho * d.Wo;
    int a_n[RA], a_h0[RA], a_w0[RA];
#pragma unroll
    for (int i = 0; i < RA; ++i) {
        const int m = m0 + r0 + 32 * i;
        if (m < M) {
            const int n = fd_div(m, fd_howo), rem = m - n * HoWo, ho = fd_div(rem, fd_wo), wo = rem - ho * d.Wo;
            a_n[i] = n; a_h0[i] = ho * d.sn - d.pad; a_w0[i] = wo * d.sn - d.pad;
        } else { a_n[i] = -1; a_h0[i] = 0; a_w0[i] = 0; }
    }

    // ---- FAST-path invariants
    constexpr int INV = 0x7fffffff;                                     // beyond any buffer we accept (< 2 GiB)
    int baseA[RA], baseA2[RA], baseB[RB];
    unsigned maskA[RA];
    __amdgpu_buffer_rsrc_t rsrcX, rsrcX2, rsrcW;
    if (FAST) {
        rsrcX = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.x), 0, d.x_bytes, 0x00020000);
        rsrcX2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.C2 > 0 ? d.x2 : d.x), 0, d.C2 > 0 ? d.x2_bytes : d.x_bytes, 0x00020000);
        rsrcW = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.w), 0, d.w_bytes, 0x00020000);
#pragma unroll
        for (int i = 0; i < RA; ++i) {
            const int pixb = (a_n[i] * d.Hi + a_h0[i]) * d.Wi + a_w0[i];
            baseA[i] = pixb * d.ldx * 4 + c4 * 16;
            baseA2[i] = pixb * d.ldx2 * 4 + c4 * 16;
            maskA[i] = a_n[i] >= 0 ? tap_mask(a_h0[i], a_w0[i], d.Hi, d.Wi, d.KH, d.KW) : 0u;
        }
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            const int co = n0 + r0 + 32 * i;
            baseB[i] = co < d.Cout ? (co * Ktot + 4 * c4) * 4 : INV;
        }
    }

    // ---- tile staging, split in two so that the global loads of tile t+1 are IN FLIGHT while tile t's MFMAs run:
    //      issue_tile(): address math + unconditional loads (out-of-range lanes read a safe address and are zeroed
    //      later: no divergent branch, hence no s_waitcnt, sits between a load and the MFMAs);
    //      finish_tile(): prologue math on the landed registers + LDS writes, after the MFMAs.
    floatx4 ra[RA], rb[RB], rs[PRO == 2 ? RA : 1], rt[PRO == 2 ? RA : 1];
    unsigned okmask = 0;
    int cur_c = 0;
    const int sd_shift = d.sd > 1 ? 31 - __builtin_clz(d.sd) : 0, sd_mask = d.sd - 1;   // sd is a power of two

    // K-tile cursor kept in scalar registers and advanced incrementally (tiles are issued in order): no integer
    // division in the loop and everything derived from it is provably wave-uniform (buffer descriptors, soffsets).
    int q_tap = 0, q_chunk = 0, q_kh = 0, q_kw = 0;
    auto seek_tile = [&](const int t) __attribute__((always_inline)) {
        q_tap = __builtin_amdgcn_readfirstlane(t / nkc);
        q_chunk = __builtin_amdgcn_readfirstlane(t - q_tap * nkc);
        q_kh = __builtin_amdgcn_readfirstlane(q_tap / d.KW);
        q_kw = q_tap - q_kh * d.KW;
    };

    auto issue_tile = [&](const int) __attribute__((always_inline)) {
        const int tap = q_tap, c0 = q_chunk * BK;
        const int kh = q_kh, kw = q_kw;
        if (++q_chunk == nkc) { q_chunk = 0; ++q_tap; if (++q_kw == d.KW) { q_kw = 0; ++q_kh; } }
        const int c = c0 + 4 * c4;
        cur_c = c;
        okmask = 0;
        if (FAST) {
            const bool in_x = c0 < d.C1;                                 // uniform: a chunk never straddles the sources
            const int lim = (in_x ? d.C1 : Ctot) - c0;
            const bool cval = 4 * c4 < lim;
            const int delta = (kh * d.Wi + kw) * (in_x ? d.ldx : d.ldx2) * 4;
            const int soffA = (in_x ? c0 : c0 - d.C1) * 4;
            const unsigned bit = 1u << tap;
            if (in_x) {
#pragma unroll
                for (int i = 0; i < RA; ++i) {
                    const bool valid = cval & ((maskA[i] & bit) != 0);
                    const int off = valid ? baseA[i] + delta : INV;
                    okmask |= (valid ? 1u : 0u) << i;
                    ra[i] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(rsrcX, off, soffA, 0));
                }
            } else {
#pragma unroll
                for (int i = 0; i < RA; ++i) {
                    const bool valid = cval & ((maskA[i] & bit) != 0);
                    const int off = valid ? baseA2[i] + delta : INV;
                    okmask |= (valid ? 1u : 0u) << i;
                    ra[i] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(rsrcX2, off, soffA, 0));
                }
            }
            const int soffB = (tap * Ctot + c0) * 4;
#pragma unroll
            for (int i = 0; i < RB; ++i) {
                const int off = cval ? baseB[i] : INV;
                rb[i] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(rsrcW, off, soffB, 0));
            }
            if (PRO == 1) {
                if (d.pro_scale) {
                    const int pc = (in_x & cval) ? c : 0;
                    rs[0] = *reinterpret_cast<const floatx4*>(d.pro_scale + pc);
                    rt[0] = *reinterpret_cast<const floatx4*>(d.pro_shift + pc);
                }
            } else if (PRO == 2) {
#pragma unroll
                for (int i = 0; i < RA; ++i) {
                    const size_t po = (in_x && ((okmask >> i) & 1u)) ? (size_t)a_n[i] * d.C1 + c : 0;
                    rs[i] = *reinterpret_cast<const floatx4*>(d.pro_scale + po);
                    rt[i] = *reinterpret_cast<const floatx4*>(d.pro_shift + po);
                }
            }
        } else if (VEC) {
            const bool first = c < d.C1;
            const bool cok = c < Ctot;
            const float* src[RA];
            const float* wsrc[RB];
#pragma unroll
            for (int i = 0; i < RA; ++i) {
                int hi = a_h0[i] + kh, wi = a_w0[i] + kw;
                bool ok = cok & (a_n[i] >= 0) & (hi >= 0) & (wi >= 0) & (((hi | wi) & sd_mask) == 0);
                hi >>= sd_shift; wi >>= sd_shift;
                ok = ok & (hi < d.Hi) & (wi < d.Wi);
                const size_t pix = ((size_t)a_n[i] * d.Hi + hi) * d.Wi + wi;
                const size_t o1 = ok ? pix * d.ldx + c : 0;
                const size_t o2 = ok ? pix * d.ldx2 + (c - d.C1) : 0;
                src[i] = (first | !ok) ? d.x + o1 : d.x2 + o2;
                okmask |= (ok ? 1u : 0u) << i;
            }
#pragma unroll
            for (int i = 0; i < RB; ++i) {
                const int co = n0 + r0 + 32 * i;
                const bool ok = (co < d.Cout) & cok;
                wsrc[i] = d.w + (ok ? (size_t)co * Ktot + (size_t)tap * Ctot + c : 0);
                okmask |= (ok ? 1u : 0u) << (16 + i);
            }
#pragma unroll
            for (int i = 0; i < RA; ++i) ra[i] = *reinterpret_cast<const floatx4*>(src[i]);
#pragma unroll
            for (int i = 0; i < RB; ++i) rb[i] = *reinterpret_cast<const floatx4*>(wsrc[i]);
            if (PRO == 1) {
                if (d.pro_scale) {
                    const int pc = first ? c : 0;
                    rs[0] = *reinterpret_cast<const floatx4*>(d.pro_scale + pc);
                    rt[0] = *reinterpret_cast<const floatx4*>(d.pro_shift + pc);
                }
            } else if (PRO == 2) {
#pragma unroll
                for (int i = 0; i < RA; ++i) {
                    const size_t po = (first && ((okmask >> i) & 1u)) ? (size_t)a_n[i] * d.C1 + c : 0;
                    rs[i] = *reinterpret_cast<const floatx4*>(d.pro_scale + po);
                    rt[i] = *reinterpret_cast<const floatx4*>(d.pro_shift + po);
                }
            }
        } else {
            // scalar path (channel counts that are not multiples of 4: the 3-channel image, odd latent sizes)
#pragma unroll
            for (int i = 0; i < RA; ++i) {
                floatx4 v = {0.f, 0.f, 0.f, 0.f};
                if (a_n[i] >= 0 && c < Ctot) {
                    int hi = a_h0[i] + kh, wi = a_w0[i] + kw;
                    bool ok = (hi >= 0) & (wi >= 0);
                    if (d.sd != 1) { ok = ok && (hi % d.sd == 0) && (wi % d.sd == 0); hi /= d.sd; wi /= d.sd; }
                    ok = ok && hi < d.Hi && wi < d.Wi;
                    if (ok) {
                        const size_t pix = ((size_t)a_n[i] * d.Hi + hi) * d.Wi + wi;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int ce = c + e;
                            if (ce < d.C1) {
                                float u = d.x[pix * d.ldx + ce];
                                if (d.pro_scale && (d.flags & GA_CONV_PRO_PRELU)) {
                                    u = u > 0.f ? u : u * d.pro_scale[ce];
                                } else if (d.pro_scale) {
                                    const size_t po = (d.pro_per_row ? (size_t)a_n[i] * d.C1 : 0) + ce;
                                    u = u * d.pro_scale[po] + d.pro_shift[po];
                                }
                                v[e] = act_fwd_fast(u, d.pro_act);
                            } else if (ce < Ctot) {
                                v[e] = d.x2[pix * d.ldx2 + (ce - d.C1)];
                            }
                        }
                    }
                }
                ra[i] = v;
            }
#pragma unroll
            for (int i = 0; i < RB; ++i) {
                floatx4 v = {0.f, 0.f, 0.f, 0.f};
                const int co = n0 + r0 + 32 * i;
                if (co < d.Cout && c < Ctot) {
                    const float* wp = d.w + (size_t)co * Ktot + (size_t)tap * Ctot + c;
#pragma unroll
                    for (int e = 0; e < 4; ++e) if (c + e < Ctot) v[e] = wp[e];
                }
                rb[i] = v;
            }
        }
    };

    auto finish_tile = [&](const int buf) __attribute__((always_inline)) {
        float* Ab = As + buf * BM * LDK;
        float* Bb = Bs + buf * BN * LDK;
        const floatx4 zero = {0.f, 0.f, 0.f, 0.f};
        if (FAST) {
            const bool first = cur_c < d.C1;
#pragma unroll
            for (int i = 0; i < RA; ++i) {
                floatx4 v = ra[i];
                if (PRO != 0) {
                    floatx4 pv = v;
                    if (PRO == 2) pv = pv * rs[i] + rt[i];
                    else if (d.pro_scale && (d.flags & GA_CONV_PRO_PRELU)) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) pv[e] = pv[e] > 0.f ? pv[e] : pv[e] * rs[0][e];
                    } else if (d.pro_scale) pv = pv * rs[0] + rt[0];
                    if (d.pro_act == GA_ACT_SILU) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) pv[e] = pv[e] * fast_sigmoid(pv[e]);
                    } else if (d.pro_act != GA_ACT_NONE) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) pv[e] = act_fwd_fast(pv[e], d.pro_act);
                    }
                    v = first ? pv : v;
                    v = (okmask >> i) & 1u ? v : zero;                  // act(shift) != 0 on padded taps
                }
                *reinterpret_cast<floatx4*>(Ab + (r0 + 32 * i) * LDK + 4 * c4) = v;
            }
#pragma unroll
            for (int i = 0; i < RB; ++i) *reinterpret_cast<floatx4*>(Bb + (r0 + 32 * i) * LDK + 4 * c4) = rb[i];
        } else if (VEC) {
            const bool first = cur_c < d.C1;
#pragma unroll
            for (int i = 0; i < RA; ++i) {
                floatx4 v = ra[i];
                if (PRO != 0) {
                    floatx4 pv = v;
                    if (PRO == 2) pv = pv * rs[i] + rt[i];
                    else if (d.pro_scale && (d.flags & GA_CONV_PRO_PRELU)) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) pv[e] = pv[e] > 0.f ? pv[e] : pv[e] * rs[0][e];
                    } else if (d.pro_scale) pv = pv * rs[0] + rt[0];
                    if (d.pro_act == GA_ACT_SILU) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) pv[e] = pv[e] * fast_sigmoid(pv[e]);
                    } else if (d.pro_act != GA_ACT_NONE) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) pv[e] = act_fwd_fast(pv[e], d.pro_act);
                    }
                    v = first ? pv : v;
                }
                v = (okmask >> i) & 1u ? v : zero;
                *reinterpret_cast<floatx4*>(Ab + (r0 + 32 * i) * LDK + 4 * c4) = v;
            }
#pragma unroll
            for (int i = 0; i < RB; ++i)
                *reinterpret_cast<floatx4*>(Bb + (r0 + 32 * i) * LDK + 4 * c4) = (okmask >> (16 + i)) & 1u ? rb[i] : zero;
        } else {
#pragma unroll
            for (int i = 0; i < RA; ++i) *reinterpret_cast<floatx4*>(Ab + (r0 + 32 * i) * LDK + 4 * c4) = ra[i];
#pragma unroll
            for (int i = 0; i < RB; ++i) *reinterpret_cast<floatx4*>(Bb + (r0 + 32 * i) * LDK + 4 * c4) = rb[i];
        }
    };

    floatx16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // ---- K range of this block (split-K: grid.y slices of whole K tiles)
    const int T = d.KH * d.KW * nkc;
    const int splits = gridDim.y, split = blockIdx.y;
    const int tper = (T + splits - 1) / splits;
    const int t_begin = split * tper;
    const int t_end = min(T, t_begin + tper);

    const int lrow = lane & 31, lh = lane >> 5;
    if (t_begin < t_end) {
        seek_tile(t_begin);
        issue_tile(t_begin);
        finish_tile(0);
    }
    __syncthreads();
    for (int t = t_begin; t < t_end; ++t) {
        const int buf = (t - t_begin) & 1;
        if (t + 1 < t_end) issue_tile(t + 1);
        const float* Ab = As + buf * BM * LDK + (wm * TM * 32 + lrow) * LDK + 4 * lh;
        const float* Bb = Bs + buf * BN * LDK + (wn * TN * 32 + lrow) * LDK + 4 * lh;
#pragma unroll
        for (int kk = 0; kk < BK / 8; ++kk) {
            floatx4 af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const floatx4*>(Ab + i * 32 * LDK + kk * 8);
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const floatx4*>(Bb + j * 32 * LDK + kk * 8);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][s], bf[j][s], acc[i][j], 0, 0, 0);
        }
        if (t + 1 < t_end) finish_tile(buf ^ 1);
        __syncthreads();
    }

    conv_epilogue<WM, WN, TM, TN>(d, acc, smem, m0, n0, M, vec_out, splits, split);
}

// sums the split-K partial tiles in split order (bitwise reproducible) and applies the epilogue
__global__ void __launch_bounds__(256) conv_splitk_reduce_kernel(const ga_conv_desc d, const int M, const int splits, const int vec_out) {
    const int HoWo = d.Ho * d.Wo;
    const size_t slab = (size_t)M * d.Cout;
    if (vec_out) {
        const long total4 = (long)slab / 4;
        const int C4 = d.Cout / 4;
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long)gridDim.x * 256) {
            floatx4 v = *reinterpret_cast<const floatx4*>(d.ws + i * 4);
            for (int s = 1; s < splits; ++s) v += *reinterpret_cast<const floatx4*>(d.ws + s * slab + i * 4);
            epilogue4(d, (int)(i / C4), (int)(i % C4) * 4, v, HoWo);
        }
    } else {
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < (long)slab; i += (long)gridDim.x * 256) {
            float v = d.ws[i];
            for (int s = 1; s < splits; ++s) v += d.ws[s * slab + i];
            epilogue1(d, (int)(i / d.Cout), (int)(i % d.Cout), v, HoWo);
        }
    }
}

template <int WM, int WN, int TM, int TN, bool VEC, int PRO, bool FAST>
static void launch_inst(const ga_conv_desc& d, hipStream_t stream, dim3 grid, size_t lds, int tilesN, int M, int Ctot,
                        int Ktot, int nkc, int vec_out) {
    static dyn_lds_cache attr;
    (void)ensure_dyn_lds(attr, reinterpret_cast<const void*>(&conv_mfma_kernel<WM, WN, TM, TN, VEC, PRO, FAST>), lds);
    hipLaunchKernelGGL((conv_mfma_kernel<WM, WN, TM, TN, VEC, PRO, FAST>), grid, dim3(256), lds, stream, d, tilesN, M, Ctot, Ktot, nkc, vec_out,
                       make_fastdiv(d.Ho * d.Wo), make_fastdiv(d.Wo));
}

template <int WM, int WN, int TM, int TN>
static int launch_conv(const ga_conv_desc& d, hipStream_t stream, bool vec, int vec_out, int splits) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    const int M = d.N * d.Ho * d.Wo;
    const int Ctot = d.C1 + d.C2;
    const int Ktot = d.KH * d.KW * Ctot;
    const int nkc = (Ctot + BK - 1) / BK;
    const int tilesM = (M + BM - 1) / BM, tilesN = (d.Cout + BN - 1) / BN;
    size_t lds = (size_t)2 * (BM + BN) * LDK * sizeof(float);
    const size_t lds_c = (size_t)BM * (BN + 4) * sizeof(float);
    if (lds_c > lds) lds = lds_c;
    const dim3 grid(tilesM * tilesN, splits);
    const int pro = d.pro_scale && d.pro_per_row ? 2 : ((d.pro_scale || d.pro_act) ? 1 : 0);
    const bool fast = vec && d.sd == 1 && (d.C2 == 0 || d.C1 % BK == 0) && d.KH * d.KW <= 32 &&
                      d.x_bytes > 0 && d.w_bytes > 0 && (d.C2 == 0 || d.x2_bytes > 0);
    if (fast) {
        if (pro == 0) launch_inst<WM, WN, TM, TN, true, 0, true>(d, stream, grid, lds, tilesN, M, Ctot, Ktot, nkc, vec_out);
        else if (pro == 1) launch_inst<WM, WN, TM, TN, true, 1, true>(d, stream, grid, lds, tilesN, M, Ctot, Ktot, nkc, vec_out);
        else launch_inst<WM, WN, TM, TN, true, 2, true>(d, stream, grid, lds, tilesN, M, Ctot, Ktot, nkc, vec_out);
    } else if (vec) {
        if (pro == 0) launch_inst<WM, WN, TM, TN, true, 0, false>(d, stream, grid, lds, tilesN, M, Ctot, Ktot, nkc, vec_out);
        else if (pro == 1) launch_inst<WM, WN, TM, TN, true, 1, false>(d, stream, grid, lds, tilesN, M, Ctot, Ktot, nkc, vec_out);
        else launch_inst<WM, WN, TM, TN, true, 2, false>(d, stream, grid, lds, tilesN, M, Ctot, Ktot, nkc, vec_out);
    } else {
        launch_inst<WM, WN, TM, TN, false, 1, false>(d, stream, grid, lds, tilesN, M, Ctot, Ktot, nkc, vec_out);
    }
    return check_launch();
}

int conv_bf3_dispatch(const ga_conv_desc& d, hipStream_t stream, int tile, int vec_out, int splits);   // conv_bf3.hip
int conv_bf3_supports(const ga_conv_desc& d);
int conv_halo3_dispatch(const ga_conv_desc& d, hipStream_t stream, int tile, int vec_out, int splits);  // conv_halo3.hip
int conv_halo3_supports(const ga_conv_desc& d);
int conv_thin3_dispatch(const ga_conv_desc& d, hipStream_t stream, int vec_out, int splits);             // conv_thin3.hip
int conv_thin3_supports(const ga_conv_desc& d);

}  // namespace ga

// byte extent of one operand above which ga_conv2d convolves row sub-batches (the fast loaders' 31-bit offsets); lowered by
// ga_debug_set_conv_row_limit so that tests reach the sub-batch path at small sizes
static long g_conv_row_limit = 0x7fffff00L;
static long ga_conv_row_limit() { return g_conv_row_limit; }
extern "C" long ga_debug_set_conv_row_limit(long bytes) {
    const long old = g_conv_row_limit;
    g_conv_row_limit = bytes > 0 ? bytes : 0x7fffff00L;
    return old;
}

extern "C" int ga_conv2d(const ga_conv_desc* dp, void* stream_) {
    ga::clear_stale_error();
    using namespace ga;
    if (!dp) return GA_E_BADARG;
    const ga_conv_desc& d = *dp;
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    if (!d.x || !d.w || !d.y) return GA_E_BADARG;
    if (d.N <= 0 || d.Hi <= 0 || d.Wi <= 0 || d.C1 <= 0 || d.C2 < 0 || d.Ho <= 0 || d.Wo <= 0 || d.Cout <= 0) return GA_E_BADARG;
    if (d.KH <= 0 || d.KW <= 0 || d.sn <= 0 || d.sd <= 0 || d.pad < 0) return GA_E_BADARG;
    if (d.sd & (d.sd - 1)) return GA_E_UNSUPPORTED;          // transposed stride must be a power of two
    if (d.C2 > 0 && !d.x2) return GA_E_BADARG;
    if ((d.pro_scale == nullptr) != (d.pro_shift == nullptr)) return GA_E_BADARG;
    if (d.dact_x && ((d.dact_scale == nullptr) != (d.dact_shift == nullptr))) return GA_E_BADARG;
    if (d.ldx < d.C1 || (d.C2 > 0 && d.ldx2 < d.C2) || d.ldy < d.Cout) return GA_E_BADARG;
    if (d.addend && d.ldadd < d.Cout) return GA_E_BADARG;
    if (d.addend2 && d.ldadd2 < d.Cout) return GA_E_BADARG;
    if (d.dact_x && d.lddact < d.Cout) return GA_E_BADARG;
    if (d.dact_x && d.dact_rep > 1 && d.N % d.dact_rep) return GA_E_BADARG;
    // Rows are independent: a tensor beyond the fast loader's 31-bit byte offsets (2 GB; StyleGAN2's 1024^2 x 32-channel maps
    // at a few dozen rows) is convolved in sub-batches of rows that fit, each an ordinary launch on the same stream.
    {
        const long row_x = (long)d.Hi * d.Wi * d.ldx * 4, row_x2 = d.C2 > 0 ? (long)d.Hi * d.Wi * d.ldx2 * 4 : 0;
        long row_max = row_x > row_x2 ? row_x : row_x2;
        const long pix_out = (long)d.Ho * d.Wo * 4;          // every per-row operand counts: output, addends, act' source
        if (pix_out * d.ldy > row_max) row_max = pix_out * d.ldy;
        if (d.addend && !d.addend_bcast_n && pix_out * d.ldadd > row_max) row_max = pix_out * d.ldadd;
        if (d.addend2 && pix_out * d.ldadd2 > row_max) row_max = pix_out * d.ldadd2;
        if (d.dact_x && pix_out * d.lddact > row_max) row_max = pix_out * d.lddact;
        const long lim = ga_conv_row_limit();
        if (d.N > 1 && row_max * d.N >= lim && row_max < lim) {
            int sub = (int)((lim - 1) / row_max);
            const int arep = d.addend && !d.addend_bcast_n && d.addend_rep > 1 ? d.addend_rep : 1;
            const int drep = d.dact_x && d.dact_rep > 1 ? d.dact_rep : 1;
            if (arep > 1 || drep > 1) {                      // operands shared by consecutive rows: cut at their row boundaries
                const long both = (long)arep * drep;         // (a common multiple; the two never meet on one launch in practice)
                if (d.N % arep || d.N % drep) return GA_E_BADARG;
                sub -= (int)(sub % both);
                if (sub < both) return GA_E_UNSUPPORTED;
            }
            for (int n0 = 0; n0 < d.N; n0 += sub) {
                ga_conv_desc s = d;
                s.N = d.N - n0 < sub ? d.N - n0 : sub;
                const size_t pin = (size_t)n0 * d.Hi * d.Wi, pout = (size_t)n0 * d.Ho * d.Wo;
                s.x = d.x + pin * d.ldx;
                if (d.x2) s.x2 = d.x2 + pin * d.ldx2;
                s.y = d.y + pout * d.ldy;
                if (d.addend && !d.addend_bcast_n) s.addend = d.addend + (size_t)(n0 / arep) * d.Ho * d.Wo * d.ldadd;
                if (d.addend2) s.addend2 = d.addend2 + pout * d.ldadd2;
                if (d.dact_x) s.dact_x = d.dact_x + (size_t)(n0 / drep) * d.Ho * d.Wo * d.lddact;
                if (d.pro_scale && d.pro_per_row) {
                    s.pro_scale = d.pro_scale + (size_t)n0 * d.C1;
                    s.pro_shift = d.pro_shift + (size_t)n0 * d.C1;
                }
                const int rc = ga_conv2d(&s, stream_);
                if (rc != GA_OK) return rc;
            }
            return GA_OK;
        }
    }
    if ((long)d.N * d.Ho * d.Wo > 0x7fffffffL) return GA_E_UNSUPPORTED;
    if ((long)d.KH * d.KW * (d.C1 + d.C2) > 0x7fffffffL) return GA_E_UNSUPPORTED;
    const long M = (long)d.N * d.Ho * d.Wo;
    int splits = d.splits <= 1 ? 1 : d.splits;
    if (splits > 1) {
        if (!d.ws || d.ws_floats < (long)splits * M * d.Cout) return GA_E_BADARG;
        const int T = d.KH * d.KW * ((d.C1 + d.C2 + 31) / 32);
        if (splits > T) splits = T;
        if (splits > 65535) return GA_E_UNSUPPORTED;
    }

    const bool vec = (d.C1 % 4 == 0) && (d.C2 % 4 == 0) && (d.ldx % 4 == 0) && (d.C2 == 0 || d.ldx2 % 4 == 0) &&
                     aligned16(d.x) && aligned16(d.w) && (d.C2 == 0 || aligned16(d.x2)) &&
                     (!d.pro_scale || (aligned16(d.pro_scale) && aligned16(d.pro_shift)));
    const bool vec_out = (d.Cout % 4 == 0) && (d.ldy % 4 == 0) && aligned16(d.y) && (!d.bias || aligned16(d.bias)) &&
                         (!d.addend || (d.ldadd % 4 == 0 && aligned16(d.addend))) &&
                         (!d.addend2 || (d.ldadd2 % 4 == 0 && aligned16(d.addend2))) &&
                         (!d.dact_x || (d.lddact % 4 == 0 && aligned16(d.dact_x) &&
                                        (!d.dact_scale || (aligned16(d.dact_scale) && aligned16(d.dact_shift))))) &&
                         (splits == 1 || aligned16(d.ws));

    int tile = d.tile;
    if (tile == 0) {
        if (d.Cout <= 32) tile = 4;
        else if (d.Cout <= 64) tile = (M >= 128 * 256) ? 2 : 3;
        else {
            const long b128 = ((M + 127) / 128) * ((d.Cout + 127) / 128);
            const long b12864 = ((M + 127) / 128) * ((d.Cout + 63) / 64);
            if (b128 >= 512) tile = 1;
            else if (b12864 >= 512) tile = 2;
            else tile = 3;
        }
    }
    // buffer extents for the FAST loader (0 = tensor too large for 31-bit byte offsets -> generic loader)
    ga_conv_desc k = d;
    const long xb = ((long)d.N * d.Hi * d.Wi - 1) * d.ldx * 4 + (long)d.C1 * 4;
    const long x2b = d.C2 > 0 ? ((long)d.N * d.Hi * d.Wi - 1) * d.ldx2 * 4 + (long)d.C2 * 4 : 0;
    const long wb = (long)d.Cout * d.KH * d.KW * (d.C1 + d.C2) * 4;
    const long lim = 0x7fffff00L;
    k.x_bytes = xb < lim ? (unsigned)xb : 0;
    k.x2_bytes = x2b < lim ? (unsigned)x2b : 0;
    k.w_bytes = wb < lim ? (unsigned)wb : 0;
    const int Ctot = d.C1 + d.C2;
    const bool bf3 = d.w_hi && d.w_lo && vec && d.sd == 1 && (d.C2 == 0 || d.C1 % 32 == 0) && (Ctot % 8 == 0) &&
                     d.KH * d.KW <= 32 && k.x_bytes > 0 && k.w_bytes > 0 && (d.C2 == 0 || k.x2_bytes > 0) &&
                     aligned16(d.w_hi) && aligned16(d.w_lo) && conv_bf3_supports(d);
    int rc;
    if (tile == 11) {                   // persistent weights-resident 3x3 for 32 input channels (explicit request only; w_frag in the thin order)
        if (!(bf3 && vec_out && conv_thin3_supports(d))) return GA_E_UNSUPPORTED;
        rc = conv_thin3_dispatch(k, stream, vec_out, splits);
    } else if (tile >= 5 && tile <= 10) {      // halo-staged 3x3 (explicit request only: the tune table names it per shape)
        if (!(bf3 && vec_out && conv_halo3_supports(d))) return GA_E_UNSUPPORTED;
        rc = conv_halo3_dispatch(k, stream, tile, vec_out, splits);
    } else if (bf3) {
        rc = conv_bf3_dispatch(k, stream, tile, vec_out, splits);
    } else {
        switch (tile) {
            case 1: rc = launch_conv<2, 2, 2, 2>(k, stream, vec, vec_out, splits); break;
            case 2: rc = launch_conv<4, 1, 1, 2>(k, stream, vec, vec_out, splits); break;
            case 3: rc = launch_conv<2, 2, 1, 1>(k, stream, vec, vec_out, splits); break;
            case 4: rc = launch_conv<4, 1, 1, 1>(k, stream, vec, vec_out, splits); break;
            default: return GA_E_UNSUPPORTED;
        }
    }
    if (rc != GA_OK || splits == 1) return rc;
    long items = M * d.Cout / (vec_out ? 4 : 1);
    long blocks = (items + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(conv_splitk_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, k, (int)M, splits, vec_out);
    return check_launch();
}
