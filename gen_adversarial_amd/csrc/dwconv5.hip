// ga_dwconv5 — depthwise 5x5 (pad 2) with fused activation prologue / act' epilogue, NHWC, HBM-bound.
//
// One block = NB images x (TH x TW) output window x 32 channels.  The (TH+4)x(TW+4) halo window is staged in LDS
// once, with the activation (SiLU) applied once per element (not once per tap), zero-filled outside the image.
// Threads are laid out as 8 channel-quads (float4, 16 B/lane => a pixel's 32 channels are one 128-B line) x 32
// strip lanes.  Staging issues its global loads in groups of 4 before touching them (memory-level parallelism at
// ~80 VGPRs instead of one dependent round trip per element); the taps live in LDS ([25][32]) and are read 5 at a
// time per kernel row; each thread produces a strip of SW=4 horizontally adjacent outputs, so a kernel row costs
// SW+4 input reads + 5 weight reads for 5*SW fused multiply-adds (16 LDS reads per output instead of 25).
//   up2   : the input is at half resolution and read through nearest-neighbour x2 (forward of an upsampling cell)
//   pool2 : the output is the 2x2 sum of the window results (the adjoint of up2, backward of an upsampling cell)
#include "ga_common.h"

namespace ga {

constexpr int DW_CC = 32;   // channels per block

template <int SW>
__global__ void __launch_bounds__(256, 4)     // 4 workgroups per CU: the kernel is latency/HBM-bound, occupancy matters
dwconv5_kernel(const ga_dwconv5_desc d, const int NB, const int TH, const int TW, const int tilesH, const int tilesW,
               const int nchunks) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, c4 = tid & 7, pl = tid >> 3;

    int b = blockIdx.x;
    const int chunk = b % nchunks; b /= nchunks;
    const int tw = b % tilesW; b /= tilesW;
    const int th = b % tilesH; b /= tilesH;
    const int n0 = b * NB;
    const int nb = min(NB, d.N - n0);
    const int h0 = th * TH, w0 = tw * TW;
    const int c = chunk * DW_CC + 4 * c4;
    const bool cok = c < d.C;

    const int HH = TH + 4, WW = TW + 4;
    const int Hs = d.up2 ? d.H / 2 : d.H, Ws = d.up2 ? d.W / 2 : d.W;
    float* wS = smem;                         // [25][32]
    float* tile = smem + 25 * DW_CC;          // [nb][HH][WW][32]

    const floatx4 zero = {0.f, 0.f, 0.f, 0.f};
    if (tid < 200) {                          // 25 taps x 8 channel-quads
        const int t = tid >> 3;
        *reinterpret_cast<floatx4*>(wS + t * DW_CC + 4 * c4) =
            cok ? *reinterpret_cast<const floatx4*>(d.w + (size_t)t * d.C + c) : zero;
    }
    floatx4 bias = zero;
    if (cok && d.bias) bias = *reinterpret_cast<const floatx4*>(d.bias + c);

    // ---- stage halo window: loads in groups of 4, then activation + LDS write
    const int halo_px = nb * HH * WW;
    for (int p0 = pl; p0 < halo_px; p0 += 32 * 4) {
        floatx4 v[4];
        bool ok[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int p = p0 + 32 * k;
            const int ww = p % WW; const int q = p / WW;
            const int hh = q % HH; const int ni = q / HH;
            const int h = h0 + hh - 2, w = w0 + ww - 2;
            ok[k] = cok && p < halo_px && h >= 0 && h < d.H && w >= 0 && w < d.W;
            const int hs = d.up2 ? (h >> 1) : h, ws = d.up2 ? (w >> 1) : w;
            const size_t off = ok[k] ? (((size_t)(n0 + ni) * Hs + hs) * Ws + ws) * d.C + c : 0;
            v[k] = *reinterpret_cast<const floatx4*>(d.x + off);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int p = p0 + 32 * k;
            if (p < halo_px) {
                floatx4 o = v[k];
                if (d.pro_act == GA_ACT_SILU) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = o[e] * fast_sigmoid(o[e]);
                } else if (d.pro_act) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = act_fwd_fast(o[e], d.pro_act);
                }
                *reinterpret_cast<floatx4*>(tile + (size_t)p * DW_CC + 4 * c4) = ok[k] ? o : zero;
            }
        }
    }
    __syncthreads();
    if (!cok) return;

    // SW adjacent window results of output row hh starting at column ws (tile coordinates)
    auto strip = [&](const int ni, const int hh, const int ws, floatx4 (&acc)[SW]) {
#pragma unroll
        for (int j = 0; j < SW; ++j) acc[j] = zero;
        const float* base = tile + ((size_t)(ni * HH + hh) * WW + ws) * DW_CC + 4 * c4;
#pragma unroll 1
        for (int kh = 0; kh < 5; ++kh) {
            floatx4 w5[5], in[SW + 4];
#pragma unroll
            for (int kw = 0; kw < 5; ++kw) w5[kw] = *reinterpret_cast<const floatx4*>(wS + (kh * 5 + kw) * DW_CC + 4 * c4);
#pragma unroll
            for (int j = 0; j < SW + 4; ++j) in[j] = *reinterpret_cast<const floatx4*>(base + (kh * WW + j) * DW_CC);
#pragma unroll
            for (int j = 0; j < SW; ++j)
#pragma unroll
                for (int kw = 0; kw < 5; ++kw) acc[j] += in[j + kw] * w5[kw];
        }
    };

    // output row n, offset po inside the row; the act' source is the forward's tensor: row n / act_rep with K cotangents per row
    const size_t row_out = (size_t)(d.pool2 ? (d.H / 2) * (d.W / 2) : d.H * d.W) * d.C;
    const int arep = d.act_rep > 1 ? d.act_rep : 1;
    auto finish = [&](floatx4 v, const int n, const size_t po) __attribute__((always_inline)) {
        const size_t o = (size_t)n * row_out + po;
        if (d.dact_x) {
            const floatx4 u = *reinterpret_cast<const floatx4*>(d.dact_x + (size_t)(n / arep) * row_out + po);
            if (d.dact_act == GA_ACT_SILU) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float sg = fast_sigmoid(u[e]); v[e] *= sg * (1.0f + u[e] * (1.0f - sg)); }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] *= act_bwd_fast(u[e], d.dact_act);
            }
        }
        *reinterpret_cast<floatx4*>(d.y + o) = v;
    };

    const int th_n = min(TH, d.H - h0), tw_n = min(TW, d.W - w0);
    if (!d.pool2) {
        const int spr = (tw_n + SW - 1) / SW;                            // strips per row
        const int nst = nb * th_n * spr;
        for (int s = pl; s < nst; s += 32) {
            const int sx = s % spr; const int q = s / spr;
            const int hh = q % th_n; const int ni = q / th_n;
            floatx4 acc[SW];
            strip(ni, hh, sx * SW, acc);
#pragma unroll
            for (int j = 0; j < SW; ++j) {
                const int ww = sx * SW + j;
                if (ww < tw_n)
                    finish(acc[j] + bias, n0 + ni, ((size_t)(h0 + hh) * d.W + (w0 + ww)) * d.C + c);
            }
        }
    } else {
        // outputs at half resolution: SW must be even; a strip pair (rows 2r, 2r+1) yields SW/2 outputs
        const int Wo = d.W / 2;
        const int spr = (tw_n + SW - 1) / SW;
        const int nst = nb * (th_n / 2) * spr;
        for (int s = pl; s < nst; s += 32) {
            const int sx = s % spr; const int q = s / spr;
            const int hr = q % (th_n / 2); const int ni = q / (th_n / 2);
            floatx4 a0[SW], a1[SW];
            strip(ni, 2 * hr, sx * SW, a0);
            strip(ni, 2 * hr + 1, sx * SW, a1);
#pragma unroll
            for (int j = 0; j < SW; j += 2) {
                const int ww = sx * SW + j;
                if (ww < tw_n) {
                    const floatx4 v = a0[j] + a0[j + (SW > 1 ? 1 : 0)] + a1[j] + a1[j + (SW > 1 ? 1 : 0)];
                    finish(v + bias, n0 + ni, ((size_t)(h0 / 2 + hr) * Wo + (w0 + ww) / 2) * d.C + c);
                }
            }
        }
    }
}

// 4 x 4 images (the deepest decoder cells: 512 rows x 3072 channels = 12288 workgroups of 16 KB each in the windowed form, 88 us
// of mostly workgroup turnover): no halo frame — 16 images x 16 pixels x 32 channels in 32 KB of LDS, the border handled by
// the loop bounds (a 4 x 4 image meets 9 - 16 of the 25 taps per output, the others are skipped, not multiplied by zero).
// Work item = (image, output row, channel quad): four outputs from at most four input rows.
__global__ void __launch_bounds__(256, 4) dwconv5_4x4_kernel(const ga_dwconv5_desc d, const int nchunks) {
    constexpr int NB = 16;
    __shared__ __attribute__((aligned(16))) float wS[25 * DW_CC];
    __shared__ __attribute__((aligned(16))) float tile[NB * 16 * DW_CC];
    const int tid = threadIdx.x, c4 = tid & 7, pl = tid >> 3;
    const int chunk = blockIdx.x % nchunks;
    const int n0 = (blockIdx.x / nchunks) * NB;
    const int nb = min(NB, d.N - n0);
    const int c = chunk * DW_CC + 4 * c4;
    const bool cok = c < d.C;
    const floatx4 zero = {0.f, 0.f, 0.f, 0.f};
    if (tid < 200) {
        const int t = tid >> 3;
        *reinterpret_cast<floatx4*>(wS + t * DW_CC + 4 * c4) = cok ? *reinterpret_cast<const floatx4*>(d.w + (size_t)t * d.C + c) : zero;
    }
    floatx4 bias = zero;
    if (cok && d.bias) bias = *reinterpret_cast<const floatx4*>(d.bias + c);
    // ---- stage nb * 16 pixels: 8 loads per thread, all issued before the first use
    {
        floatx4 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int p = pl + 32 * k;
            const bool ok = cok && p < nb * 16;
            v[k] = ok ? *reinterpret_cast<const floatx4*>(d.x + ((size_t)n0 * 16 + p) * d.C + c) : zero;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int p = pl + 32 * k;
            floatx4 o = v[k];
            if (d.pro_act == GA_ACT_SILU) {
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = o[e] * fast_sigmoid(o[e]);
            } else if (d.pro_act) {
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = act_fwd_fast(o[e], d.pro_act);
            }
            *reinterpret_cast<floatx4*>(tile + p * DW_CC + 4 * c4) = o;
        }
    }
    __syncthreads();
    if (!cok) return;
    for (int s = pl; s < nb * 4; s += 32) {
        const int ni = s >> 2, h = s & 3;
        floatx4 acc[4] = {bias, bias, bias, bias};
        const int kh0 = max(0, 2 - h), kh1 = min(5, 6 - h);          // input row h + kh - 2 inside [0, 4)
        for (int kh = kh0; kh < kh1; ++kh) {
            const float* row = tile + ((ni * 4 + h + kh - 2) * 4) * DW_CC + 4 * c4;
            floatx4 in[4], w5[5];
#pragma unroll
            for (int j = 0; j < 4; ++j) in[j] = *reinterpret_cast<const floatx4*>(row + j * DW_CC);
#pragma unroll
            for (int kw = 0; kw < 5; ++kw) w5[kw] = *reinterpret_cast<const floatx4*>(wS + (kh * 5 + kw) * DW_CC + 4 * c4);
#pragma unroll
            for (int w = 0; w < 4; ++w)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int kw = j - w + 2;                       // compile-time after unrolling
                    if (kw >= 0 && kw < 5) acc[w] += in[j] * w5[kw];
                }
        }
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const size_t o = (((size_t)(n0 + ni) * 4 + h) * 4 + w) * d.C + c;
            floatx4 v = acc[w];
            if (d.dact_x) {
                const int na = d.act_rep > 1 ? (n0 + ni) / d.act_rep : n0 + ni;       // K cotangents per forward row
                const floatx4 u = *reinterpret_cast<const floatx4*>(d.dact_x + (((size_t)na * 4 + h) * 4 + w) * d.C + c);
                if (d.dact_act == GA_ACT_SILU) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { const float sg = fast_sigmoid(u[e]); v[e] *= sg * (1.0f + u[e] * (1.0f - sg)); }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] *= act_bwd_fast(u[e], d.dact_act);
                }
            }
            *reinterpret_cast<floatx4*>(d.y + o) = v;
        }
    }
}

}  // namespace ga

extern "C" int ga_dwconv5(const ga_dwconv5_desc* dp, void* stream_) {
    ga::clear_stale_error();
    using namespace ga;
    if (!dp) return GA_E_BADARG;
    const ga_dwconv5_desc& d = *dp;
    if (!d.x || !d.w || !d.y || d.N <= 0 || d.H <= 0 || d.W <= 0 || d.C <= 0) return GA_E_BADARG;
    if (d.C % 4) return GA_E_UNSUPPORTED;
    if ((d.up2 || d.pool2) && ((d.H | d.W) & 1)) return GA_E_BADARG;
    if (d.up2 && d.pool2) return GA_E_UNSUPPORTED;
    if (d.act_rep > 1 && d.N % d.act_rep) return GA_E_BADARG;
    if (!aligned16(d.x) || !aligned16(d.w) || !aligned16(d.y) || (d.bias && !aligned16(d.bias)) ||
        (d.dact_x && !aligned16(d.dact_x))) return GA_E_ALIGN;
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);

    if (d.H == 4 && d.W == 4 && !d.up2 && !d.pool2) {
        const int nchunks = (d.C + DW_CC - 1) / DW_CC;
        const long blocks = (long)((d.N + 15) / 16) * nchunks;
        if (blocks > 0x7fffffffL) return GA_E_UNSUPPORTED;
        hipLaunchKernelGGL(dwconv5_4x4_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, d, nchunks);
        return check_launch();
    }
    // 8 x 16 output windows: (12 x 20) x 32 ch halo = 30 KB of LDS -> 4 workgroups per CU
    const int TH = d.H < 8 ? d.H : 8, TW = d.W < 16 ? d.W : 16;
    const int halo_bytes = (TH + 4) * (TW + 4) * DW_CC * 4;
    int NB = 256 / (TH * TW);
    if (NB < 1) NB = 1;
    const int fit = (40 * 1024 - 25 * DW_CC * 4) / halo_bytes;        // 4 workgroups x 40 KB = the CU's 160 KB
    if (NB > fit) NB = fit;
    if (NB > d.N) NB = d.N;
    if (NB < 1) NB = 1;
    const int tilesH = (d.H + TH - 1) / TH, tilesW = (d.W + TW - 1) / TW;
    const int nchunks = (d.C + DW_CC - 1) / DW_CC;
    const long blocks = (long)((d.N + NB - 1) / NB) * tilesH * tilesW * nchunks;
    if (blocks > 0x7fffffffL) return GA_E_UNSUPPORTED;
    const size_t lds = (size_t)NB * halo_bytes + 25 * DW_CC * 4;
    // strips of 4 when every tile width is a multiple of 4 (the strip reads SW+4 columns: stays inside the halo),
    // else strips of 2 (pool2 needs an even strip), else single outputs
    // (small images: strips of 2 when strips of 4 would leave half of the 32 strip lanes without work)
    const bool w2 = (d.W % 2 == 0) && (TW % 2 == 0);
    const bool w4 = (d.W % 4 == 0) && (TW % 4 == 0) && !(w2 && NB * TH * (TW / 4) < 32);
    if (w4) hipLaunchKernelGGL(dwconv5_kernel<4>, dim3((unsigned)blocks), dim3(256), lds, stream, d, NB, TH, TW, tilesH, tilesW, nchunks);
    else if (w2) hipLaunchKernelGGL(dwconv5_kernel<2>, dim3((unsigned)blocks), dim3(256), lds, stream, d, NB, TH, TW, tilesH, tilesW, nchunks);
    else {
        if (d.pool2) return GA_E_UNSUPPORTED;
        hipLaunchKernelGGL(dwconv5_kernel<1>, dim3((unsigned)blocks), dim3(256), lds, stream, d, NB, TH, TW, tilesH, tilesW, nchunks);
    }
    return check_launch();
}
