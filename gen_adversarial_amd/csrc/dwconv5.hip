// ga_dwconv5 — depthwise 5x5 (pad 2) with fused activation prologue / act' epilogue, NHWC, HBM-bound.
//
// One block = NB images x (TH x TW) output window x 32 channels.  The (TH+4)x(TW+4) halo window is staged in LDS
// once, with the activation (SiLU) applied once per element (not once per tap), zero-filled outside the image.
// Threads are laid out as 8 channel-quads (float4, 16 B/lane => a pixel's 32 channels are one 128-B line) x 32
// pixel lanes; the 25 per-channel taps live in registers for the block's lifetime.
//   up2   : the input is at half resolution and read through nearest-neighbour x2 (forward of an upsampling cell)
//   pool2 : the output is the 2x2 sum of the window results (the adjoint of up2, backward of an upsampling cell)
#include "ga_common.h"

namespace ga {

constexpr int DW_CC = 32;   // channels per block

__global__ void __launch_bounds__(256)
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

    // taps -> registers
    floatx4 wt[25];
#pragma unroll
    for (int t = 0; t < 25; ++t) {
        wt[t] = cok ? *reinterpret_cast<const floatx4*>(d.w + (size_t)t * d.C + c) : floatx4{0.f, 0.f, 0.f, 0.f};
    }
    floatx4 bias = {0.f, 0.f, 0.f, 0.f};
    if (cok && d.bias) bias = *reinterpret_cast<const floatx4*>(d.bias + c);

    // ---- stage halo window (activation applied once)
    const int halo_px = nb * HH * WW;
    for (int p = pl; p < halo_px; p += 32) {
        const int ww = p % WW; int q = p / WW;
        const int hh = q % HH; const int ni = q / HH;
        const int h = h0 + hh - 2, w = w0 + ww - 2;
        floatx4 v = {0.f, 0.f, 0.f, 0.f};
        if (cok && h >= 0 && h < d.H && w >= 0 && w < d.W) {
            const int hs = d.up2 ? (h >> 1) : h, ws = d.up2 ? (w >> 1) : w;
            v = *reinterpret_cast<const floatx4*>(d.x + (((size_t)(n0 + ni) * Hs + hs) * Ws + ws) * d.C + c);
            if (d.pro_act) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = act_fwd(v[e], d.pro_act);
            }
        }
        *reinterpret_cast<floatx4*>(smem + (size_t)p * DW_CC + 4 * c4) = v;
    }
    __syncthreads();
    if (!cok) return;

    auto window = [&](const int ni, const int hh, const int ww) -> floatx4 {   // hh,ww: output coords inside the tile
        floatx4 acc = {0.f, 0.f, 0.f, 0.f};
        const float* base = smem + ((size_t)(ni * HH + hh) * WW + ww) * DW_CC + 4 * c4;
#pragma unroll
        for (int kh = 0; kh < 5; ++kh)
#pragma unroll
            for (int kw = 0; kw < 5; ++kw)
                acc += *reinterpret_cast<const floatx4*>(base + (kh * WW + kw) * DW_CC) * wt[kh * 5 + kw];
        return acc;
    };

    if (!d.pool2) {
        const int th_n = min(TH, d.H - h0), tw_n = min(TW, d.W - w0);
        const int npx = nb * th_n * tw_n;
        for (int p = pl; p < npx; p += 32) {
            const int ww = p % tw_n; int q = p / tw_n;
            const int hh = q % th_n; const int ni = q / th_n;
            floatx4 v = window(ni, hh, ww) + bias;
            const size_t o = (((size_t)(n0 + ni) * d.H + (h0 + hh)) * d.W + (w0 + ww)) * d.C + c;
            if (d.dact_x) {
                const floatx4 u = *reinterpret_cast<const floatx4*>(d.dact_x + o);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] *= act_bwd(u[e], d.dact_act);
            }
            *reinterpret_cast<floatx4*>(d.y + o) = v;
        }
    } else {
        const int Ho = d.H / 2, Wo = d.W / 2;
        const int th_n = min(TH, d.H - h0) / 2, tw_n = min(TW, d.W - w0) / 2;
        const int npx = nb * th_n * tw_n;
        for (int p = pl; p < npx; p += 32) {
            const int ww = p % tw_n; int q = p / tw_n;
            const int hh = q % th_n; const int ni = q / th_n;
            floatx4 v = window(ni, 2 * hh, 2 * ww) + window(ni, 2 * hh, 2 * ww + 1) +
                        window(ni, 2 * hh + 1, 2 * ww) + window(ni, 2 * hh + 1, 2 * ww + 1);
            const size_t o = (((size_t)(n0 + ni) * Ho + (h0 / 2 + hh)) * Wo + (w0 / 2 + ww)) * d.C + c;
            if (d.dact_x) {
                const floatx4 u = *reinterpret_cast<const floatx4*>(d.dact_x + o);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] *= act_bwd(u[e], d.dact_act);
            }
            *reinterpret_cast<floatx4*>(d.y + o) = v;
        }
    }
}

}  // namespace ga

extern "C" int ga_dwconv5(const ga_dwconv5_desc* dp, void* stream_) {
    using namespace ga;
    if (!dp) return GA_E_BADARG;
    const ga_dwconv5_desc& d = *dp;
    if (!d.x || !d.w || !d.y || d.N <= 0 || d.H <= 0 || d.W <= 0 || d.C <= 0) return GA_E_BADARG;
    if (d.C % 4) return GA_E_UNSUPPORTED;
    if ((d.up2 || d.pool2) && ((d.H | d.W) & 1)) return GA_E_BADARG;
    if (d.up2 && d.pool2) return GA_E_UNSUPPORTED;
    if (!aligned16(d.x) || !aligned16(d.w) || !aligned16(d.y) || (d.bias && !aligned16(d.bias)) ||
        (d.dact_x && !aligned16(d.dact_x))) return GA_E_ALIGN;
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);

    const int TH = d.H < 16 ? d.H : 16, TW = d.W < 16 ? d.W : 16;
    const int halo_bytes = (TH + 4) * (TW + 4) * DW_CC * 4;
    int NB = 256 / (TH * TW);
    if (NB < 1) NB = 1;
    const int fit = 65536 / halo_bytes;
    if (NB > fit) NB = fit;
    if (NB > d.N) NB = d.N;
    if (NB < 1) NB = 1;
    const int tilesH = (d.H + TH - 1) / TH, tilesW = (d.W + TW - 1) / TW;
    const int nchunks = (d.C + DW_CC - 1) / DW_CC;
    const long blocks = (long)((d.N + NB - 1) / NB) * tilesH * tilesW * nchunks;
    if (blocks > 0x7fffffffL) return GA_E_UNSUPPORTED;
    const size_t lds = (size_t)NB * halo_bytes;
    hipLaunchKernelGGL(dwconv5_kernel, dim3((unsigned)blocks), dim3(256), lds, stream, d, NB, TH, TW, tilesH, tilesW, nchunks);
    return check_launch();
}
