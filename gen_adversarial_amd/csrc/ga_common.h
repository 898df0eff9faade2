// Shared device helpers for libga_ops (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include "ga_ops.h"

#define GA_MAX_DEVICES 64

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

namespace ga {

extern thread_local hipError_t g_last_err;

__device__ __forceinline__ float sigmoidf_(float u) { return 1.0f / (1.0f + expf(-u)); }

// activation applied as a prologue (reference: torch.nn.SiLU / ELU / ReLU in NVAE cells and VGG)
__device__ __forceinline__ float act_fwd(float u, int act) {
    switch (act) {
        case GA_ACT_SILU: return u * sigmoidf_(u);
        case GA_ACT_ELU:  return u > 0.0f ? u : expm1f(u);
        case GA_ACT_RELU: return fmaxf(u, 0.0f);
        case GA_ACT_LRELU: return u > 0.0f ? u : 0.01f * u;
        case GA_ACT_FLRELU: return (u > 0.0f ? u : 0.2f * u) * 1.41421356237309515f;
        default:          return u;
    }
}

// d act(u) / du
__device__ __forceinline__ float act_bwd(float u, int act) {
    switch (act) {
        case GA_ACT_SILU: { float s = sigmoidf_(u); return s * (1.0f + u * (1.0f - s)); }
        case GA_ACT_ELU:  return u > 0.0f ? 1.0f : expf(u);
        case GA_ACT_RELU: return u > 0.0f ? 1.0f : 0.0f;
        case GA_ACT_LRELU: return u > 0.0f ? 1.0f : 0.01f;
        case GA_ACT_FLRELU: return (u > 0.0f ? 1.0f : 0.2f) * 1.41421356237309515f;
        default:          return 1.0f;
    }
}

// ---- fast variants for the hot kernels: v_exp_f32 / v_rcp_f32 (≈1 ulp each) instead of the libm-accurate expf and
//      IEEE division.  Relative error ~1e-6 at |u| ~ 10, far inside the 1e-3 parity bar (tests run at 2e-4).
__device__ __forceinline__ float fast_sigmoid(float u) { return __builtin_amdgcn_rcpf(1.0f + __expf(-u)); }

__device__ __forceinline__ float act_fwd_fast(float u, int act) {
    switch (act) {
        case GA_ACT_SILU: return u * fast_sigmoid(u);
        case GA_ACT_ELU:  return u > 0.0f ? u : expm1f(u);          // rare on the path: keep libm accuracy near 0
        case GA_ACT_RELU: return fmaxf(u, 0.0f);
        case GA_ACT_LRELU: return u > 0.0f ? u : 0.01f * u;
        default:          return u;
    }
}

__device__ __forceinline__ float act_bwd_fast(float u, int act) {
    switch (act) {
        case GA_ACT_SILU: { float s = fast_sigmoid(u); return s * (1.0f + u * (1.0f - s)); }
        case GA_ACT_ELU:  return u > 0.0f ? 1.0f : expf(u);
        case GA_ACT_RELU: return u > 0.0f ? 1.0f : 0.0f;
        case GA_ACT_LRELU: return u > 0.0f ? 1.0f : 0.01f;
        default:          return 1.0f;
    }
}

// ---- division of row indices by image sizes in the conv kernels' setup: n / d for 0 <= n < 2^31 as one 32x32->64 multiply
//      and a shift (m = ceil(2^(31+s) / d), s = ceil(log2 d): exact because the rounding excess e < d <= 2^s gives
//      n * e < 2^(31+s)).  The host makes the constants, the kernels take them as arguments.
struct fastdiv { unsigned m; int s; };
inline fastdiv make_fastdiv(int d) {
    int s = 0;
    while ((1LL << s) < d) ++s;
    const unsigned long long p = 1ULL << (31 + s);
    return fastdiv{(unsigned)((p + (unsigned long long)d - 1) / (unsigned long long)d), 31 + s};
}
__device__ __forceinline__ int fd_div(int n, const fastdiv f) { return (int)(((unsigned long long)(unsigned)n * f.m) >> f.s); }

// bit (kh*KW + kw) set when tap (kh, kw) of the window whose top-left input pixel is (h0, w0) lies inside the Hi x Wi image
__device__ __forceinline__ unsigned tap_mask(int h0, int w0, int Hi, int Wi, int KH, int KW) {
    const int wl = max(0, -w0), wh = min(KW, Wi - w0);
    const unsigned cols = wh > wl ? ((1u << (wh - wl)) - 1u) << wl : 0u;
    const int hl = max(0, -h0), hh = min(KH, Hi - h0);
    unsigned mk = 0;
    for (int kh = hl; kh < hh; ++kh) mk |= cols << (kh * KW);
    return mk;
}

// a sticky error left by an unrelated earlier HIP call must not be blamed on our launch: entry points clear it first
inline void clear_stale_error() { (void)hipGetLastError(); }

inline int check_launch() {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { g_last_err = e; return GA_E_LAUNCH; }
    return GA_OK;
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// hipFuncAttributeMaxDynamicSharedMemorySize is a property of (kernel, DEVICE): one cache slot per device, one cache object per
// kernel instantiation (a function-local static of the launcher).  Relaxed atomics: two host threads racing here both set the
// attribute, which is idempotent.  Returns false when the runtime refuses (the launch would fail with an LDS size error).
struct dyn_lds_cache { std::atomic<size_t> have[GA_MAX_DEVICES]; };
inline bool ensure_dyn_lds(dyn_lds_cache& c, const void* fn, size_t lds) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0) return false;
    std::atomic<size_t>* slot = dev < GA_MAX_DEVICES ? &c.have[dev] : nullptr;
    if (slot && slot->load(std::memory_order_relaxed) >= lds) return true;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return false;
    if (slot) slot->store(lds, std::memory_order_relaxed);
    return true;
}

}  // namespace ga
