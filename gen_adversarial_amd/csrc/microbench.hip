// On-box peak microbenchmarks ("confirm with a bandwidth + MFMA microbenchmark on the box and normalise by
// measured peaks too").  bench.py times them with HIP events and reports roofline.frac_of_measured_peak beside the fraction of
// the data-sheet peaks of /opt/skills/guides/MI355X_MICROARCH.md.  Not part of the purification path.
#include "ga_common.h"

namespace ga {

typedef __bf16 mb_bf16x8 __attribute__((ext_vector_type(8)));

// float4 grid-stride copy: 16 B per lane, the widest coalesced access (the guide measures 6.29 TB/s = 79 % of 8 TB/s this way)
__global__ void __launch_bounds__(256) mb_copy_kernel(const floatx4* __restrict__ src, floatx4* __restrict__ dst, const long n4) {
    const long stride = (long)gridDim.x * 256;
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * stride < n4; i += 4 * stride) {          // four 16-B loads in flight per lane before the first store
        const floatx4 a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
        dst[i] = a; dst[i + stride] = b; dst[i + 2 * stride] = c; dst[i + 3 * stride] = d;
    }
    for (; i < n4; i += stride) dst[i] = src[i];
}

// bare v_mfma_f32_32x32x16_bf16 loop: one wave per SIMD (4 waves per workgroup, 4 workgroups per CU resident), 4 independent
// accumulators, operands in registers, pseudo-random full-range values (zero-filled operands read ~20 % high: the chip clocks up)
__global__ void __launch_bounds__(256) mb_mfma_kernel(float* __restrict__ out, const int iters, const unsigned seed) {
    mb_bf16x8 a[2], b[2];
    unsigned s = seed ^ (blockIdx.x * 2654435761u) ^ (threadIdx.x * 40503u);
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            s = s * 1664525u + 1013904223u;
            a[k][e] = (__bf16)(((int)(s >> 8) % 2001 - 1000) * 1e-3f);
            s = s * 1664525u + 1013904223u;
            b[k][e] = (__bf16)(((int)(s >> 8) % 2001 - 1000) * 1e-3f);
        }
    floatx16 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[j & 1], acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[(j + 1) & 1], acc[j], 0, 0, 0);
        }
    }
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) sum += acc[j][r];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = sum;
}

// the same loop on v_mfma_f32_16x16x32_bf16 (same flops per wave and per iteration: 16 MFMAs of half the size on 8 accumulators of
// 4 registers).  MI355X_MICROARCH.md, DVFS give-back item 7: under the chip's power limit the clock it holds depends on the MFMA
// shape; this pair of loops measures that on the box (roofline.measured_peak.mfma_shapes).
__global__ void __launch_bounds__(256) mb_mfma16_kernel(float* __restrict__ out, const int iters, const unsigned seed) {
    mb_bf16x8 a[2], b[2];
    unsigned s = seed ^ (blockIdx.x * 2654435761u) ^ (threadIdx.x * 40503u);
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            s = s * 1664525u + 1013904223u;
            a[k][e] = (__bf16)(((int)(s >> 8) % 2001 - 1000) * 1e-3f);
            s = s * 1664525u + 1013904223u;
            b[k][e] = (__bf16)(((int)(s >> 8) % 2001 - 1000) * 1e-3f);
        }
    floatx4 acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[j][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[j & 1], acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[(j + 1) & 1], acc[j], 0, 0, 0);
        }
    }
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) sum += acc[j][r];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = sum;
}

}  // namespace ga

using namespace ga;

extern "C" int ga_microbench_hbm_copy(const float* src, float* dst, long n_floats, void* s) {
    ga::clear_stale_error();
    if (!src || !dst || n_floats <= 0 || (n_floats & 3)) return GA_E_BADARG;
    if (!aligned16(src) || !aligned16(dst)) return GA_E_ALIGN;
    hipLaunchKernelGGL(mb_copy_kernel, dim3(256 * 8), dim3(256), 0, (hipStream_t)s, reinterpret_cast<const floatx4*>(src),
                       reinterpret_cast<floatx4*>(dst), n_floats / 4);
    return check_launch();
}

// out: >= blocks * 256 floats.  Algorithmic flops of one call = blocks * 4 waves * iters * 8 MFMAs * 2 * 32 * 32 * 16.
extern "C" int ga_microbench_mfma_bf16(float* out, int blocks, int iters, void* s) {
    ga::clear_stale_error();
    if (!out || blocks <= 0 || iters <= 0) return GA_E_BADARG;
    hipLaunchKernelGGL(mb_mfma_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)s, out, iters, 0x9e3779b9u);
    return check_launch();
}

// shape 32: the loop above; shape 16: v_mfma_f32_16x16x32_bf16, the same algorithmic flops per call.
extern "C" int ga_microbench_mfma_bf16_shape(float* out, int blocks, int iters, int shape, void* s) {
    ga::clear_stale_error();
    if (!out || blocks <= 0 || iters <= 0 || (shape != 16 && shape != 32)) return GA_E_BADARG;
    if (shape == 32) hipLaunchKernelGGL(mb_mfma_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)s, out, iters, 0x9e3779b9u);
    else hipLaunchKernelGGL(mb_mfma16_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)s, out, iters, 0x9e3779b9u);
    return check_launch();
}
