// Element-wise / reduction pieces of the A-VAE competitor purifier (src/defenses/competitors/a_vae: model.py:9-141,
// modules.py:98-104, 278-381, 384-416; purification_model.py:16-20), one entry point with a mode:
//   GA_AVAE_ADAIN      u = lrelu_0.2(t + wn[c] * noise[n,p]);  y = gamma[n,c] * (u - mean_p u) * rstd + beta[n,c]
//                      (NoiseInjection -> LeakyReLU -> AdaptiveInstanceNorm of StyledConvBlock.forward, modules.py:367-381;
//                      InstanceNorm2d: biased variance over the pixels, eps 1e-5, no affine)
//   GA_AVAE_AVGPOOL    k x k / stride k mean of an NHWC tensor (avg_pool2d in AVaeDefenseModel.purify)
//   GA_AVAE_PIXELNORM  y = x * rsqrt(mean_c x^2 + 1e-8) with its backward (the style MLP's first layer sees the sampled latent,
//                      which depends on the input image: unlike ga_pixelnorm it needs the adjoint)
//   GA_AVAE_SAMPLE     m, v = halves of lrelu_0.2(t);  z = m + eps * exp(0.5 v) * temp   (Generator.forward, model.py:80-91)
// HBM-bound vector code; every reduction has a fixed order (bitwise reproducible).
#include "ga_common.h"

namespace ga {

__device__ __forceinline__ float lrelu02(const float v) { return v > 0.f ? v : 0.2f * v; }

// one workgroup per (row, 32-channel chunk): 8 channel quads x 32 pixel lanes; two passes over the row's pixels
__global__ void __launch_bounds__(256) avae_adain_kernel(const ga_avae_desc d, const int nchunks) {
    __shared__ __attribute__((aligned(16))) float red[3][32][32];
    __shared__ float sm_mean[32], sm_rstd[32], sm_a[32], sm_b[32];
    const int n = blockIdx.x / nchunks, chunk = blockIdx.x % nchunks;
    const int tid = threadIdx.x, c4 = tid & 7, pl = tid >> 3;
    const int c = chunk * 32 + 4 * c4;
    const bool cok = c < d.C;
    const floatx4 zero = {0.f, 0.f, 0.f, 0.f};
    const float* t = d.x + (size_t)n * d.P * d.C;
    const float* nz = d.a ? d.a + (size_t)n * d.P : nullptr;
    const floatx4 wn = (cok && d.b) ? *reinterpret_cast<const floatx4*>(d.b + c) : zero;
    if (!d.backward) {
        floatx4 s1 = zero, s2 = zero;
        if (cok)
            for (int p = pl; p < d.P; p += 32) {
                floatx4 u = *reinterpret_cast<const floatx4*>(t + (size_t)p * d.C + c);
                if (nz) u += wn * nz[p];
#pragma unroll
                for (int e = 0; e < 4; ++e) u[e] = lrelu02(u[e]);
                s1 += u; s2 += u * u;
            }
#pragma unroll
        for (int e = 0; e < 4; ++e) { red[0][pl][4 * c4 + e] = s1[e]; red[1][pl][4 * c4 + e] = s2[e]; }
        __syncthreads();
        if (tid < 32) {
            float a = 0.f, b = 0.f;
            for (int k = 0; k < 32; ++k) { a += red[0][k][tid]; b += red[1][k][tid]; }
            const float mean = a / (float)d.P;
            const float var = fmaxf(b / (float)d.P - mean * mean, 0.f);
            const float rstd = rsqrtf(var + 1e-5f);
            sm_mean[tid] = mean; sm_rstd[tid] = rstd;
            const int cc = chunk * 32 + tid;
            if (cc < d.C) { d.y2[((size_t)n * d.C + cc) * 2] = mean; d.y2[((size_t)n * d.C + cc) * 2 + 1] = rstd; }
        }
        __syncthreads();
        if (!cok) return;
        floatx4 mean, rstd;
#pragma unroll
        for (int e = 0; e < 4; ++e) { mean[e] = sm_mean[4 * c4 + e]; rstd[e] = sm_rstd[4 * c4 + e]; }
        const floatx4 gamma = *reinterpret_cast<const floatx4*>(d.c + (size_t)n * 2 * d.C + c) * rstd;
        const floatx4 beta = *reinterpret_cast<const floatx4*>(d.c + (size_t)n * 2 * d.C + d.C + c);
        float* y = d.y + (size_t)n * d.P * d.C;
        for (int p = pl; p < d.P; p += 32) {
            floatx4 u = *reinterpret_cast<const floatx4*>(t + (size_t)p * d.C + c);
            if (nz) u += wn * nz[p];
#pragma unroll
            for (int e = 0; e < 4; ++e) u[e] = lrelu02(u[e]);
            *reinterpret_cast<floatx4*>(y + (size_t)p * d.C + c) = (u - mean) * gamma + beta;
        }
    } else {
        // dy -> dt, d gamma, d beta.  xhat = (u - mean) rstd;  dgamma = sum dy xhat;  dbeta = sum dy;
        // du = gamma rstd (dy - dbeta / P - xhat dgamma / P);  dt = du * lrelu'(t + wn noise)
        floatx4 mean = zero, rstd = zero;
        if (cok)
#pragma unroll
            for (int e = 0; e < 4; ++e) { mean[e] = d.s[((size_t)n * d.C + c + e) * 2]; rstd[e] = d.s[((size_t)n * d.C + c + e) * 2 + 1]; }
        const float* dy = d.dy + (size_t)n * d.P * d.C;
        floatx4 s1 = zero, s2 = zero;
        if (cok)
            for (int p = pl; p < d.P; p += 32) {
                floatx4 u = *reinterpret_cast<const floatx4*>(t + (size_t)p * d.C + c);
                if (nz) u += wn * nz[p];
#pragma unroll
                for (int e = 0; e < 4; ++e) u[e] = lrelu02(u[e]);
                const floatx4 g = *reinterpret_cast<const floatx4*>(dy + (size_t)p * d.C + c);
                s1 += g; s2 += g * (u - mean) * rstd;
            }
#pragma unroll
        for (int e = 0; e < 4; ++e) { red[0][pl][4 * c4 + e] = s1[e]; red[1][pl][4 * c4 + e] = s2[e]; }
        __syncthreads();
        if (tid < 32) {
            float a = 0.f, b = 0.f;
            for (int k = 0; k < 32; ++k) { a += red[0][k][tid]; b += red[1][k][tid]; }
            sm_a[tid] = a; sm_b[tid] = b;
            const int cc = chunk * 32 + tid;
            if (cc < d.C) { d.y2[(size_t)n * 2 * d.C + cc] = b; d.y2[(size_t)n * 2 * d.C + d.C + cc] = a; }      // (d gamma | d beta)
        }
        __syncthreads();
        if (!cok) return;
        floatx4 dbeta, dgamma;
#pragma unroll
        for (int e = 0; e < 4; ++e) { dbeta[e] = sm_a[4 * c4 + e]; dgamma[e] = sm_b[4 * c4 + e]; }
        const float invP = 1.0f / (float)d.P;
        const floatx4 gr = *reinterpret_cast<const floatx4*>(d.c + (size_t)n * 2 * d.C + c) * rstd;
        float* dt = d.y + (size_t)n * d.P * d.C;
        for (int p = pl; p < d.P; p += 32) {
            floatx4 pre = *reinterpret_cast<const floatx4*>(t + (size_t)p * d.C + c);
            if (nz) pre += wn * nz[p];
            floatx4 u;
#pragma unroll
            for (int e = 0; e < 4; ++e) u[e] = lrelu02(pre[e]);
            const floatx4 g = *reinterpret_cast<const floatx4*>(dy + (size_t)p * d.C + c);
            floatx4 du = gr * (g - dbeta * invP - (u - mean) * rstd * (dgamma * invP));
#pragma unroll
            for (int e = 0; e < 4; ++e) du[e] *= pre[e] > 0.f ? 1.f : 0.2f;
            *reinterpret_cast<floatx4*>(dt + (size_t)p * d.C + c) = du;
        }
    }
}

__global__ void __launch_bounds__(256) avae_avgpool_kernel(const ga_avae_desc d, const long total4) {
    const int C4 = d.C / 4, k = d.k, Ho = d.H / k, Wo = d.W / k;
    const float inv = 1.0f / (float)(k * k);
    if (!d.backward) {
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long)gridDim.x * 256) {
            const int q = (int)(i % C4); long r = i / C4;
            const int wo = (int)(r % Wo); r /= Wo;
            const int ho = (int)(r % Ho); const long n = r / Ho;
            floatx4 acc = {0.f, 0.f, 0.f, 0.f};
            for (int a = 0; a < k; ++a)
                for (int b = 0; b < k; ++b)
                    acc += *reinterpret_cast<const floatx4*>(d.x + (((size_t)n * d.H + ho * k + a) * d.W + wo * k + b) * d.C + 4 * q);
            *reinterpret_cast<floatx4*>(d.y + i * 4) = acc * inv;
        }
    } else {
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long)gridDim.x * 256) {
            const int q = (int)(i % C4); long r = i / C4;
            const int w = (int)(r % d.W); r /= d.W;
            const int h = (int)(r % d.H); const long n = r / d.H;
            const floatx4 g = *reinterpret_cast<const floatx4*>(d.dy + (((size_t)n * Ho + h / k) * Wo + w / k) * d.C + 4 * q);
            *reinterpret_cast<floatx4*>(d.y + i * 4) = g * inv;
        }
    }
}

// one wavefront per row
__global__ void __launch_bounds__(256) avae_pixelnorm_kernel(const ga_avae_desc d) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= d.N) return;
    const float* x = d.x + row * d.C;
    float ss = 0.f, sd = 0.f;
    for (int c = lane; c < d.C; c += 64) {
        const float v = x[c];
        ss += v * v;
        if (d.backward) sd += v * d.dy[row * d.C + c];
    }
    for (int o = 32; o > 0; o >>= 1) { ss += __shfl_xor(ss, o); sd += __shfl_xor(sd, o); }
    const float r = rsqrtf(ss / (float)d.C + 1e-8f);
    if (!d.backward) {
        for (int c = lane; c < d.C; c += 64) d.y[row * d.C + c] = x[c] * r;
    } else {
        // y = x r, r = (mean x^2 + eps)^-1/2:  dx = r dy - x r^3 (sum_c dy x) / C
        const float k = r * r * r * sd / (float)d.C;
        for (int c = lane; c < d.C; c += 64) d.y[row * d.C + c] = r * d.dy[row * d.C + c] - x[c] * k;
    }
}

__global__ void __launch_bounds__(256) avae_sample_kernel(const ga_avae_desc d, const long total) {
    // t: [N, P, 2C] (NHWC), eps: [N, C, P] (NCHW, as the reference draws it), z / dz: [N, P, C]
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = (int)(i % d.C); long r = i / d.C;
        const int p = (int)(r % d.P); const long n = r / d.P;
        const float tm = d.x[(n * d.P + p) * 2 * d.C + c], tv = d.x[(n * d.P + p) * 2 * d.C + d.C + c];
        const float e = d.a[(n * d.C + c) * d.P + p];
        const float sg = expf(0.5f * lrelu02(tv)) * d.f0;
        if (!d.backward) {
            d.y[i] = lrelu02(tm) + e * sg;
        } else {
            const float dz = d.dy[i];
            d.y[(n * d.P + p) * 2 * d.C + c] = dz * (tm > 0.f ? 1.f : 0.2f);
            d.y[(n * d.P + p) * 2 * d.C + d.C + c] = dz * e * sg * 0.5f * (tv > 0.f ? 1.f : 0.2f);
        }
    }
}

static inline unsigned grid_for3(long items) {
    long b = (items + 255) / 256;
    if (b > 8192) b = 8192;
    if (b < 1) b = 1;
    return (unsigned)b;
}

}  // namespace ga

using namespace ga;

extern "C" int ga_avae(const ga_avae_desc* d, void* s) {
    ga::clear_stale_error();
    if (!d || !d->x || !d->y || d->N <= 0 || d->C <= 0) return GA_E_BADARG;
    if (d->backward && !d->dy) return GA_E_BADARG;
    hipStream_t st = (hipStream_t)s;
    switch (d->mode) {
        case GA_AVAE_ADAIN: {
            if (d->P <= 0 || !d->c || !d->y2 || (d->backward && !d->s) || ((d->a == nullptr) != (d->b == nullptr))) return GA_E_BADARG;
            if (d->C % 4) return GA_E_UNSUPPORTED;
            const void* ptrs[] = {d->x, d->b, d->c, d->dy, d->y};
            for (const void* p : ptrs) if (p && !aligned16(p)) return GA_E_ALIGN;
            const int nchunks = (d->C + 31) / 32;
            hipLaunchKernelGGL(avae_adain_kernel, dim3((unsigned)(d->N * nchunks)), dim3(256), 0, st, *d, nchunks);
            break;
        }
        case GA_AVAE_AVGPOOL: {
            if (d->k <= 0 || d->H <= 0 || d->W <= 0 || d->H % d->k || d->W % d->k) return GA_E_BADARG;
            if (d->C % 4) return GA_E_UNSUPPORTED;
            if (!aligned16(d->x) || !aligned16(d->y) || (d->dy && !aligned16(d->dy))) return GA_E_ALIGN;
            const long total4 = (long)d->N * (d->backward ? (long)d->H * d->W : (long)(d->H / d->k) * (d->W / d->k)) * (d->C / 4);
            hipLaunchKernelGGL(avae_avgpool_kernel, dim3(grid_for3(total4)), dim3(256), 0, st, *d, total4);
            break;
        }
        case GA_AVAE_PIXELNORM:
            hipLaunchKernelGGL(avae_pixelnorm_kernel, dim3((unsigned)((d->N + 3) / 4)), dim3(256), 0, st, *d);
            break;
        case GA_AVAE_SAMPLE: {
            if (!d->a || d->P <= 0) return GA_E_BADARG;
            const long total = (long)d->N * d->P * d->C;
            hipLaunchKernelGGL(avae_sample_kernel, dim3(grid_for3(total)), dim3(256), 0, st, *d, total);
            break;
        }
        default: return GA_E_UNSUPPORTED;
    }
    return check_launch();
}
