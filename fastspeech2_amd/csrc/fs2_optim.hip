// fs2_optim.hip — flat-buffer optimiser step: global-norm clip + Adam in two launches over ONE contiguous
// parameter / gradient / moment buffer (reference train.py:93 clip_grad_norm_(1.0) + model/optimizer.py:10-51
// torch.optim.Adam with per-step lr).  All step-dependent scalars live in device memory (hyper[]), so the
// launch sequence is capturable in a hipGraph and replayable with no host-side argument changes.
#include "fs2_common.h"

__global__ void sumsq_kernel(const float* __restrict__ x, size_t n, float* __restrict__ out) {
    __shared__ float s[4];
    float acc = 0.f;
    size_t n4 = n / 4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float4 v = *reinterpret_cast<const float4*>(x + i * 4);
        acc += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) { float v = x[n4 * 4 + threadIdx.x]; acc += v * v; }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, s[0] + s[1] + s[2] + s[3]);
}
// out[0] += sum x^2   (caller zeroes out)
extern "C" int fs2_sumsq(const float* x, size_t n, float* out, hipStream_t stream) {
    FS2_CHECK_ARG(x && out, "sumsq: null pointer");
    FS2_CHECK_ARG(((uintptr_t)x & 15) == 0, "sumsq: x must be 16-byte aligned");
    if (n == 0) return FS2_OK;
    size_t blocks = (n / 4 + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    if (blocks == 0) blocks = 1;
    sumsq_kernel<<<(unsigned)blocks, 256, 0, stream>>>(x, n, out);
    FS2_CHECK_LAUNCH("sumsq");
    return FS2_OK;
}

// hyper = {lr, bias_correction1, bias_correction2, grad_scale_extra}
// clip coefficient (torch.nn.utils.clip_grad_norm_): c = min(1, max_norm / (sqrt(gnorm_sq) + 1e-6))
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                            size_t n, const float* __restrict__ gnorm_sq, float max_norm, const float* __restrict__ hyper,
                            float b1, float b2, float eps, float wd) {
    float coef = 1.f;
    if (gnorm_sq && max_norm > 0.f) {
        float nrm = sqrtf(gnorm_sq[0]);
        coef = fminf(1.f, max_norm / (nrm + 1e-6f));
    }
    const float lr = hyper[0], bc1 = hyper[1], bc2 = hyper[2];
    const float step = lr / bc1, rbc2 = rsqrtf(bc2);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float gi = g[i] * coef;
        float pi = p[i];
        if (wd != 0.f) gi += wd * pi;
        float mi = b1 * m[i] + (1.f - b1) * gi;
        float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi; v[i] = vi;
        float denom = sqrtf(vi) * rbc2 + eps;
        p[i] = pi - step * (mi / denom);
    }
}
extern "C" int fs2_adam_step(float* p, const float* g, float* m, float* v, size_t n, const float* gnorm_sq, float max_norm,
                             const float* hyper, float b1, float b2, float eps, float wd, hipStream_t stream) {
    FS2_CHECK_ARG(p && g && m && v && hyper, "adam_step: null pointer");
    if (n == 0) return FS2_OK;
    size_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    adam_kernel<<<(unsigned)blocks, 256, 0, stream>>>(p, g, m, v, n, gnorm_sq, max_norm, hyper, b1, b2, eps, wd);
    FS2_CHECK_LAUNCH("adam_step");
    return FS2_OK;
}
