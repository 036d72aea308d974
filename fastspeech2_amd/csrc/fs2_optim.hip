// fs2_optim.hip — flat-buffer optimiser step: global-norm clip + Adam in two launches over ONE contiguous
// parameter / gradient / moment buffer (reference train.py:93 clip_grad_norm_(1.0) + model/optimizer.py:10-51
// torch.optim.Adam with per-step lr).  All step-dependent scalars live in device memory (hyper[]), so the
// launch sequence is capturable in a hipGraph and replayable with no host-side argument changes.
#include "fs2_common.h"

// ||x||^2 in two stages with a FIXED summation order: per-block partials into the caller's workspace, then one block adds
// them in index order.  (The first version accumulated block sums with atomicAdd: the arrival order - and therefore the
// last bits of the norm, the clip coefficient and every parameter after the Adam step - differed from run to run and,
// in data-parallel training, from rank to rank although the all-reduced gradients were bit-identical: the two-rank test
// tests/test_ddp_gpu.py caught replicas drifting apart by 1 ulp per step.)
#define FS2_SUMSQ_BLOCKS 1024
__global__ void __launch_bounds__(256) sumsq_partial_kernel(const float* __restrict__ x, size_t n, float* __restrict__ ws) {
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
    if (threadIdx.x == 0) ws[blockIdx.x] = (s[0] + s[1]) + (s[2] + s[3]);
}
__global__ void __launch_bounds__(256) sumsq_final_kernel(const float* __restrict__ ws, int nblocks, float* __restrict__ out) {
    __shared__ float s[4];
    float acc = 0.f;
    for (int i = threadIdx.x; i < nblocks; i += 256) acc += ws[i];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) out[0] += (s[0] + s[1]) + (s[2] + s[3]);
}
// out[0] += sum x^2   (caller zeroes out; ws = FS2_SUMSQ_BLOCKS floats of workspace)
extern "C" int fs2_sumsq(const float* x, size_t n, float* out, float* ws, hipStream_t stream) {
    FS2_CHECK_ARG(x && out && ws, "sumsq: null pointer");
    FS2_CHECK_ARG(((uintptr_t)x & 15) == 0, "sumsq: x must be 16-byte aligned");
    if (n == 0) return FS2_OK;
    size_t blocks = (n / 4 + 255) / 256;
    if (blocks > FS2_SUMSQ_BLOCKS) blocks = FS2_SUMSQ_BLOCKS;
    if (blocks == 0) blocks = 1;
    sumsq_partial_kernel<<<(unsigned)blocks, 256, 0, stream>>>(x, n, ws);
    sumsq_final_kernel<<<1, 256, 0, stream>>>(ws, (int)blocks, out);
    FS2_CHECK_LAUNCH("sumsq");
    return FS2_OK;
}

// hyper = {lr, bias_correction1, bias_correction2, grad_scale_extra}
// clip coefficient (torch.nn.utils.clip_grad_norm_): c = min(1, max_norm / (sqrt(gnorm_sq) + 1e-6))
// Fused into the same pass over the flat buffers (each saves one full sweep of HBM per step):
//   p_lowp   (optional) the compute-dtype copy of the updated parameters - the forward weight pack of every layer is a
//            view into it (master weights are stored in the GEMM's own [n][tap][c] order), so no per-layer cast runs;
//   zero_grad (optional) the gradient buffer is cleared as it is consumed (optimizer.zero_grad()).
__global__ void __launch_bounds__(256) adam_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, size_t n4, const float* __restrict__ gnorm_sq,
                                                   float max_norm, const float* __restrict__ hyper, float b1, float b2, float eps,
                                                   float wd, bf16_t* __restrict__ p_lowp, int zero_grad) {
    float coef = 1.f;
    if (gnorm_sq && max_norm > 0.f) {
        float nrm = sqrtf(gnorm_sq[0]);
        coef = fminf(1.f, max_norm / (nrm + 1e-6f));
    }
    const float lr = hyper[0], bc1 = hyper[1], bc2 = hyper[2];
    const float step = lr / bc1, rbc2 = rsqrtf(bc2);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float4 g4 = reinterpret_cast<const float4*>(g)[i], p4 = reinterpret_cast<const float4*>(p)[i];
        float4 m4 = reinterpret_cast<const float4*>(m)[i], v4 = reinterpret_cast<const float4*>(v)[i];
        float gg[4] = {g4.x, g4.y, g4.z, g4.w}, pp[4] = {p4.x, p4.y, p4.z, p4.w};
        float mm[4] = {m4.x, m4.y, m4.z, m4.w}, vv[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float gi = gg[k] * coef;
            if (wd != 0.f) gi += wd * pp[k];
            mm[k] = b1 * mm[k] + (1.f - b1) * gi;
            vv[k] = b2 * vv[k] + (1.f - b2) * gi * gi;
            float denom = sqrtf(vv[k]) * rbc2 + eps;
            pp[k] = pp[k] - step * (mm[k] / denom);
        }
        reinterpret_cast<float4*>(m)[i] = make_float4(mm[0], mm[1], mm[2], mm[3]);
        reinterpret_cast<float4*>(v)[i] = make_float4(vv[0], vv[1], vv[2], vv[3]);
        reinterpret_cast<float4*>(p)[i] = make_float4(pp[0], pp[1], pp[2], pp[3]);
        if (p_lowp) st4<bf16_t>(p_lowp + i * 4, make_float4(pp[0], pp[1], pp[2], pp[3]));
        if (zero_grad) reinterpret_cast<float4*>(g)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
}
extern "C" int fs2_adam_step(float* p, float* g, float* m, float* v, size_t n, const float* gnorm_sq, float max_norm,
                             const float* hyper, float b1, float b2, float eps, float wd, void* p_lowp, int lowp_dtype,
                             int zero_grad, hipStream_t stream) {
    FS2_CHECK_ARG(p && g && m && v && hyper, "adam_step: null pointer");
    FS2_CHECK_ARG(n % 4 == 0 && (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0,
                  "adam_step: flat buffers must be 16-byte aligned with n %% 4 == 0 (n=%zu)", n);
    FS2_CHECK_ARG(!p_lowp || lowp_dtype == FS2_BF16, "adam_step: the low-precision parameter copy must be bf16");
    if (n == 0) return FS2_OK;
    size_t n4 = n / 4;
    size_t blocks = (n4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    adam_kernel<<<(unsigned)blocks, 256, 0, stream>>>(p, g, m, v, n4, gnorm_sq, max_norm, hyper, b1, b2, eps, wd, (bf16_t*)p_lowp, zero_grad);
    FS2_CHECK_LAUNCH("adam_step");
    return FS2_OK;
}
