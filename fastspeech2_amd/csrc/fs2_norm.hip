// fs2_norm.hip — wavefront-reduction normalisation kernels (HBM-bound).
//   add+dropout+LayerNorm+mask   (reference transformer/SubLayers.py:54-55,90-91; Layers.py:25,28)
//   ReLU'd conv -> LayerNorm -> dropout (reference model/modules.py:209-240)
//   BatchNorm1d train/eval + tanh + dropout (reference transformer/Layers.py:129-137)
// Statistics are always fp32 (two-pass, in registers); one 64-lane wave owns one row of C channels.
#include "fs2_common.h"

#define FS2_LN_MAXV 8   // C <= 8*256 = 2048
#define FS2_LN_BWD_GRID 512

struct LnArgs {
    void* y;               // in: GEMM output (bias included); overwritten with z = drop_pre(y) + res  (saved for bwd)
    const void* res;       // residual or null
    const float* gamma; const float* beta;
    const int32_t* lens;   // rows t >= lens[b] are written as 0, or null
    void* out;
    float* mean; float* rstd;   // [rows] saved statistics
    int rows, S, C;
    float eps;
    float p_pre, p_post;   // dropout before the residual add / after the LayerNorm
    uint64_t seed_pre, seed_post;
    const uint64_t* seed_dev;   // optional device-resident per-step seed offset (keeps hipGraph replays fresh)
};

template <typename T, int NV>
__global__ void ln_fwd_kernel(LnArgs a) {
    int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= a.rows) return;
    int lane = threadIdx.x & 63;
    T* y = reinterpret_cast<T*>(a.y) + (size_t)row * a.C;
    const T* res = a.res ? reinterpret_cast<const T*>(a.res) + (size_t)row * a.C : nullptr;
    T* out = reinterpret_cast<T*>(a.out) + (size_t)row * a.C;
    float4 v[NV];
    const float ik_pre = a.p_pre > 0.f ? 1.f / (1.f - a.p_pre) : 1.f;
    if (a.seed_dev) { uint64_t o = *a.seed_dev; a.seed_pre += o; a.seed_post += o; }
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        int c = lane * 4 + i * 256;
        if (c < a.C) {
            float4 x = ld4<T>(y + c);
            if (a.p_pre > 0.f) {
                uint32_t e = (uint32_t)row * (uint32_t)a.C + c;
                x.x *= fs2_drop_scale(a.seed_pre, e, a.p_pre, ik_pre);
                x.y *= fs2_drop_scale(a.seed_pre, e + 1, a.p_pre, ik_pre);
                x.z *= fs2_drop_scale(a.seed_pre, e + 2, a.p_pre, ik_pre);
                x.w *= fs2_drop_scale(a.seed_pre, e + 3, a.p_pre, ik_pre);
            }
            if (res) {
                float4 r = ld4<T>(res + c);
                x.x += r.x; x.y += r.y; x.z += r.z; x.w += r.w;
            }
            if (res || a.p_pre > 0.f) st4<T>(y + c, x);
            // statistics on the values as stored (matters for bf16 so that backward sees the same z)
            if (sizeof(T) == 2) x = ld4<T>(y + c);
            v[i] = x;
            sum += x.x + x.y + x.z + x.w;
        }
    }
    float mean = wave_sum(sum) / (float)a.C;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        int c = lane * 4 + i * 256;
        if (c < a.C) {
            float dx = v[i].x - mean, dy = v[i].y - mean, dz = v[i].z - mean, dw = v[i].w - mean;
            sq += dx * dx + dy * dy + dz * dz + dw * dw;
        }
    }
    float rstd = rsqrtf(wave_sum(sq) / (float)a.C + a.eps);
    if (lane == 0) { a.mean[row] = mean; a.rstd[row] = rstd; }
    bool pad = false;
    if (a.lens) { int b = row / a.S; pad = (row - b * a.S) >= a.lens[b]; }
    const float ik_post = a.p_post > 0.f ? 1.f / (1.f - a.p_post) : 1.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        int c = lane * 4 + i * 256;
        if (c < a.C) {
            float4 g = *reinterpret_cast<const float4*>(a.gamma + c);
            float4 bt = *reinterpret_cast<const float4*>(a.beta + c);
            float4 o;
            o.x = (v[i].x - mean) * rstd * g.x + bt.x;
            o.y = (v[i].y - mean) * rstd * g.y + bt.y;
            o.z = (v[i].z - mean) * rstd * g.z + bt.z;
            o.w = (v[i].w - mean) * rstd * g.w + bt.w;
            if (a.p_post > 0.f) {
                uint32_t e = (uint32_t)row * (uint32_t)a.C + c;
                o.x *= fs2_drop_scale(a.seed_post, e, a.p_post, ik_post);
                o.y *= fs2_drop_scale(a.seed_post, e + 1, a.p_post, ik_post);
                o.z *= fs2_drop_scale(a.seed_post, e + 2, a.p_post, ik_post);
                o.w *= fs2_drop_scale(a.seed_post, e + 3, a.p_post, ik_post);
            }
            if (pad) o = make_float4(0.f, 0.f, 0.f, 0.f);
            st4<T>(out + c, o);
        }
    }
}

extern "C" int fs2_ln_fwd(void* y, const void* res, const float* gamma, const float* beta, const int32_t* lens, void* out,
                          float* mean, float* rstd, int B, int S, int C, float eps, float p_pre, uint64_t seed_pre,
                          float p_post, uint64_t seed_post, const uint64_t* seed_dev, int dtype, hipStream_t stream) {
    FS2_CHECK_ARG(y && gamma && beta && out && mean && rstd, "ln_fwd: null pointer");
    FS2_CHECK_ARG(C > 0 && C % 4 == 0 && C <= 256 * FS2_LN_MAXV, "ln_fwd: unsupported C=%d", C);
    FS2_CHECK_ARG(p_pre < 1.f && p_post < 1.f, "ln_fwd: dropout p must be < 1");
    LnArgs a;
    a.y = y; a.res = res; a.gamma = gamma; a.beta = beta; a.lens = lens; a.out = out; a.mean = mean; a.rstd = rstd;
    a.rows = B * S; a.S = S; a.C = C; a.eps = eps; a.p_pre = p_pre; a.p_post = p_post; a.seed_pre = seed_pre; a.seed_post = seed_post;
    a.seed_dev = seed_dev;
    if (a.rows == 0) return FS2_OK;
    const int nv = C <= 256 ? 1 : (C <= 512 ? 2 : (C <= 1024 ? 4 : 8));
#define LN_FWD_LAUNCH(TT, NVV) ln_fwd_kernel<TT, NVV><<<fs2_cdiv(a.rows, 4), 256, 0, stream>>>(a)
#define LN_FWD_NV(TT) do { if (nv == 1) LN_FWD_LAUNCH(TT, 1); else if (nv == 2) LN_FWD_LAUNCH(TT, 2); else if (nv == 4) LN_FWD_LAUNCH(TT, 4); else LN_FWD_LAUNCH(TT, 8); } while (0)
    if (dtype == FS2_F32) LN_FWD_NV(float);
    else if (dtype == FS2_BF16) LN_FWD_NV(bf16_t);
    else { fs2_set_error("ln_fwd: dtype"); return FS2_EDTYPE; }
    FS2_CHECK_LAUNCH("ln_fwd");
    return FS2_OK;
}

// Backward.  g = dout * dropmask_post (0 on padded rows); xhat = (z-mean)*rstd
//   dz = rstd * (g*gamma - mean_c(g*gamma) - xhat * mean_c(g*gamma*xhat))
//   d1 = dz (+ d1_add)            -> gradient of the residual path / of z
//   d2 = dz * dropmask_pre * (relu_bwd ? z > 0 : 1)   -> gradient of the GEMM output y
//   dgamma += sum_rows g*xhat ; dbeta += sum_rows g
struct LnBwdArgs {
    const void* z; const void* dout; const float* gamma; const int32_t* lens;
    const float* mean; const float* rstd;
    const void* d1_add;    // optional tensor added into d1 (fuses the "+ upstream residual gradient")
    void* d1; void* d2;    // either may be null
    float* dgamma; float* dbeta;
    float* partial;        // workspace [grid][2][C]
    int rows, S, C;
    float p_pre, p_post; uint64_t seed_pre, seed_post;
    const uint64_t* seed_dev;
    int relu_bwd;
};

template <typename T, int NV>
__global__ void __launch_bounds__(512) ln_bwd_kernel(LnBwdArgs a) {
    __shared__ float s_red[8][64 * 4];
    int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float4 ag[NV], ab[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) { ag[i] = make_float4(0.f, 0.f, 0.f, 0.f); ab[i] = ag[i]; }
    const float ik_pre = a.p_pre > 0.f ? 1.f / (1.f - a.p_pre) : 1.f;
    const float ik_post = a.p_post > 0.f ? 1.f / (1.f - a.p_post) : 1.f;
    if (a.seed_dev) { uint64_t o = *a.seed_dev; a.seed_pre += o; a.seed_post += o; }
    for (int row = blockIdx.x * 8 + w; row < a.rows; row += gridDim.x * 8) {
        bool pad = false;
        if (a.lens) { int b = row / a.S; pad = (row - b * a.S) >= a.lens[b]; }
        const T* z = reinterpret_cast<const T*>(a.z) + (size_t)row * a.C;
        const T* dout = reinterpret_cast<const T*>(a.dout) + (size_t)row * a.C;
        float mean = a.mean[row], rstd = a.rstd[row];
        float4 xh[NV], gg[NV];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            int c = lane * 4 + i * 256;
            if (c < a.C) {
                float4 zz = ld4<T>(z + c);
                float4 g = pad ? make_float4(0.f, 0.f, 0.f, 0.f) : ld4<T>(dout + c);
                if (a.p_post > 0.f) {
                    uint32_t e = (uint32_t)row * (uint32_t)a.C + c;
                    g.x *= fs2_drop_scale(a.seed_post, e, a.p_post, ik_post);
                    g.y *= fs2_drop_scale(a.seed_post, e + 1, a.p_post, ik_post);
                    g.z *= fs2_drop_scale(a.seed_post, e + 2, a.p_post, ik_post);
                    g.w *= fs2_drop_scale(a.seed_post, e + 3, a.p_post, ik_post);
                }
                float4 x;
                x.x = (zz.x - mean) * rstd; x.y = (zz.y - mean) * rstd; x.z = (zz.z - mean) * rstd; x.w = (zz.w - mean) * rstd;
                ag[i].x += g.x * x.x; ag[i].y += g.y * x.y; ag[i].z += g.z * x.z; ag[i].w += g.w * x.w;
                ab[i].x += g.x; ab[i].y += g.y; ab[i].z += g.z; ab[i].w += g.w;
                float4 gm = *reinterpret_cast<const float4*>(a.gamma + c);
                g.x *= gm.x; g.y *= gm.y; g.z *= gm.z; g.w *= gm.w;
                s1 += g.x + g.y + g.z + g.w;
                s2 += g.x * x.x + g.y * x.y + g.z * x.z + g.w * x.w;
                xh[i] = x; gg[i] = g;
                // keep z sign for relu backward in xh? need z itself: recompute from xh below
            }
        }
        s1 = wave_sum(s1) / (float)a.C;
        s2 = wave_sum(s2) / (float)a.C;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            int c = lane * 4 + i * 256;
            if (c < a.C) {
                float4 dz;
                dz.x = rstd * (gg[i].x - s1 - xh[i].x * s2);
                dz.y = rstd * (gg[i].y - s1 - xh[i].y * s2);
                dz.z = rstd * (gg[i].z - s1 - xh[i].z * s2);
                dz.w = rstd * (gg[i].w - s1 - xh[i].w * s2);
                if (a.d1) {
                    float4 o = dz;
                    if (a.d1_add) {
                        float4 r = ld4<T>(reinterpret_cast<const T*>(a.d1_add) + (size_t)row * a.C + c);
                        o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
                    }
                    st4<T>(reinterpret_cast<T*>(a.d1) + (size_t)row * a.C + c, o);
                }
                if (a.d2) {
                    float4 o = dz;
                    if (a.p_pre > 0.f) {
                        uint32_t e = (uint32_t)row * (uint32_t)a.C + c;
                        o.x *= fs2_drop_scale(a.seed_pre, e, a.p_pre, ik_pre);
                        o.y *= fs2_drop_scale(a.seed_pre, e + 1, a.p_pre, ik_pre);
                        o.z *= fs2_drop_scale(a.seed_pre, e + 2, a.p_pre, ik_pre);
                        o.w *= fs2_drop_scale(a.seed_pre, e + 3, a.p_pre, ik_pre);
                    }
                    if (a.relu_bwd) {
                        float4 zz = ld4<T>(z + c);
                        if (!(zz.x > 0.f)) o.x = 0.f;
                        if (!(zz.y > 0.f)) o.y = 0.f;
                        if (!(zz.z > 0.f)) o.z = 0.f;
                        if (!(zz.w > 0.f)) o.w = 0.f;
                    }
                    st4<T>(reinterpret_cast<T*>(a.d2) + (size_t)row * a.C + c, o);
                }
            }
        }
    }
    // block reduction of dgamma / dbeta partials: 4 waves -> LDS -> wave 0 -> partial[block][2][C]
    // (a second tiny kernel sums the per-block partials: no same-address atomic storm)
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        int c = lane * 4 + i * 256;
        if (c >= a.C) break;   // uniform across waves for a given i
        __syncthreads();
        *reinterpret_cast<float4*>(&s_red[w][lane * 4]) = ag[i];
        __syncthreads();
        if (w == 0) {
            float4 t = ag[i];
            for (int k = 1; k < 8; ++k) { float4 o = *reinterpret_cast<float4*>(&s_red[k][lane * 4]); t.x += o.x; t.y += o.y; t.z += o.z; t.w += o.w; }
            *reinterpret_cast<float4*>(a.partial + ((size_t)blockIdx.x * 2) * a.C + c) = t;
        }
        __syncthreads();
        *reinterpret_cast<float4*>(&s_red[w][lane * 4]) = ab[i];
        __syncthreads();
        if (w == 0) {
            float4 t = ab[i];
            for (int k = 1; k < 8; ++k) { float4 o = *reinterpret_cast<float4*>(&s_red[k][lane * 4]); t.x += o.x; t.y += o.y; t.z += o.z; t.w += o.w; }
            *reinterpret_cast<float4*>(a.partial + ((size_t)blockIdx.x * 2 + 1) * a.C + c) = t;
        }
    }
}

// sums the per-block partials: block = 32 channels x 8 row-groups; coalesced 128-B reads, LDS tree at the end.
__global__ void ln_bwd_reduce_kernel(const float* __restrict__ partial, int nblocks, int C, float* __restrict__ dgamma,
                                     float* __restrict__ dbeta) {
    __shared__ float s[8][32];
    int c = blockIdx.x * 32 + (threadIdx.x & 31), rg = threadIdx.x >> 5;
    float acc = 0.f;
    if (c < 2 * C)
        for (int b = rg; b < nblocks; b += 8) acc += partial[(size_t)b * 2 * C + c];
    s[rg][threadIdx.x & 31] = acc;
    __syncthreads();
    if (rg == 0 && c < 2 * C) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) t += s[k][threadIdx.x];
        if (c < C) dgamma[c] += t; else dbeta[c - C] += t;
    }
}

extern "C" int fs2_ln_bwd(const void* z, const void* dout, const float* gamma, const int32_t* lens, const float* mean,
                          const float* rstd, const void* d1_add, void* d1, void* d2, float* dgamma, float* dbeta,
                          float* partial_ws, int B, int S, int C, float p_pre, uint64_t seed_pre, float p_post, uint64_t seed_post,
                          const uint64_t* seed_dev, int relu_bwd, int dtype, hipStream_t stream) {
    FS2_CHECK_ARG(z && dout && gamma && mean && rstd && dgamma && dbeta, "ln_bwd: null pointer");
    FS2_CHECK_ARG(C > 0 && C % 4 == 0 && C <= 256 * FS2_LN_MAXV, "ln_bwd: unsupported C=%d", C);
    // the wave-uniform early break in the reduction requires whole 256-channel groups
    FS2_CHECK_ARG(C % 256 == 0, "ln_bwd: C=%d must be a multiple of 256", C);
    LnBwdArgs a;
    a.z = z; a.dout = dout; a.gamma = gamma; a.lens = lens; a.mean = mean; a.rstd = rstd; a.d1_add = d1_add; a.d1 = d1; a.d2 = d2;
    a.dgamma = dgamma; a.dbeta = dbeta; a.rows = B * S; a.S = S; a.C = C; a.p_pre = p_pre; a.p_post = p_post;
    a.seed_pre = seed_pre; a.seed_post = seed_post; a.seed_dev = seed_dev; a.relu_bwd = relu_bwd;
    FS2_CHECK_ARG(partial_ws, "ln_bwd: partial_ws (FS2_LN_BWD_GRID*2*C floats) is required");
    a.partial = partial_ws;
    if (a.rows == 0) return FS2_OK;
    int grid = fs2_cdiv(a.rows, 8);
    if (grid > FS2_LN_BWD_GRID) grid = FS2_LN_BWD_GRID;
    const int nv = C <= 256 ? 1 : (C <= 512 ? 2 : (C <= 1024 ? 4 : 8));
#define LN_BWD_LAUNCH(TT, NVV) ln_bwd_kernel<TT, NVV><<<grid, 512, 0, stream>>>(a)
#define LN_BWD_NV(TT) do { if (nv == 1) LN_BWD_LAUNCH(TT, 1); else if (nv == 2) LN_BWD_LAUNCH(TT, 2); else if (nv == 4) LN_BWD_LAUNCH(TT, 4); else LN_BWD_LAUNCH(TT, 8); } while (0)
    if (dtype == FS2_F32) LN_BWD_NV(float);
    else if (dtype == FS2_BF16) LN_BWD_NV(bf16_t);
    else { fs2_set_error("ln_bwd: dtype"); return FS2_EDTYPE; }
    ln_bwd_reduce_kernel<<<fs2_cdiv(2 * C, 32), 256, 0, stream>>>(partial_ws, grid, C, dgamma, dbeta);
    FS2_CHECK_LAUNCH("ln_bwd");
    return FS2_OK;
}

// ================================================================== BatchNorm1d over rows (PostNet)
// Column statistics over ALL M = B*T rows including padded frames (Appendix A #8).
// pass 1: sum  -> mean ; pass 2: sum (x-mean)^2 -> biased var.  Partial sums via fp32 atomics.
template <typename T>
__global__ void bn_colsum_kernel(const T* __restrict__ x, const float* __restrict__ mean_in, float* __restrict__ out,
                                 int M, int C, int rows_per_block, float inv_m) {
    __shared__ float s[4][64];
    int col = blockIdx.x * 64 + (threadIdx.x & 63);
    int w = threadIdx.x >> 6;
    int mbeg = blockIdx.y * rows_per_block, mend = min(M, mbeg + rows_per_block);
    float acc = 0.f;
    if (col < C) {
        if (mean_in) {
            float mu = mean_in[col] * inv_m;
            for (int m = mbeg + w; m < mend; m += 4) { float d = Elem<T>::ld(x + (size_t)m * C + col) - mu; acc += d * d; }
        } else {
            for (int m = mbeg + w; m < mend; m += 4) acc += Elem<T>::ld(x + (size_t)m * C + col);
        }
    }
    s[w][threadIdx.x & 63] = acc;
    __syncthreads();
    if (w == 0 && col < C) atomicAdd(out + col, s[0][threadIdx.x] + s[1][threadIdx.x] + s[2][threadIdx.x] + s[3][threadIdx.x]);
}

// stats[0..C) = sum, stats[C..2C) = sum of squared deviations  (caller zeroes stats first)
extern "C" int fs2_bn_stats(const void* x, float* stats, int M, int C, int dtype, hipStream_t stream) {
    FS2_CHECK_ARG(x && stats, "bn_stats: null pointer");
    FS2_CHECK_ARG(M > 0 && C > 0, "bn_stats: bad shape");
    int rpb = 128;
    dim3 grid(fs2_cdiv(C, 64), fs2_cdiv(M, rpb));
    float inv_m = 1.0f / (float)M;
    if (dtype == FS2_F32) {
        bn_colsum_kernel<float><<<grid, 256, 0, stream>>>((const float*)x, nullptr, stats, M, C, rpb, inv_m);
        bn_colsum_kernel<float><<<grid, 256, 0, stream>>>((const float*)x, stats, stats + C, M, C, rpb, inv_m);
    } else if (dtype == FS2_BF16) {
        bn_colsum_kernel<bf16_t><<<grid, 256, 0, stream>>>((const bf16_t*)x, nullptr, stats, M, C, rpb, inv_m);
        bn_colsum_kernel<bf16_t><<<grid, 256, 0, stream>>>((const bf16_t*)x, stats, stats + C, M, C, rpb, inv_m);
    } else { fs2_set_error("bn_stats: dtype"); return FS2_EDTYPE; }
    FS2_CHECK_LAUNCH("bn_stats");
    return FS2_OK;
}

// running-stat update (momentum, unbiased variance) + per-channel scale/shift for the apply pass.
__global__ void bn_finalize_kernel(const float* __restrict__ stats, const float* __restrict__ gamma,
                                   const float* __restrict__ beta, float* __restrict__ running_mean,
                                   float* __restrict__ running_var, float* __restrict__ mean_rstd, int M, int C, float eps,
                                   float momentum) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float mean = stats[c] / (float)M;
    float var = stats[C + c] / (float)M;
    float rstd = rsqrtf(var + eps);
    mean_rstd[c] = mean;
    mean_rstd[C + c] = rstd;
    if (running_mean) {
        float unb = M > 1 ? stats[C + c] / (float)(M - 1) : var;
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * unb;
    }
    (void)gamma; (void)beta;
}
extern "C" int fs2_bn_finalize(const float* stats, float* running_mean, float* running_var, float* mean_rstd, int M, int C,
                               float eps, float momentum, hipStream_t stream) {
    FS2_CHECK_ARG(stats && mean_rstd, "bn_finalize: null pointer");
    bn_finalize_kernel<<<fs2_cdiv(C, 256), 256, 0, stream>>>(stats, nullptr, nullptr, running_mean, running_var, mean_rstd, M, C, eps, momentum);
    FS2_CHECK_LAUNCH("bn_finalize");
    return FS2_OK;
}

// out = drop(act((x-mean)*rstd*gamma+beta))
template <typename T>
__global__ void bn_apply_kernel(const T* __restrict__ x, const float* __restrict__ mean_rstd, const float* __restrict__ gamma,
                                const float* __restrict__ beta, const T* __restrict__ res, T* __restrict__ out, size_t total4,
                                int C, int act, float p, uint64_t seed, const uint64_t* __restrict__ seed_dev) {
    const float ik = p > 0.f ? 1.f / (1.f - p) : 1.f;
    if (seed_dev) seed += *seed_dev;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (size_t)gridDim.x * blockDim.x) {
        size_t e = i * 4;
        int c = (int)(e % C);
        float4 v = ld4<T>(x + e);
        float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float o = (vv[k] - mean_rstd[c + k]) * mean_rstd[C + c + k] * gamma[c + k] + beta[c + k];
            if (act == FS2_ACT_TANH) o = tanhf(o);
            if (p > 0.f) o *= fs2_drop_scale(seed, (uint32_t)(e + k), p, ik);
            vv[k] = o;
        }
        if (res) { float4 r = ld4<T>(res + e); vv[0] += r.x; vv[1] += r.y; vv[2] += r.z; vv[3] += r.w; }
        st4<T>(out + e, make_float4(vv[0], vv[1], vv[2], vv[3]));
    }
}
extern "C" int fs2_bn_apply(const void* x, const float* mean_rstd, const float* gamma, const float* beta, const void* res,
                            void* out, int M, int C, int act, float p, uint64_t seed, const uint64_t* seed_dev, int dtype,
                            hipStream_t stream) {
    FS2_CHECK_ARG(x && mean_rstd && gamma && beta && out, "bn_apply: null pointer");
    FS2_CHECK_ARG(C % 4 == 0, "bn_apply: C%%4");
    size_t total4 = (size_t)M * C / 4;
    if (total4 == 0) return FS2_OK;
    int grid = (int)((total4 + 255) / 256 < 4096 ? (total4 + 255) / 256 : 4096);
    if (dtype == FS2_F32) bn_apply_kernel<float><<<grid, 256, 0, stream>>>((const float*)x, mean_rstd, gamma, beta, (const float*)res, (float*)out, total4, C, act, p, seed, seed_dev);
    else if (dtype == FS2_BF16) bn_apply_kernel<bf16_t><<<grid, 256, 0, stream>>>((const bf16_t*)x, mean_rstd, gamma, beta, (const bf16_t*)res, (bf16_t*)out, total4, C, act, p, seed, seed_dev);
    else { fs2_set_error("bn_apply: dtype"); return FS2_EDTYPE; }
    FS2_CHECK_LAUNCH("bn_apply");
    return FS2_OK;
}

// Backward, pass 1: g = dout * dropmask * act'(.)  ;  sums[c] += g, sums[C+c] += g*xhat  (also = dbeta, dgamma)
//           pass 2: dx = gamma*rstd*(g - sums[c]/M - xhat*sums[C+c]/M)
template <typename T, int PASS>
__global__ void bn_bwd_kernel(const T* __restrict__ x, const T* __restrict__ dout, const float* __restrict__ mean_rstd,
                              const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ sums,
                              T* __restrict__ dx, int M, int C, int rows_per_block, int act, float p, uint64_t seed,
                              const uint64_t* __restrict__ seed_dev) {
    __shared__ float s[2][4][64];
    if (seed_dev) seed += *seed_dev;
    int col = blockIdx.x * 64 + (threadIdx.x & 63);
    int w = threadIdx.x >> 6;
    int mbeg = blockIdx.y * rows_per_block, mend = min(M, mbeg + rows_per_block);
    const float ik = p > 0.f ? 1.f / (1.f - p) : 1.f;
    float a1 = 0.f, a2 = 0.f;
    if (col < C) {
        float mu = mean_rstd[col], rs = mean_rstd[C + col], gm = gamma[col], bt = beta[col];
        float m1 = 0.f, m2 = 0.f;
        if (PASS == 2) { m1 = sums[col] / (float)M; m2 = sums[C + col] / (float)M; }
        for (int m = mbeg + w; m < mend; m += 4) {
            size_t e = (size_t)m * C + col;
            float xh = (Elem<T>::ld(x + e) - mu) * rs;
            float g = Elem<T>::ld(dout + e);
            if (p > 0.f) g *= fs2_drop_scale(seed, (uint32_t)e, p, ik);
            if (act == FS2_ACT_TANH) { float t = tanhf(xh * gm + bt); g *= (1.f - t * t); }
            if (PASS == 1) { a1 += g; a2 += g * xh; }
            else Elem<T>::st(dx + e, gm * rs * (g - m1 - xh * m2));
        }
    }
    if (PASS == 1) {
        s[0][w][threadIdx.x & 63] = a1; s[1][w][threadIdx.x & 63] = a2;
        __syncthreads();
        if (w == 0 && col < C) {
            atomicAdd(sums + col, s[0][0][threadIdx.x] + s[0][1][threadIdx.x] + s[0][2][threadIdx.x] + s[0][3][threadIdx.x]);
            atomicAdd(sums + C + col, s[1][0][threadIdx.x] + s[1][1][threadIdx.x] + s[1][2][threadIdx.x] + s[1][3][threadIdx.x]);
        }
    }
}
// sums (2C floats, zeroed by the caller) receives dbeta (first C) and dgamma (last C).
extern "C" int fs2_bn_bwd(const void* x, const void* dout, const float* mean_rstd, const float* gamma, const float* beta,
                          float* sums, void* dx, int M, int C, int act, float p, uint64_t seed, const uint64_t* seed_dev,
                          int dtype, hipStream_t stream) {
    FS2_CHECK_ARG(x && dout && mean_rstd && gamma && beta && sums && dx, "bn_bwd: null pointer");
    int rpb = 128;
    dim3 grid(fs2_cdiv(C, 64), fs2_cdiv(M, rpb));
    if (M == 0) return FS2_OK;
    if (dtype == FS2_F32) {
        bn_bwd_kernel<float, 1><<<grid, 256, 0, stream>>>((const float*)x, (const float*)dout, mean_rstd, gamma, beta, sums, (float*)dx, M, C, rpb, act, p, seed, seed_dev);
        bn_bwd_kernel<float, 2><<<grid, 256, 0, stream>>>((const float*)x, (const float*)dout, mean_rstd, gamma, beta, sums, (float*)dx, M, C, rpb, act, p, seed, seed_dev);
    } else if (dtype == FS2_BF16) {
        bn_bwd_kernel<bf16_t, 1><<<grid, 256, 0, stream>>>((const bf16_t*)x, (const bf16_t*)dout, mean_rstd, gamma, beta, sums, (bf16_t*)dx, M, C, rpb, act, p, seed, seed_dev);
        bn_bwd_kernel<bf16_t, 2><<<grid, 256, 0, stream>>>((const bf16_t*)x, (const bf16_t*)dout, mean_rstd, gamma, beta, sums, (bf16_t*)dx, M, C, rpb, act, p, seed, seed_dev);
    } else { fs2_set_error("bn_bwd: dtype"); return FS2_EDTYPE; }
    FS2_CHECK_LAUNCH("bn_bwd");
    return FS2_OK;
}
