// fs2_norm.hip — wavefront-reduction normalisation kernels (HBM-bound).
//   add+dropout+LayerNorm+mask   (reference transformer/SubLayers.py:54-55,90-91; Layers.py:25,28)
//   ReLU'd conv -> LayerNorm -> dropout (reference model/modules.py:209-240)
//   BatchNorm1d train/eval + tanh + dropout (reference transformer/Layers.py:129-137)
// Statistics are always fp32 (two-pass, in registers); one 64-lane wave owns one row of C channels.
#include "fs2_common.h"

#define FS2_LN_MAXV 8   // C <= 8*256 = 2048
#define FS2_LN_BWD_GRID 1024

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

// ---------------------------------------------------------------- C = 256, bf16: the transformer's LayerNorm shape
// r02i ISA of the generic kernel: one 8-byte load per lane and row, y waited for before the residual load is even issued,
// then twelve ds_bpermute_b32 (LDS crossbar, each individually waited) for the two wave sums, then lens, then gamma/beta -
// five dependent memory round trips per row with at most 512 B per wave in flight (2.6-3.1 TB/s).  Here HALF a wave owns a
// row (32 lanes x 16 B), a wave owns 2 RP rows whose loads (y, residual, gamma, beta, lens) are all issued before the first
// use, and the row sums run on DPP (quad_perm, row_half_mirror, row_mirror) plus ONE cross-row permute.
#define FS2_DPP(v, CTRL) __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, (v)), (CTRL), 0xF, 0xF, true))
__device__ __forceinline__ float half_sum32(float v) {          // sum over each 32-lane half, result in every lane of the half
    v += FS2_DPP(v, 0xB1);      // quad_perm [1,0,3,2]
    v += FS2_DPP(v, 0x4E);      // quad_perm [2,3,0,1]
    v += FS2_DPP(v, 0x141);     // row_half_mirror: lane i <-> 7 - i of its group of 8
    v += FS2_DPP(v, 0x140);     // row_mirror: lane i <-> 15 - i of its row of 16
    v += __shfl_xor(v, 16, 64);
    return v;
}
__device__ __forceinline__ void unpack8(const uint4& u, float* f) {
    f[0] = __uint_as_float(u.x << 16); f[1] = __uint_as_float(u.x & 0xffff0000u);
    f[2] = __uint_as_float(u.y << 16); f[3] = __uint_as_float(u.y & 0xffff0000u);
    f[4] = __uint_as_float(u.z << 16); f[5] = __uint_as_float(u.z & 0xffff0000u);
    f[6] = __uint_as_float(u.w << 16); f[7] = __uint_as_float(u.w & 0xffff0000u);
}
__device__ __forceinline__ uint4 pack8f(const float* f) {
    return make_uint4(pack_bf16x2(f[0], f[1]), pack_bf16x2(f[2], f[3]), pack_bf16x2(f[4], f[5]), pack_bf16x2(f[6], f[7]));
}

template <int RP>
__global__ void __launch_bounds__(256) ln_fwd_c256_bf16_kernel(LnArgs a) {
    const int lane = threadIdx.x & 63, hl = lane & 31, hh = lane >> 5;
    const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * (2 * RP) + hh;
    const int c = hl * 8;
    bf16_t* y = reinterpret_cast<bf16_t*>(a.y);
    const bf16_t* res = reinterpret_cast<const bf16_t*>(a.res);
    bf16_t* out = reinterpret_cast<bf16_t*>(a.out);
    if (a.seed_dev) { uint64_t o = *a.seed_dev; a.seed_pre += o; a.seed_post += o; }
    uint4 qy[RP], qr[RP];
    int len[RP];
#pragma unroll
    for (int i = 0; i < RP; ++i) {
        const int row = min(row0 + 2 * i, a.rows - 1);
        qy[i] = *reinterpret_cast<const uint4*>(y + (size_t)row * 256 + c);
        len[i] = a.S;
    }
    if (res) {
#pragma unroll
        for (int i = 0; i < RP; ++i) qr[i] = *reinterpret_cast<const uint4*>(res + (size_t)min(row0 + 2 * i, a.rows - 1) * 256 + c);
    }
    if (a.lens) {
#pragma unroll
        for (int i = 0; i < RP; ++i) len[i] = a.lens[min(row0 + 2 * i, a.rows - 1) / a.S];
    }
    const float4 g0 = *reinterpret_cast<const float4*>(a.gamma + c), g1 = *reinterpret_cast<const float4*>(a.gamma + c + 4);
    const float4 b0 = *reinterpret_cast<const float4*>(a.beta + c), b1 = *reinterpret_cast<const float4*>(a.beta + c + 4);
    const float gm[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
    const float bt[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
    const float ik_pre = a.p_pre > 0.f ? 1.f / (1.f - a.p_pre) : 1.f;
    const float ik_post = a.p_post > 0.f ? 1.f / (1.f - a.p_post) : 1.f;
#pragma unroll
    for (int i = 0; i < RP; ++i) {
        const int row = row0 + 2 * i;
        const bool ok = row < a.rows;
        const uint32_t e = (uint32_t)row * 256u + (uint32_t)c;
        float x[8];
        unpack8(qy[i], x);
        if (a.p_pre > 0.f) {
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] *= fs2_drop_scale(a.seed_pre, e + j, a.p_pre, ik_pre);
        }
        if (res) {
            float r[8];
            unpack8(qr[i], r);
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] += r[j];
        }
        if (res || a.p_pre > 0.f) {
            const uint4 z = pack8f(x);
            if (ok) *reinterpret_cast<uint4*>(y + (size_t)row * 256 + c) = z;
            unpack8(z, x);                       // statistics on the values as stored: backward sees the same z
        }
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) s += x[j];
        const float mean = half_sum32(s) * (1.f / 256.f);
        float sq = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float d = x[j] - mean; sq += d * d; }
        const float rstd = rsqrtf(half_sum32(sq) * (1.f / 256.f) + a.eps);
        if (hl == 0 && ok) { a.mean[row] = mean; a.rstd[row] = rstd; }
        const bool pad = a.lens && (row - (row / a.S) * a.S) >= len[i];
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (x[j] - mean) * rstd * gm[j] + bt[j];
        if (a.p_post > 0.f) {
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] *= fs2_drop_scale(a.seed_post, e + j, a.p_post, ik_post);
        }
        if (pad) {
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = 0.f;
        }
        if (ok) *reinterpret_cast<uint4*>(out + (size_t)row * 256 + c) = pack8f(o);
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
    static const int ln_fast = fs2_dev_env("FS2_LN_FAST", 3);       // dev A/B only: bit 0 forward, bit 1 backward
    if ((ln_fast & 1) && dtype == FS2_BF16 && C == 256 && (((uintptr_t)y | (uintptr_t)res | (uintptr_t)out) & 15) == 0) {
        ln_fwd_c256_bf16_kernel<2><<<fs2_cdiv(a.rows, 16), 256, 0, stream>>>(a);
        FS2_CHECK_LAUNCH("ln_fwd");
        return FS2_OK;
    }
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
    const void* z; const void* dout; const void* dout2;   // upstream gradient = dout (+ dout2 when non-null)
    const float* gamma; const int32_t* lens;
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
                if (a.dout2 && !pad) {
                    const float4 g2 = ld4<T>(reinterpret_cast<const T*>(a.dout2) + (size_t)row * a.C + c);
                    g.x += g2.x; g.y += g2.y; g.z += g2.z; g.w += g2.w;
                }
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
    if (a.dgamma == nullptr && blockIdx.x == 0 && threadIdx.x == 0)      // deferred form only: the count fs2_ln_bwd_reduce reads (ws holds GRID*2*C + 4 floats then)
        reinterpret_cast<int*>(a.partial)[(size_t)FS2_LN_BWD_GRID * 2 * a.C] = (int)gridDim.x;
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

// C = 256, bf16 backward: half a wave per row with 16-byte lanes (as ln_fwd_c256_bf16_kernel), and the NEXT row pair's loads
// (z, dout, d1_add, mean, rstd, lens) are issued before the current pair is reduced and stored, so every wave keeps a full
// set of loads in flight across its whole row loop.  The host sizes the grid so that all waves run the same number of
// iterations (44400 rows: 925 workgroups x 3 instead of 1024 x 2.7).
struct LnBwdRow { uint4 z, g, g2, r; float mean, rstd; int len; };
__device__ __forceinline__ void ln_bwd_c256_load(const LnBwdArgs& a, int row, int c, LnBwdRow& q) {
    const int rc = min(row, a.rows - 1);
    q.z = *reinterpret_cast<const uint4*>(reinterpret_cast<const bf16_t*>(a.z) + (size_t)rc * 256 + c);
    q.g = *reinterpret_cast<const uint4*>(reinterpret_cast<const bf16_t*>(a.dout) + (size_t)rc * 256 + c);
    if (a.dout2) q.g2 = *reinterpret_cast<const uint4*>(reinterpret_cast<const bf16_t*>(a.dout2) + (size_t)rc * 256 + c);
    if (a.d1 && a.d1_add) q.r = *reinterpret_cast<const uint4*>(reinterpret_cast<const bf16_t*>(a.d1_add) + (size_t)rc * 256 + c);
    q.mean = a.mean[rc]; q.rstd = a.rstd[rc];
    q.len = a.lens ? a.lens[rc / a.S] : a.S;
}

__global__ void __launch_bounds__(512) ln_bwd_c256_bf16_kernel(LnBwdArgs a) {
    __shared__ __attribute__((aligned(16))) float s_red[16][256];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, hl = lane & 31, hh = lane >> 5;
    const int c = hl * 8;
    float ag[8], ab[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { ag[j] = 0.f; ab[j] = 0.f; }
    const float ik_pre = a.p_pre > 0.f ? 1.f / (1.f - a.p_pre) : 1.f;
    const float ik_post = a.p_post > 0.f ? 1.f / (1.f - a.p_post) : 1.f;
    if (a.seed_dev) { uint64_t o = *a.seed_dev; a.seed_pre += o; a.seed_post += o; }
    const float4 g0 = *reinterpret_cast<const float4*>(a.gamma + c), g1 = *reinterpret_cast<const float4*>(a.gamma + c + 4);
    const float gm[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
    const int stride = gridDim.x * 16;
    int row = (blockIdx.x * 8 + w) * 2 + hh;
    LnBwdRow nx;
    nx.r = make_uint4(0, 0, 0, 0);
    nx.g2 = make_uint4(0, 0, 0, 0);
    ln_bwd_c256_load(a, row, c, nx);
    for (; row - hh < a.rows; row += stride) {
        const LnBwdRow q = nx;
        if (row - hh + stride < a.rows) ln_bwd_c256_load(a, row + stride, c, nx);     // wave-uniform condition
        const bool ok = row < a.rows;
        const bool live = ok && (row - (row / a.S) * a.S) < q.len;                     // padded rows: g = 0
        const uint32_t e = (uint32_t)row * 256u + (uint32_t)c;
        float z[8], g[8], x[8];
        unpack8(q.z, z);
        unpack8(q.g, g);
        if (a.dout2) {                                   // (kernel-uniform)
            float g2[8];
            unpack8(q.g2, g2);
#pragma unroll
            for (int j = 0; j < 8; ++j) g[j] += g2[j];
        }
        if (a.p_post > 0.f) {
#pragma unroll
            for (int j = 0; j < 8; ++j) g[j] *= fs2_drop_scale(a.seed_post, e + j, a.p_post, ik_post);
        }
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (!live) g[j] = 0.f;
            x[j] = (z[j] - q.mean) * q.rstd;
            ag[j] += g[j] * x[j];
            ab[j] += g[j];
            g[j] *= gm[j];
            s1 += g[j];
            s2 += g[j] * x[j];
        }
        s1 = half_sum32(s1) * (1.f / 256.f);
        s2 = half_sum32(s2) * (1.f / 256.f);
        float dz[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) dz[j] = q.rstd * (g[j] - s1 - x[j] * s2);
        if (a.d1) {
            float o[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = dz[j];
            if (a.d1_add) {
                float r[8];
                unpack8(q.r, r);
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] += r[j];
            }
            if (ok) *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(a.d1) + (size_t)row * 256 + c) = pack8f(o);
        }
        if (a.d2) {
            if (a.p_pre > 0.f) {
#pragma unroll
                for (int j = 0; j < 8; ++j) dz[j] *= fs2_drop_scale(a.seed_pre, e + j, a.p_pre, ik_pre);
            }
            if (a.relu_bwd) {
#pragma unroll
                for (int j = 0; j < 8; ++j) if (!(z[j] > 0.f)) dz[j] = 0.f;
            }
            if (ok) *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(a.d2) + (size_t)row * 256 + c) = pack8f(dz);
        }
    }
    if (a.dgamma == nullptr && blockIdx.x == 0 && threadIdx.x == 0)      // deferred form only (see ln_bwd_kernel)
        reinterpret_cast<int*>(a.partial)[(size_t)FS2_LN_BWD_GRID * 2 * 256] = (int)gridDim.x;
    // dgamma / dbeta partials of the workgroup: 16 half-waves -> LDS -> one column per thread -> partial[block][2][256]
    const int hw = w * 2 + hh;
    *reinterpret_cast<float4*>(&s_red[hw][c]) = make_float4(ag[0], ag[1], ag[2], ag[3]);
    *reinterpret_cast<float4*>(&s_red[hw][c + 4]) = make_float4(ag[4], ag[5], ag[6], ag[7]);
    __syncthreads();
    if (threadIdx.x < 256) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += s_red[k][threadIdx.x];
        a.partial[((size_t)blockIdx.x * 2) * 256 + threadIdx.x] = t;
    }
    __syncthreads();
    *reinterpret_cast<float4*>(&s_red[hw][c]) = make_float4(ab[0], ab[1], ab[2], ab[3]);
    *reinterpret_cast<float4*>(&s_red[hw][c + 4]) = make_float4(ab[4], ab[5], ab[6], ab[7]);
    __syncthreads();
    if (threadIdx.x < 256) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += s_red[k][threadIdx.x];
        a.partial[((size_t)blockIdx.x * 2 + 1) * 256 + threadIdx.x] = t;
    }
}

// sums the per-block partials [nblocks][2C]: grid (ceil(2C/256), row-groups); every thread owns one column and walks
// its row-group with independent (unrolled) coalesced loads, then one atomic per column and row-group (<= 32-way).
// (r01h: the first version walked all 1024 partial rows with 16 blocks and cost a flat 21 us per LayerNorm.)
__global__ void __launch_bounds__(256) ln_bwd_reduce_kernel(const float* __restrict__ partial, int nblocks, int C,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= 2 * C) return;
    if (nblocks < 0) nblocks = reinterpret_cast<const int*>(partial)[(size_t)FS2_LN_BWD_GRID * 2 * C];   // deferred form: the count the backward kernel left
    const int per = (nblocks + gridDim.y - 1) / gridDim.y;
    const int b0 = blockIdx.y * per, b1 = min(nblocks, b0 + per);
    float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
    int b = b0;
    for (; b + 4 <= b1; b += 4) {
        acc0 += partial[(size_t)b * 2 * C + c];
        acc1 += partial[(size_t)(b + 1) * 2 * C + c];
        acc2 += partial[(size_t)(b + 2) * 2 * C + c];
        acc3 += partial[(size_t)(b + 3) * 2 * C + c];
    }
    for (; b < b1; ++b) acc0 += partial[(size_t)b * 2 * C + c];
    const float t = (acc0 + acc1) + (acc2 + acc3);
    atomicAdd(c < C ? dgamma + c : dbeta + (c - C), t);
}

extern "C" int fs2_ln_bwd_sum(const void* z, const void* dout, const void* dout2, const float* gamma, const int32_t* lens, const float* mean,
                              const float* rstd, const void* d1_add, void* d1, void* d2, float* dgamma, float* dbeta,
                              float* partial_ws, int B, int S, int C, float p_pre, uint64_t seed_pre, float p_post, uint64_t seed_post,
                              const uint64_t* seed_dev, int relu_bwd, int dtype, hipStream_t stream);
extern "C" int fs2_ln_bwd(const void* z, const void* dout, const float* gamma, const int32_t* lens, const float* mean,
                          const float* rstd, const void* d1_add, void* d1, void* d2, float* dgamma, float* dbeta,
                          float* partial_ws, int B, int S, int C, float p_pre, uint64_t seed_pre, float p_post, uint64_t seed_post,
                          const uint64_t* seed_dev, int relu_bwd, int dtype, hipStream_t stream) {
    return fs2_ln_bwd_sum(z, dout, nullptr, gamma, lens, mean, rstd, d1_add, d1, d2, dgamma, dbeta, partial_ws, B, S, C, p_pre, seed_pre,
                          p_post, seed_post, seed_dev, relu_bwd, dtype, stream);
}

extern "C" int fs2_ln_bwd_sum(const void* z, const void* dout, const void* dout2, const float* gamma, const int32_t* lens, const float* mean,
                              const float* rstd, const void* d1_add, void* d1, void* d2, float* dgamma, float* dbeta,
                              float* partial_ws, int B, int S, int C, float p_pre, uint64_t seed_pre, float p_post, uint64_t seed_post,
                              const uint64_t* seed_dev, int relu_bwd, int dtype, hipStream_t stream) {
    FS2_CHECK_ARG(z && dout && gamma && mean && rstd && ((dgamma != nullptr) == (dbeta != nullptr)), "ln_bwd: null pointer");
    const bool reduce_now = dgamma != nullptr;              // both null: the caller runs fs2_ln_bwd_reduce later (any stream)
    FS2_CHECK_ARG(C > 0 && C % 4 == 0 && C <= 256 * FS2_LN_MAXV, "ln_bwd: unsupported C=%d", C);
    // the wave-uniform early break in the reduction requires whole 256-channel groups
    FS2_CHECK_ARG(C % 256 == 0, "ln_bwd: C=%d must be a multiple of 256", C);
    LnBwdArgs a;
    a.z = z; a.dout = dout; a.dout2 = dout2; a.gamma = gamma; a.lens = lens; a.mean = mean; a.rstd = rstd; a.d1_add = d1_add; a.d1 = d1; a.d2 = d2;
    a.dgamma = dgamma; a.dbeta = dbeta; a.rows = B * S; a.S = S; a.C = C; a.p_pre = p_pre; a.p_post = p_post;
    a.seed_pre = seed_pre; a.seed_post = seed_post; a.seed_dev = seed_dev; a.relu_bwd = relu_bwd;
    FS2_CHECK_ARG(partial_ws, "ln_bwd: partial_ws (FS2_LN_BWD_GRID*2*C floats) is required");
    a.partial = partial_ws;
    if (a.rows == 0) {                                      // nothing to launch; the deferred reduce must still find a count (0 blocks)
        if (!reduce_now && hipMemsetAsync(partial_ws + (size_t)FS2_LN_BWD_GRID * 2 * C, 0, sizeof(int), stream) != hipSuccess) {
            fs2_set_error("ln_bwd: memset failed"); return FS2_ELAUNCH;
        }
        return FS2_OK;
    }
    int grid = fs2_cdiv(a.rows, 8);
    if (grid > FS2_LN_BWD_GRID) grid = FS2_LN_BWD_GRID;
    const bool al16 = ((((uintptr_t)z | (uintptr_t)dout | (uintptr_t)dout2 | (uintptr_t)d1_add | (uintptr_t)d1 | (uintptr_t)d2) & 15) == 0);
    static const int ln_fast = fs2_dev_env("FS2_LN_FAST", 3);
    if ((ln_fast & 2) && dtype == FS2_BF16 && C == 256 && al16) {
        const int iters = fs2_cdiv(a.rows, 16 * FS2_LN_BWD_GRID);        // same trip count for every wave
        grid = fs2_cdiv(a.rows, 16 * iters);
        ln_bwd_c256_bf16_kernel<<<grid, 512, 0, stream>>>(a);
        if (reduce_now) ln_bwd_reduce_kernel<<<dim3(fs2_cdiv(2 * C, 256), 32), 256, 0, stream>>>(partial_ws, grid, C, dgamma, dbeta);
        FS2_CHECK_LAUNCH("ln_bwd");
        return FS2_OK;
    }
    const int nv = C <= 256 ? 1 : (C <= 512 ? 2 : (C <= 1024 ? 4 : 8));
#define LN_BWD_LAUNCH(TT, NVV) ln_bwd_kernel<TT, NVV><<<grid, 512, 0, stream>>>(a)
#define LN_BWD_NV(TT) do { if (nv == 1) LN_BWD_LAUNCH(TT, 1); else if (nv == 2) LN_BWD_LAUNCH(TT, 2); else if (nv == 4) LN_BWD_LAUNCH(TT, 4); else LN_BWD_LAUNCH(TT, 8); } while (0)
    if (dtype == FS2_F32) LN_BWD_NV(float);
    else if (dtype == FS2_BF16) LN_BWD_NV(bf16_t);
    else { fs2_set_error("ln_bwd: dtype"); return FS2_EDTYPE; }
    if (reduce_now) ln_bwd_reduce_kernel<<<dim3(fs2_cdiv(2 * C, 256), 32), 256, 0, stream>>>(partial_ws, grid, C, dgamma, dbeta);
    FS2_CHECK_LAUNCH("ln_bwd");
    return FS2_OK;
}

// The second half of fs2_ln_bwd called with dgamma = dbeta = NULL: sums the per-workgroup partials that call left in partial_ws
// (their count sits behind them) into dgamma / dbeta.  Nothing downstream of the LayerNorm reads the affine gradients before
// the optimiser, so the engine runs this on its weight-gradient stream, off the data-gradient chain.
extern "C" int fs2_ln_bwd_reduce(const float* partial_ws, int C, float* dgamma, float* dbeta, hipStream_t stream) {
    FS2_CHECK_ARG(partial_ws && dgamma && dbeta && C > 0 && C % 256 == 0, "ln_bwd_reduce: bad arguments");
    ln_bwd_reduce_kernel<<<dim3(fs2_cdiv(2 * C, 256), 32), 256, 0, stream>>>(partial_ws, -1, C, dgamma, dbeta);
    FS2_CHECK_LAUNCH("ln_bwd_reduce");
    return FS2_OK;
}

// ================================================================== BatchNorm1d over rows (PostNet)
// Column statistics over ALL M = B*T rows including padded frames (Appendix A #8).
// Thread mapping of every BN kernel: a thread owns 4 consecutive channels (one 8/16-byte chunk) and walks rows, so a
// wave reads whole contiguous row segments (the first version mapped one 2-byte element per lane and ran at ~1 TB/s);
// per-channel constants live in registers.  Chunks per row are padded to a power of two <= 256 so that
// (chunk, row-lane) is a shift/mask of threadIdx.
// Statistics in ONE pass: sums of d = x - shift and d^2 with shift = x[0][c] (a sample of the column, so |mean-shift|
// is of the order of the standard deviation and the variance formula does not cancel), converted afterwards to the
// (sum, sum of squared deviations) the ABI promises.
struct BnArgs {
    const void* x; const void* dout; const void* res; void* out;
    const float* mean_rstd; const float* gamma; const float* beta;
    float* sums;                 // stats / bwd sums (2C floats): WRITTEN by the last row group of each channel block (reducing modes)
    float* slab;                 // reducing modes: [row group][pass][C] partial sums (plain stores; summed in index order)
    int M, C, cprp, rows_per_block, act;
    int cblk;                    // channels per workgroup (grid.y walks the channel blocks); cprp * V when one workgroup spans the row
    float p; uint64_t seed; const uint64_t* seed_dev;
};

// tanh through one v_exp_f32 and one v_rcp_f32 (|error| < 3e-7 absolute; tanhf's branchy polynomial costs ~3x as much
// and runs three times per PostNet element: forward, backward sums, backward dx).  The reciprocal is the hardware one (1 ulp):
// written as 2.f / (e + 1.f) hipcc emits the IEEE division sequence (v_div_scale x 2, v_rcp, four v_fma, v_div_fmas,
// v_div_fixup - r04 ISA: 82 v_div_scale in the backward kernels' unrolled bodies), ~10 instructions per element and pass in
// kernels that are VALU-bound (2.3-3.3 TB/s against LayerNorm's 5+).
__device__ __forceinline__ float fs2_tanh(float x) {
    float e = __expf(2.f * fabsf(x));
    float t = 1.f - 2.f * __builtin_amdgcn_rcpf(e + 1.f);              // e = +inf -> rcp = 0 -> 1
    return copysignf(t, x);
}

// MODE 0: statistics   1: apply   2: backward pass 1 (sums)   3: backward pass 2 (dx)
// Reducing modes (0, 2) run FEW fat workgroups (NT = 1024 threads, 4 rows in flight per thread).  Streaming modes (1, 3) use
// many 256-thread workgroups.
// The column sums are BIT-REPRODUCIBLE (r04): a workgroup stores its partial sums into its own row of a slab and a tiny second
// launch (bn_slab_sum_kernel; for the forward statistics the finalize kernel that existed anyway) adds the rows in index order.
// (First attempt: the last workgroup to arrive - device-scope counter + __threadfence - summed them in the same launch.  Correct,
// and 4-6 x slower: on gfx950 an agent-scope fence is buffer_wbl2 + buffer_inv, i.e. every workgroup wrote back / invalidated
// its XCD's L2 right behind a convolution that had left 45 MB of dirty lines there - statistics 14 -> 82 us, backward sums
// 35 -> 127 us, profiles/r04p_kernel_trace_side0.md.)  Rounds 1-3 used one float atomic per channel and workgroup: the arrival order changed the last bits of mean / rstd / the backward sums from run to
// run, bf16 rounding turned some of those into one-spacing differences in the PostNet's tensors, and the whole-step gradient
// moved between runs of the same binary (profiles/r04a_spread_seed0.log: 6 runs, 6 different flat-gradient hashes, the
// mel_linear.weight error 0.77 x .. 2.47 x the emulated one; r04b_spread_local.log: the first tensors that differ are the ones
// behind the first BatchNorm).
// V = channels per thread: 4 (8 / 16 bytes per access) or, for bf16 with C % 8 == 0, 8 (16-byte lanes: r02 PMC showed the
// 8-byte version streaming at 1.5-3.5 TB/s with the per-thread bytes in flight as the limiter).
template <typename T, int V> __device__ __forceinline__ void bn_ldv(const T* p, float* f) {
    if constexpr (V == 8) { unpack8(*reinterpret_cast<const uint4*>(p), f); }
    else { float4 t = ld4<T>(p); f[0] = t.x; f[1] = t.y; f[2] = t.z; f[3] = t.w; }
}
template <typename T, int V> __device__ __forceinline__ void bn_stv(T* p, const float* f) {
    if constexpr (V == 8) { *reinterpret_cast<uint4*>(p) = pack8f(f); }
    else st4<T>(p, make_float4(f[0], f[1], f[2], f[3]));
}

template <typename T, int MODE, int NT, int V>
__global__ void __launch_bounds__(NT) bn_rows_kernel(BnArgs a) {
    __shared__ float s_red[NT * V];
    const T* x = reinterpret_cast<const T*>(a.x);
    const T* dout = reinterpret_cast<const T*>(a.dout);
    const T* res = reinterpret_cast<const T*>(a.res);
    T* out = reinterpret_cast<T*>(a.out);
    const int chunk = threadIdx.x & (a.cprp - 1), rl = threadIdx.x / a.cprp, nrl = NT / a.cprp;
    const int c = blockIdx.y * a.cblk + chunk * V;
    const bool cok = chunk * V < a.cblk && c < a.C;
    uint64_t seed = a.seed;
    if (a.seed_dev) seed += *a.seed_dev;
    const float ik = a.p > 0.f ? 1.f / (1.f - a.p) : 1.f;
    float mu[V], rs[V], gm[V], bt[V], m1[V], m2[V];
#pragma unroll
    for (int k = 0; k < V; ++k) { mu[k] = 0.f; rs[k] = 0.f; gm[k] = 0.f; bt[k] = 0.f; m1[k] = 0.f; m2[k] = 0.f; }
    if (cok) {
        if (MODE == 0) {
            bn_ldv<T, V>(x + c, mu);
        } else {
#pragma unroll
            for (int k = 0; k < V; ++k) {
                mu[k] = a.mean_rstd[c + k]; rs[k] = a.mean_rstd[a.C + c + k]; gm[k] = a.gamma[c + k]; bt[k] = a.beta[c + k];
                if (MODE == 3) { m1[k] = a.sums[c + k] / (float)a.M; m2[k] = a.sums[a.C + c + k] / (float)a.M; }
            }
        }
    }
    float a1[V], a2[V];
#pragma unroll
    for (int k = 0; k < V; ++k) { a1[k] = 0.f; a2[k] = 0.f; }
    const int r0 = blockIdx.x * a.rows_per_block, r1 = min(a.M, r0 + a.rows_per_block);
    auto row = [&](int r, float* v, const float* g) {
        const size_t e = (size_t)r * a.C + c;
        if (MODE == 0) {
#pragma unroll
            for (int k = 0; k < V; ++k) { float d = v[k] - mu[k]; a1[k] += d; a2[k] += d * d; }
        } else if (MODE == 1) {
#pragma unroll
            for (int k = 0; k < V; ++k) {
                float o = (v[k] - mu[k]) * rs[k] * gm[k] + bt[k];
                if (a.act == FS2_ACT_TANH) o = fs2_tanh(o);
                if (a.p > 0.f) o *= fs2_drop_scale(seed, (uint32_t)(e + k), a.p, ik);
                v[k] = o;
            }
            if (res) {
                float r4[V];
                bn_ldv<T, V>(res + e, r4);
#pragma unroll
                for (int k = 0; k < V; ++k) v[k] += r4[k];
            }
            bn_stv<T, V>(out + e, v);
        } else {
#pragma unroll
            for (int k = 0; k < V; ++k) {
                float xh = (v[k] - mu[k]) * rs[k];
                float gg = g[k];
                if (a.p > 0.f) gg *= fs2_drop_scale(seed, (uint32_t)(e + k), a.p, ik);
                if (a.act == FS2_ACT_TANH) { float t = fs2_tanh(xh * gm[k] + bt[k]); gg *= (1.f - t * t); }
                if (MODE == 2) { a1[k] += gg; a2[k] += gg * xh; }
                else v[k] = gm[k] * rs[k] * (gg - m1[k] - xh * m2[k]);
            }
            if (MODE == 3) bn_stv<T, V>(out + e, v);
        }
    };
    if (cok) {
        constexpr int U = (MODE == 0 || MODE == 2) ? 4 : 2;       // rows in flight per thread
        int r = r0 + rl;
        for (; r + (U - 1) * nrl < r1; r += U * nrl) {
            float vv[U][V], gg[U][V];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                bn_ldv<T, V>(x + (size_t)(r + u * nrl) * a.C + c, vv[u]);
                if (MODE >= 2) bn_ldv<T, V>(dout + (size_t)(r + u * nrl) * a.C + c, gg[u]);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) row(r + u * nrl, vv[u], gg[u]);
        }
        for (; r < r1; r += nrl) {
            float v1[V], g1[V];
            bn_ldv<T, V>(x + (size_t)r * a.C + c, v1);
            if (MODE >= 2) bn_ldv<T, V>(dout + (size_t)r * a.C + c, g1);
            row(r, v1, g1);
        }
    }
    if (MODE == 0 || MODE == 2) {                      // tree over the block's row lanes in LDS, then one slab row per workgroup
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            if (pass) __syncthreads();
#pragma unroll
            for (int k = 0; k < V; ++k) s_red[threadIdx.x * V + k] = pass ? a2[k] : a1[k];
            __syncthreads();
            for (int st = nrl >> 1; st > 0; st >>= 1) {          // nrl = NT / cprp is a power of two
                if (rl < st) {
#pragma unroll
                    for (int k = 0; k < V; ++k) s_red[threadIdx.x * V + k] += s_red[(threadIdx.x + st * a.cprp) * V + k];
                }
                __syncthreads();
            }
            if (rl == 0 && cok) {
#pragma unroll
                for (int k = 0; k < V; ++k) a.slab[((size_t)blockIdx.x * 2 + pass) * a.C + c + k] = s_red[threadIdx.x * V + k];
            }
        }
    }
}

// slab -> sums: out[p * C + c] = sum over the `rows` row groups of slab[(j * 2 + p) * C + c], IN INDEX ORDER (bit-reproducible);
// optionally the affine-gradient accumulation of the backward (acc_dbeta += sums[0..C), acc_dgamma += sums[C..2C)).
__global__ void bn_slab_sum_kernel(const float* __restrict__ slab, float* __restrict__ sums, float* __restrict__ acc_dbeta,
                                   float* __restrict__ acc_dgamma, int rows, int C) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;             // 0 .. 2C
    if (i >= 2 * C) return;
    const int p = i / C, c = i - p * C;
    float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;                    // four interleaved chains (fixed association), 4 loads in flight
    int j = 0;
    for (; j + 3 < rows; j += 4) {
        t0 += slab[((size_t)(j + 0) * 2 + p) * C + c]; t1 += slab[((size_t)(j + 1) * 2 + p) * C + c];
        t2 += slab[((size_t)(j + 2) * 2 + p) * C + c]; t3 += slab[((size_t)(j + 3) * 2 + p) * C + c];
    }
    for (; j < rows; ++j) t0 += slab[((size_t)j * 2 + p) * C + c];
    const float tot = (t0 + t1) + (t2 + t3);
    sums[i] = tot;
    if (acc_dbeta) { if (p == 0) acc_dbeta[c] += tot; else acc_dgamma[c] += tot; }
}

// shifted sums -> (sum, sum of squared deviations):  sum = S1 + M*shift ;  ssd = S2 - S1^2 / M
template <typename T>
__global__ void bn_stats_fix_kernel(const T* __restrict__ x, float* __restrict__ stats, int M, int C) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float sh = Elem<T>::ld(x + c), s1 = stats[c], s2 = stats[C + c];
    stats[c] = s1 + (float)M * sh;
    stats[C + c] = fmaxf(s2 - s1 * s1 / (float)M, 0.f);
}

// channels per thread of the STREAMING modes (1, 3): 8 when every bf16 row chunk is 16-byte addressable.  The reducing modes
// (0, 2: 1024-thread workgroups, 4 rows in flight per thread) stay at 4: at 8 the 128-register budget of a 1024-thread
// workgroup spills (r02m: statistics 19.9 -> 29.7 us, backward sums 53.8 -> 82.7 us; apply 28.9 -> 27.8, dx 34.4 -> 33.5).
static int bn_vec(const BnArgs& a, int C, int dtype) {
    const uintptr_t al = (uintptr_t)a.x | (uintptr_t)a.dout | (uintptr_t)a.res | (uintptr_t)a.out;
    static const int vec_env = fs2_dev_env("FS2_BN_VEC", 8);       // dev A/B only
    if (vec_env == 4) return 4;
    return (dtype == FS2_BF16 && C % 8 == 0 && (al & 15) == 0) ? 8 : 4;
}
// cblk = 0: a workgroup spans whole rows (the streaming modes).  cblk > 0: a workgroup owns cblk channels and grid.y walks
// the channel blocks - the REDUCING modes: every workgroup ends with one atomic per channel it owns, so the number of same-
// address atomics is (row groups) x 2C; with whole-row workgroups 256 of them meant 262 144 atomics on 1024 addresses, about
// half of the backward-sums pass (r02m: 54 us against 34 us for the dx pass that does the same arithmetic AND stores).  64-
// channel blocks leave 32 row groups: 8x fewer atomics at the same number of workgroups, 128-byte row segments per wave.
static dim3 bn_geometry(BnArgs& a, int M, int C, int nthreads, int want_blocks, int vec, int cblk = 0) {
    static const int cblk_env = fs2_dev_env("FS2_BN_CBLK", 1);       // dev A/B: 0 = whole-row workgroups in the reducing modes too
    if (!cblk_env) cblk = 0;
    int cpr = (cblk > 0 && cblk < C ? cblk : C) / vec, cprp = 1;
    while (cprp < cpr) cprp <<= 1;
    a.cprp = cprp;
    a.M = M; a.C = C;
    a.cblk = cprp * vec;
    const int gy = fs2_cdiv(C, a.cblk);
    // ~want_blocks workgroups in all, each a whole number of row-lane sweeps
    int nrl = nthreads / cprp;
    int rpb = fs2_cdiv(M, fs2_cdiv(want_blocks, gy));
    rpb = fs2_cdiv(rpb, nrl) * nrl;
    if (rpb < nrl) rpb = nrl;
    a.rows_per_block = rpb;
    return dim3((unsigned)fs2_cdiv(M, rpb), (unsigned)gy);
}
#define BN_LAUNCH(MODE, NT, grid) do { \
    if (dtype == FS2_F32) bn_rows_kernel<float, MODE, NT, 4><<<grid, NT, 0, stream>>>(a); \
    else if (dtype == FS2_BF16 && vec == 8) bn_rows_kernel<bf16_t, MODE, NT, 8><<<grid, NT, 0, stream>>>(a); \
    else if (dtype == FS2_BF16) bn_rows_kernel<bf16_t, MODE, NT, 4><<<grid, NT, 0, stream>>>(a); \
    else { fs2_set_error("bn: dtype"); return FS2_EDTYPE; } } while (0)
#define BN_REDUCE_BLOCKS 256       /* statistics */
#define BN_WS_ROWS 256             /* slab rows of a workspace >= row groups of any reducing launch (<= BN_*_BLOCKS) */
// workspace of the reducing launches: [2C reduced sums][BN_WS_ROWS x 2C partial sums]
extern "C" int fs2_bn_ws_floats(int C) { return C > 0 ? 2 * C + BN_WS_ROWS * 2 * C : 0; }
static int bn_ws_bind(BnArgs& a, float* ws, long ws_floats, int C, dim3 grid) {
    if (ws_floats < (long)fs2_bn_ws_floats(C)) {
        fs2_set_error("bn: workspace of %ld floats, fs2_bn_ws_floats(%d) = %d needed", ws_floats, C, fs2_bn_ws_floats(C));
        return FS2_EINVAL;
    }
    a.sums = ws;
    a.slab = ws + 2 * C;
    if (grid.x > BN_WS_ROWS) { fs2_set_error("bn: launch geometry %u x %u exceeds the workspace", grid.x, grid.y); return FS2_EINVAL; }
    return FS2_OK;
}
static void bn_slab_sum(const BnArgs& a, dim3 grid, int C, float* acc_dbeta, float* acc_dgamma, hipStream_t stream) {
    bn_slab_sum_kernel<<<fs2_cdiv(2 * C, 256), 256, 0, stream>>>(a.slab, a.sums, acc_dbeta, acc_dgamma, (int)grid.x, C);
}
#define BN_REDUCE_CBLK 64          /* channels per workgroup of the reducing modes */
#define BN_BWD1_BLOCKS 256         /* backward sums: tanh + dropout hash per element -> needs every CU */

// stats[0..C) = sum, stats[C..2C) = sum of squared deviations.  `stats` is a workspace of fs2_bn_ws_floats(C) floats whose
// counter words are zero on entry (zero the whole workspace once, when it is allocated; the kernels leave them zero).
extern "C" int fs2_bn_stats(const void* x, float* stats, long ws_floats, int M, int C, int dtype, hipStream_t stream) {
    FS2_CHECK_ARG(x && stats, "bn_stats: null pointer");
    FS2_CHECK_ARG(M > 0 && C > 0 && C % 4 == 0 && C <= 1024, "bn_stats: bad shape M=%d C=%d", M, C);
    BnArgs a = {};
    a.x = x;
    const int vec = 4;                   // reducing modes stay at 4 channels per thread (see bn_vec)
    dim3 grid = bn_geometry(a, M, C, 1024, BN_REDUCE_BLOCKS, vec, BN_REDUCE_CBLK);
    if (int e = bn_ws_bind(a, stats, ws_floats, C, grid)) return e;
    BN_LAUNCH(0, 1024, grid);
    bn_slab_sum(a, grid, C, nullptr, nullptr, stream);
    if (dtype == FS2_F32) bn_stats_fix_kernel<float><<<fs2_cdiv(C, 256), 256, 0, stream>>>((const float*)x, stats, M, C);
    else bn_stats_fix_kernel<bf16_t><<<fs2_cdiv(C, 256), 256, 0, stream>>>((const bf16_t*)x, stats, M, C);
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
// Train-mode statistics in two launches and no housekeeping: shifted sums (MODE 0) -> ONE kernel that converts them, updates
// the running statistics, writes mean / rstd and counts the batch (nn.BatchNorm1d.num_batches_tracked).  The workspace
// (fs2_bn_ws_floats(C) floats) is zeroed once by the caller, when it allocates it.
template <typename T>
__global__ void bn_fix_finalize_kernel(const T* __restrict__ x, float* __restrict__ stats, float* __restrict__ running_mean,
                                       float* __restrict__ running_var, long long* __restrict__ nbt, float* __restrict__ mean_rstd,
                                       int M, int C, float eps, float momentum, int rows) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c == 0 && nbt) *nbt += 1;
    if (c >= C) return;
    // the row groups' partial sums, added in a FIXED association: eight independent loads per sum in flight (a single dependent
    // chain over 32-128 L2 loads made this 256-thread kernel 9.5 us), reduced pairwise, groups of eight added in index order
    const float* slab = stats + 2 * C;
    float s1 = 0.f, s2 = 0.f;
    int j = 0;
    for (; j + 7 < rows; j += 8) {
        float a[8], b[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { a[u] = slab[((size_t)(j + u) * 2) * C + c]; b[u] = slab[((size_t)(j + u) * 2 + 1) * C + c]; }
        s1 += ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
        s2 += ((b[0] + b[1]) + (b[2] + b[3])) + ((b[4] + b[5]) + (b[6] + b[7]));
    }
    for (; j < rows; ++j) { s1 += slab[((size_t)j * 2) * C + c]; s2 += slab[((size_t)j * 2 + 1) * C + c]; }
    const float sh = Elem<T>::ld(x + c);
    const float sum = s1 + (float)M * sh, ssd = fmaxf(s2 - s1 * s1 / (float)M, 0.f);
    const float mean = sum / (float)M, var = ssd / (float)M;
    mean_rstd[c] = mean;
    mean_rstd[C + c] = rsqrtf(var + eps);
    if (running_mean) {
        const float unb = M > 1 ? ssd / (float)(M - 1) : var;
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * unb;
    }
}
extern "C" int fs2_bn_train_stats(const void* x, float* stats_ws, long ws_floats, float* running_mean, float* running_var,
                                  int64_t* num_batches_tracked, float* mean_rstd, int M, int C, float eps, float momentum, int dtype,
                                  hipStream_t stream) {
    FS2_CHECK_ARG(x && stats_ws && mean_rstd, "bn_train_stats: null pointer");
    FS2_CHECK_ARG(M > 0 && C > 0 && C % 4 == 0 && C <= 1024, "bn_train_stats: bad shape M=%d C=%d", M, C);
    BnArgs a = {};
    a.x = x;
    const int vec = 4;                   // reducing modes stay at 4 channels per thread (see bn_vec)
    dim3 grid = bn_geometry(a, M, C, 1024, BN_REDUCE_BLOCKS, vec, BN_REDUCE_CBLK);
    if (int e = bn_ws_bind(a, stats_ws, ws_floats, C, grid)) return e;
    BN_LAUNCH(0, 1024, grid);
    if (dtype == FS2_F32) bn_fix_finalize_kernel<float><<<fs2_cdiv(C, 256), 256, 0, stream>>>((const float*)x, stats_ws, running_mean, running_var, (long long*)num_batches_tracked, mean_rstd, M, C, eps, momentum, (int)grid.x);
    else bn_fix_finalize_kernel<bf16_t><<<fs2_cdiv(C, 256), 256, 0, stream>>>((const bf16_t*)x, stats_ws, running_mean, running_var, (long long*)num_batches_tracked, mean_rstd, M, C, eps, momentum, (int)grid.x);
    FS2_CHECK_LAUNCH("bn_train_stats");
    return FS2_OK;
}

extern "C" int fs2_bn_finalize(const float* stats, float* running_mean, float* running_var, float* mean_rstd, int M, int C,
                               float eps, float momentum, hipStream_t stream) {
    FS2_CHECK_ARG(stats && mean_rstd, "bn_finalize: null pointer");
    bn_finalize_kernel<<<fs2_cdiv(C, 256), 256, 0, stream>>>(stats, nullptr, nullptr, running_mean, running_var, mean_rstd, M, C, eps, momentum);
    FS2_CHECK_LAUNCH("bn_finalize");
    return FS2_OK;
}

// out = drop(act((x-mean)*rstd*gamma+beta)) (+ res)
extern "C" int fs2_bn_apply(const void* x, const float* mean_rstd, const float* gamma, const float* beta, const void* res,
                            void* out, int M, int C, int act, float p, uint64_t seed, const uint64_t* seed_dev, int dtype,
                            hipStream_t stream) {
    FS2_CHECK_ARG(x && mean_rstd && gamma && beta && out, "bn_apply: null pointer");
    FS2_CHECK_ARG(C % 4 == 0 && C <= 1024, "bn_apply: C=%d must be a multiple of 4, <= 1024", C);
    if ((size_t)M * C == 0) return FS2_OK;
    BnArgs a = {};
    a.x = x; a.res = res; a.out = out; a.mean_rstd = mean_rstd; a.gamma = gamma; a.beta = beta; a.act = act; a.p = p;
    a.seed = seed; a.seed_dev = seed_dev;
    const int vec = bn_vec(a, C, dtype);
    dim3 grid = bn_geometry(a, M, C, 256, 2048, vec);
    BN_LAUNCH(1, 256, grid);
    FS2_CHECK_LAUNCH("bn_apply");
    return FS2_OK;
}

// Backward, pass 1: g = dout * dropmask * act'(.)  ;  sums[c] += g, sums[C+c] += g*xhat  (also = dbeta, dgamma)
//           pass 2: dx = gamma*rstd*(g - sums[c]/M - xhat*sums[C+c]/M)
// sums (a workspace of fs2_bn_ws_floats(C) floats, counter words zero on entry) receives dbeta (first C) and dgamma (next C).
extern "C" int fs2_bn_bwd(const void* x, const void* dout, const float* mean_rstd, const float* gamma, const float* beta,
                          float* sums, long ws_floats, void* dx, int M, int C, int act, float p, uint64_t seed, const uint64_t* seed_dev,
                          int dtype, hipStream_t stream) {
    FS2_CHECK_ARG(x && dout && mean_rstd && gamma && beta && sums && dx, "bn_bwd: null pointer");
    FS2_CHECK_ARG(C % 4 == 0 && C <= 1024, "bn_bwd: C=%d must be a multiple of 4, <= 1024", C);
    if (M == 0) return FS2_OK;
    BnArgs a = {};
    a.x = x; a.dout = dout; a.out = dx; a.mean_rstd = mean_rstd; a.gamma = gamma; a.beta = beta; a.act = act;
    a.p = p; a.seed = seed; a.seed_dev = seed_dev;
    int vec = 4;                         // reducing pass: 4 channels per thread (see bn_vec)
    dim3 grid = bn_geometry(a, M, C, 1024, BN_BWD1_BLOCKS, vec, BN_REDUCE_CBLK);
    if (int e = bn_ws_bind(a, sums, ws_floats, C, grid)) return e;
    BN_LAUNCH(2, 1024, grid);
    bn_slab_sum(a, grid, C, nullptr, nullptr, stream);
    vec = bn_vec(a, C, dtype);
    grid = bn_geometry(a, M, C, 256, 2048, vec);
    BN_LAUNCH(3, 256, grid);
    FS2_CHECK_LAUNCH("bn_bwd");
    return FS2_OK;
}

// the same with the affine gradients accumulated: dgamma_acc / dbeta_acc (parameter-gradient buffers) += the reduced sums.
// Successive calls (any width-C layer, one stream) may share one workspace.
extern "C" int fs2_bn_bwd_acc(const void* x, const void* dout, const float* mean_rstd, const float* gamma, const float* beta,
                              float* sums, long ws_floats, void* dx, float* dgamma_acc, float* dbeta_acc, int M, int C, int act,
                              float p, uint64_t seed, const uint64_t* seed_dev, int dtype, hipStream_t stream) {
    FS2_CHECK_ARG(x && dout && mean_rstd && gamma && beta && sums && dx && dgamma_acc && dbeta_acc, "bn_bwd_acc: null pointer");
    FS2_CHECK_ARG(C % 4 == 0 && C <= 1024, "bn_bwd_acc: C=%d must be a multiple of 4, <= 1024", C);
    if (M == 0) return FS2_OK;
    BnArgs a = {};
    a.x = x; a.dout = dout; a.out = dx; a.mean_rstd = mean_rstd; a.gamma = gamma; a.beta = beta; a.act = act;
    a.p = p; a.seed = seed; a.seed_dev = seed_dev;
    int vec = 4;                         // reducing pass: 4 channels per thread (see bn_vec)
    dim3 grid = bn_geometry(a, M, C, 1024, BN_BWD1_BLOCKS, vec, BN_REDUCE_CBLK);
    if (int e = bn_ws_bind(a, sums, ws_floats, C, grid)) return e;
    BN_LAUNCH(2, 1024, grid);
    bn_slab_sum(a, grid, C, dbeta_acc, dgamma_acc, stream);       // ordered sums + the affine gradients (sum g = dbeta, sum g*xhat = dgamma)
    vec = bn_vec(a, C, dtype);
    grid = bn_geometry(a, M, C, 256, 2048, vec);
    BN_LAUNCH(3, 256, grid);
    FS2_CHECK_LAUNCH("bn_bwd_acc");
    return FS2_OK;
}
