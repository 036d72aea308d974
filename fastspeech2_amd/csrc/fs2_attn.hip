// fs2_attn.hip — fused scaled-dot-product attention with key-padding mask (forward + backward), dk = dv = 128.
// Reference: transformer/Modules.py:14-25 (bmm, /sqrt(dk), masked_fill(-inf), softmax, bmm) and the head
// split/merge of transformer/SubLayers.py:39-52.  The S x S score matrix never reaches HBM: online softmax
// over 32-key tiles, only the per-row log-sum-exp is saved for backward (flash-style).
//
// Layout: the fused QKV projection writes one buffer qkv[M][3*H*128] = (q | k | v), head h in columns
// h*128..h*128+127 of each third; the context is written as ctx[M][H*128] (== the reference's merged
// (B, S, H*dv) tensor), so no permute/contiguous copy exists anywhere.
//
// MFMA mapping (v_mfma_f32_32x32x2_f32, exact f32): scores are computed TRANSPOSED where that makes the
// probability tile land directly in the A-operand layout of the following product (C layout: lane = col,
// regs = rows (r&3)+8*(r>>2)+4*(lane>>5);  A layout: lane = row, k-slot = lane>>5) -> P never leaves registers.
// Storage dtype T may be bf16; tiles are widened to f32 when staged into LDS (first revision: f32 MFMA rate).
#include "fs2_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define DK 128
// row r of a [32][128] f32 tile, 16-B chunk c (0..31): XOR-swizzled so ds_read_b128 by 16 distinct rows is conflict-free
__device__ __forceinline__ int swz(int row, int chunk) { return row * 512 + ((chunk ^ (row & 15)) << 4); }
__device__ __forceinline__ int crow(int r, int h2) { return (r & 3) + 8 * (r >> 2) + 4 * h2; }

// stage a [32 rows][128] tile (rows row0.., column offset col0 of a ld-strided matrix) into LDS as f32, swizzled.
// rows >= nrows_valid are zero-filled. 256 threads (or NT threads).
template <typename T, int NT>
__device__ __forceinline__ void stage_tile(unsigned char* lds, const T* __restrict__ base, long ld, int row0, int nrows_valid,
                                           int tid) {
#pragma unroll
    for (int i = 0; i < 1024 / NT; ++i) {
        int idx = tid + NT * i;
        int r = idx >> 5, c = idx & 31;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < nrows_valid) v = ld4<T>(base + (size_t)(row0 + r) * ld + c * 4);
        *reinterpret_cast<float4*>(lds + swz(r, c)) = v;
    }
}

// ------------------------------------------------------------------ forward
template <typename T>
__global__ void __launch_bounds__(256, 1) attn_fwd_kernel(const T* __restrict__ qkv, long ld, T* __restrict__ ctx, long ldo,
                                                          float* __restrict__ lse, const int32_t* __restrict__ lens, int S,
                                                          int H, float scale) {
    __shared__ __attribute__((aligned(16))) unsigned char sK[32 * 512];
    __shared__ __attribute__((aligned(16))) unsigned char sV[32 * 512];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, fl = lane & 31, h2 = lane >> 5;
    const int qt = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    const int len = lens ? min(lens[b], S) : S;
    const int q0 = qt * 128;
    const size_t rowbase = (size_t)b * S;
    T* out = ctx + rowbase * ldo + h * DK;
    float* lse_o = lse + ((size_t)b * H + h) * S;
    if (q0 >= len) {  // fully padded query tile: zeros (the rows are masked to zero downstream anyway)
        for (int i = tid; i < 128 * 32; i += 256) {
            int r = i >> 5, c = (i & 31) * 4;
            if (q0 + r < S) st4<T>(out + (size_t)(q0 + r) * ldo + c, make_float4(0.f, 0.f, 0.f, 0.f));
        }
        if (tid < 128 && q0 + tid < S) lse_o[q0 + tid] = 0.f;
        return;
    }
    const T* Q = qkv + rowbase * ld + h * DK;
    const T* K = qkv + rowbase * ld + (size_t)H * DK + h * DK;
    const T* V = qkv + rowbase * ld + (size_t)2 * H * DK + h * DK;

    const int myq = q0 + w * 32 + fl;
    float qf[64];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (myq < S) v = ld4<T>(Q + (size_t)myq * ld + h2 * 64 + u * 4);
        qf[u * 4 + 0] = v.x * scale; qf[u * 4 + 1] = v.y * scale; qf[u * 4 + 2] = v.z * scale; qf[u * 4 + 3] = v.w * scale;
    }
    f32x16 o[4];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[nb][r] = 0.f;
    float m = -INFINITY, l = 0.f;

    for (int k0 = 0; k0 < len; k0 += 32) {
        __syncthreads();
        stage_tile<T, 256>(sK, K, ld, k0, min(32, S - k0), tid);
        stage_tile<T, 256>(sV, V, ld, k0, min(32, S - k0), tid);
        __syncthreads();
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            float4 kf = *reinterpret_cast<const float4*>(sK + swz(fl, h2 * 16 + u));
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.x, qf[u * 4 + 0], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.y, qf[u * 4 + 1], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.z, qf[u * 4 + 2], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.w, qf[u * 4 + 3], s, 0, 0, 0);
        }
        // s[r] = score(q = lane&31, key = k0 + crow(r,h2))
        float mx = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (k0 + crow(r, h2) >= len) s[r] = -INFINITY;
            mx = fmaxf(mx, s[r]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float mn = fmaxf(m, mx);
        float alpha = __expf(m - mn);   // m = -inf on the first tile -> 0
        float rs = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = __expf(s[r] - mn); rs += s[r]; }
        rs += __shfl_xor(rs, 32, 64);
        l = l * alpha + rs;
        m = mn;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float ar = __shfl(alpha, crow(r, h2), 64);
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) o[nb][r] *= ar;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            int key = crow(r, h2);
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) {
                int col = nb * 32 + fl;
                float vb = *reinterpret_cast<const float*>(sV + swz(key, col >> 2) + (col & 3) * 4);
                o[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(s[r], vb, o[nb], 0, 0, 0);
            }
        }
    }
    float linv = l > 0.f ? 1.f / l : 0.f;
    if (h2 == 0 && myq < S) lse_o[myq] = (l > 0.f) ? m + __logf(l) : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float li = __shfl(linv, crow(r, h2), 64);
        int q = q0 + w * 32 + crow(r, h2);
        if (q < S) {
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) Elem<T>::st(out + (size_t)q * ldo + nb * 32 + fl, o[nb][r] * li);
        }
    }
}


// ====================================================================================================
// bf16 path: v_mfma_f32_32x32x16_bf16 (2.5 PF roof), same register-resident-P structure as the f32 kernels.
// Tiles are [rows][128] bf16 (256-B rows) with ONE swizzle that serves both access patterns:
//   16-B chunk c of row r lives at chunk c ^ S(r),  S(r) = ((r&3)<<2) | ((r>>2)&3)
//   * ds_read_b128 of one chunk by 16 distinct rows (K-contiguous operand)      -> 16 distinct slots
//   * ds_read_b64_tr_b16 of 4 consecutive rows x 64 B (row-major operand, transposing) -> 4 distinct bank quarters
// k-slot convention of every MFMA here: lane half h = lane>>5, element e = 0..7.
//   K-contiguous operands: d = 16*s + 8*h + e.
//   row-indexed operands (P / dS as A, V / dO / Q / K as B): row = 16*u + 4*h + (e&3) + 8*(e>>2)  — exactly the rows
//   a lane owns in the 32x32 C layout, so probabilities feed the next MFMA straight from registers.
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) s16x4* lds_s4p;

__device__ __forceinline__ int swzb(int row, int c16) { return row * 256 + ((c16 ^ (((row & 3) << 2) | ((row >> 2) & 3))) << 4); }

template <int NT, int ROWS>
__device__ __forceinline__ void stage_tile_bf16(unsigned char* lds, const bf16_t* __restrict__ base, long ld, int row0,
                                                int nrows_valid, int tid) {
#pragma unroll
    for (int i = 0; i < ROWS * 16 / NT; ++i) {
        int idx = tid + NT * i;
        int r = idx >> 4, c = idx & 15;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (r < nrows_valid) v = *reinterpret_cast<const uint4*>(base + (size_t)(row0 + r) * ld + c * 8);
        *reinterpret_cast<uint4*>(lds + swzb(r, c)) = v;
    }
}
// K-contiguous fragment: row `row`, d-slots 16*s + 8*h .. +7
__device__ __forceinline__ bf16x8 frag_k(const unsigned char* lds, int row, int s, int h) {
    return __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(lds + swzb(row, 2 * s + h)));
}
// transposed fragment: rows rowbase + 4h + {0..3, 8..11}, column col (= 32*nb + lane&31)
__device__ __forceinline__ bf16x8 frag_t(const unsigned char* lds, int rowbase, int nb, int lane) {
    int li = lane & 15, g = lane >> 4, h = g >> 1;
    int row = rowbase + 4 * h + (li >> 2);
    int col = nb * 32 + 16 * (g & 1) + 4 * (li & 3);       // 8-byte piece inside 16-B chunk col>>3
    int off0 = swzb(row, col >> 3) + ((col & 7) << 1);
    int off1 = swzb(row + 8, col >> 3) + ((col & 7) << 1);
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4p)(lds + off0));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4p)(lds + off1));
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}
__device__ __forceinline__ bf16x8 pack8(const float* p) {
    uint4 v;
    v.x = pack_bf16x2(p[0], p[1]); v.y = pack_bf16x2(p[2], p[3]); v.z = pack_bf16x2(p[4], p[5]); v.w = pack_bf16x2(p[6], p[7]);
    return __builtin_bit_cast(bf16x8, v);
}
// load this lane's K-contiguous register fragments of one row (8 steps x 8 bf16), optionally scaled
__device__ __forceinline__ void load_row_frags(bf16x8* f, const bf16_t* row, int h, bool ok) {
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        uint4 v = make_uint4(0, 0, 0, 0);
        if (ok) v = *reinterpret_cast<const uint4*>(row + 16 * s + 8 * h);
        f[s] = __builtin_bit_cast(bf16x8, v);
    }
}

// Register-staged prefetch of a [64 rows][128] bf16 tile (256 threads: 4 x 16 B per thread).  Rows are CLAMPED to the
// sequence's last row instead of zero-filled: every consumer already multiplies rows >= len by an exact 0 probability /
// score gradient, so only finiteness matters, and the load needs no predicate (r01h: the synchronous
// load -> ds_write -> barrier staging left the MFMA pipe idle for the whole L2 round trip of every tile).
#define TILE_LD1(t, base, ld_, row0, last_row, I) \
    (t##I) = *reinterpret_cast<const uint4*>((base) + (size_t)min((row0) + ((tid + 256 * I) >> 4), (last_row)) * (ld_) + (tid & 15) * 8)
#define TILE_LOAD_REGS(t, base, ld_, row0, last_row) do { TILE_LD1(t, base, ld_, row0, last_row, 0); TILE_LD1(t, base, ld_, row0, last_row, 1); \
    TILE_LD1(t, base, ld_, row0, last_row, 2); TILE_LD1(t, base, ld_, row0, last_row, 3); } while (0)
#define TILE_ST1(lds, t, I) *reinterpret_cast<uint4*>((lds) + swzb((tid + 256 * I) >> 4, tid & 15)) = (t##I)
#define TILE_STORE_REGS(lds, t) do { TILE_ST1(lds, t, 0); TILE_ST1(lds, t, 1); TILE_ST1(lds, t, 2); TILE_ST1(lds, t, 3); } while (0)

__global__ void __launch_bounds__(256, 2) attn_fwd_bf16_kernel(const bf16_t* __restrict__ qkv, long ld, bf16_t* __restrict__ ctx,
                                                               long ldo, float* __restrict__ lse,
                                                               const int32_t* __restrict__ lens, int S, int H, float scale, int abl) {
    __shared__ __attribute__((aligned(16))) unsigned char sKV[2][2][64 * 256];   // [buffer][K | V]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, fl = lane & 31, h2 = lane >> 5;
    const int qt = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    const int len = lens ? min(lens[b], S) : S;
    const int q0 = qt * 128;
    const size_t rowbase = (size_t)b * S;
    bf16_t* out = ctx + rowbase * ldo + h * DK;
    float* lse_o = lse + ((size_t)b * H + h) * S;
    if (q0 >= len) {
        for (int i = tid; i < 128 * 32; i += 256) {
            int r = i >> 5, c = (i & 31) * 4;
            if (q0 + r < S) st4<bf16_t>(out + (size_t)(q0 + r) * ldo + c, make_float4(0.f, 0.f, 0.f, 0.f));
        }
        if (tid < 128 && q0 + tid < S) lse_o[q0 + tid] = 0.f;
        return;
    }
    const bf16_t* Q = qkv + rowbase * ld + h * DK;
    const bf16_t* K = qkv + rowbase * ld + (size_t)H * DK + h * DK;
    const bf16_t* V = qkv + rowbase * ld + (size_t)2 * H * DK + h * DK;
    const int myq = q0 + w * 32 + fl;
    bf16x8 qf[8];
    load_row_frags(qf, Q + (size_t)min(myq, S - 1) * ld, h2, myq < S);
    f32x16 o[4];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[nb][r] = 0.f;
    float m = -INFINITY, l = 0.f;                        // m: running maximum of s * scale * log2(e)
    const float sc2 = scale * 1.4426950408889634f;

    uint4 tk0, tk1, tk2, tk3, tv0, tv1, tv2, tv3;
    TILE_LOAD_REGS(tk, K, ld, 0, S - 1);
    TILE_LOAD_REGS(tv, V, ld, 0, S - 1);
    TILE_STORE_REGS(sKV[0][0], tk);
    TILE_STORE_REGS(sKV[0][1], tv);
    __syncthreads();
    int buf = 0;
    for (int k0 = 0; k0 < len; k0 += 64, buf ^= 1) {
        const bool more = (k0 + 64 < len) && !abl;
        if (more) {                                       // next tile travels while this one is multiplied
            TILE_LOAD_REGS(tk, K, ld, k0 + 64, S - 1);
            TILE_LOAD_REGS(tv, V, ld, k0 + 64, S - 1);
        }
        const unsigned char* sK = sKV[buf][0];
        const unsigned char* sV = sKV[buf][1];
        // software pipeline over the 8 d-slices: the two K fragments of slice st+1 are in flight while slice st is
        // multiplied (two independent accumulator chains); sched_barrier pins the order - unpinned, the scheduler sinks
        // each read to just above its MFMA (r01i ISA: ds_read / s_waitcnt lgkmcnt(0) / v_mfma, 16 times in a row)
        f32x16 s[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) s[kb][r] = 0.f;
        {
            bf16x8 c0 = frag_k(sK, fl, 0, h2), c1 = frag_k(sK, 32 + fl, 0, h2);
#pragma unroll
            for (int st = 0; st < 8; ++st) {
                bf16x8 n0 = c0, n1 = c1;
                if (st < 7) { n0 = frag_k(sK, fl, st + 1, h2); n1 = frag_k(sK, 32 + fl, st + 1, h2); }
                __builtin_amdgcn_sched_barrier(0);
                s[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(c0, qf[st], s[0], 0, 0, 0);
                s[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(c1, qf[st], s[1], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                c0 = n0; c1 = n1;
            }
        }
        // Softmax bookkeeping in the log2 domain (p = exp2(s * scale * log2 e - m): one fma + one v_exp_f32 per score), the
        // key-padding mask only on the sequence's last tile, and a LAZY running maximum: the accumulators are rescaled only
        // when some query's maximum grew by more than 2^8 - after the first tiles that is almost never, and a rescale costs
        // 16 cross-lane permutes + 64 multiplies per wave and tile (r02s PMC: 15.7 VALU instructions per MFMA made this kernel
        // VALU-bound 2:1).  A stale maximum only scales p and l by the same factor <= 2^8: o / l is unchanged.
        if (k0 + 64 > len) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (k0 + kb * 32 + crow(r, h2) >= len) s[kb][r] = -INFINITY;
        }
        float mx = -INFINITY;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[kb][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64)) * sc2;                       // scale > 0: the maximum of the raw scores, scaled once
        if (__builtin_amdgcn_ballot_w64(mx > m + 8.f) != 0ull) {           // wave-uniform; always taken on the first tile (m = -inf)
            const float mn = fmaxf(m, mx);
            const float alpha = __builtin_amdgcn_exp2f(m - mn);
            l *= alpha;
            m = mn;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float ar = __shfl(alpha, crow(r, h2), 64);
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) o[nb][r] *= ar;
            }
        }
        float rs = 0.f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[kb][r] = __builtin_amdgcn_exp2f(fmaf(s[kb][r], sc2, -m)); rs += s[kb][r]; }
        rs += __shfl_xor(rs, 32, 64);
        l += rs;
        {   // P V: the four V fragments of the next 16-key group are fetched while the current group is multiplied
            bf16x8 cv[4], nv[4];
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) cv[nb] = frag_t(sV, 0, nb, lane);
#pragma unroll
            for (int g = 0; g < 4; ++g) {                 // g = 2*kb + u: keys 16g .. 16g+15 of the tile
                if (g < 3) {
#pragma unroll
                    for (int nb = 0; nb < 4; ++nb) nv[nb] = frag_t(sV, 16 * (g + 1), nb, lane);
                }
                float pv[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) pv[e] = s[g >> 1][8 * (g & 1) + e];
                bf16x8 pa = pack8(pv);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) o[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa, cv[nb], o[nb], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) cv[nb] = nv[nb];
            }
        }
        if (more) {
            TILE_STORE_REGS(sKV[buf ^ 1][0], tk);
            TILE_STORE_REGS(sKV[buf ^ 1][1], tv);
        }
        __syncthreads();
    }
    float linv = l > 0.f ? 1.f / l : 0.f;
    if (h2 == 0 && myq < S) lse_o[myq] = (l > 0.f) ? m * 0.6931471805599453f + __logf(l) : 0.f;      // natural-log lse for the backward
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float li = __shfl(linv, crow(r, h2), 64);
        int q = q0 + w * 32 + crow(r, h2);
        if (q < S) {
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) out[(size_t)q * ldo + nb * 32 + fl] = f32_to_bf16(o[nb][r] * li);
        }
    }
}

// dK, dV: one wave owns 32 keys (K, V fragments in registers); the block streams 64-query tiles of Q / dO.
// Per tile the log-sum-exp and delta values of its 64 queries travel with the Q / dO prefetch into LDS: the first
// version fetched them from GLOBAL memory element by element inside the `key_ok && q < len` branch - 32 dependent,
// individually waited loads per 32-query block (r01i ISA: global_load_dword / s_waitcnt vmcnt(0) / v_exp, x32), which is
// why dK/dV ran at 323 TF next to dQ's 724 TF.  The probability / score-gradient math is branch-free (invalid pairs are
// multiplied by an exact 0), and the fragment reads of each MFMA run are issued together in front of it (sched_barrier
// keeps the scheduler from sinking them back to their consumers).
__global__ void __launch_bounds__(256, 1) attn_bwd_dkv_bf16_kernel(const bf16_t* __restrict__ qkv, long ld,
                                                                   const bf16_t* __restrict__ dctx, long ldo,
                                                                   const float* __restrict__ lse, const float* __restrict__ delta,
                                                                   bf16_t* __restrict__ dqkv, const int32_t* __restrict__ lens,
                                                                   int S, int H, float scale) {
    __shared__ __attribute__((aligned(16))) unsigned char sQD[2][2][64 * 256];   // [buffer][Q | dO]
    __shared__ __attribute__((aligned(16))) float sLD[2][2][64];                 // [buffer][lse | delta] of the tile's queries
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, fl = lane & 31, h2 = lane >> 5;
    const int kt = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    const int len = lens ? min(lens[b], S) : S;
    const size_t rowbase = (size_t)b * S;
    const int kbase = kt * 128 + w * 32;
    bf16_t* dK = dqkv + rowbase * ld + (size_t)H * DK + h * DK;
    bf16_t* dV = dqkv + rowbase * ld + (size_t)2 * H * DK + h * DK;
    if (kt * 128 >= len) {
        for (int i = tid; i < 128 * 32; i += 256) {
            int r = i >> 5, c = (i & 31) * 4;
            if (kt * 128 + r < S) {
                st4<bf16_t>(dK + (size_t)(kt * 128 + r) * ld + c, make_float4(0.f, 0.f, 0.f, 0.f));
                st4<bf16_t>(dV + (size_t)(kt * 128 + r) * ld + c, make_float4(0.f, 0.f, 0.f, 0.f));
            }
        }
        return;
    }
    const bf16_t* Q = qkv + rowbase * ld + h * DK;
    const bf16_t* K = qkv + rowbase * ld + (size_t)H * DK + h * DK;
    const bf16_t* V = qkv + rowbase * ld + (size_t)2 * H * DK + h * DK;
    const bf16_t* dO = dctx + rowbase * ldo + h * DK;
    const float* lse_b = lse + ((size_t)b * H + h) * S;
    const float* del_b = delta + ((size_t)b * H + h) * S;
    const int mykey = kbase + fl;
    const float key_ok = mykey < len ? 1.f : 0.f;
    bf16x8 kf[8], vf[8];
    load_row_frags(kf, K + (size_t)min(mykey, S - 1) * ld, h2, mykey < S);
    load_row_frags(vf, V + (size_t)min(mykey, S - 1) * ld, h2, mykey < S);
    f32x16 dk[4], dv[4];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dk[nb][r] = 0.f; dv[nb][r] = 0.f; }

    // threads 0..63 carry lse, 64..127 delta of query (tile start + tid & 63); a query >= len gets lse = +inf -> p = 0
    const float* ld_src = tid < 64 ? lse_b : del_b;
    // (lse travels in the log2 domain: p = exp2(s * scale * log2 e - lse * log2 e) is one fma + one v_exp_f32 per score)
    const float sc2 = scale * 1.4426950408889634f;
    auto load_ld = [&](int q0) -> float {
        int q = q0 + (tid & 63);
        float v = ld_src[min(q, S - 1)];
        if (tid < 64) v *= 1.4426950408889634f;
        if (q >= len) v = tid < 64 ? INFINITY : 0.f;
        return v;
    };

    uint4 tq0, tq1, tq2, tq3, td0, td1, td2, td3;
    float tl = 0.f;
    TILE_LOAD_REGS(tq, Q, ld, 0, S - 1);
    TILE_LOAD_REGS(td, dO, ldo, 0, S - 1);
    if (tid < 128) tl = load_ld(0);
    TILE_STORE_REGS(sQD[0][0], tq);
    TILE_STORE_REGS(sQD[0][1], td);
    if (tid < 128) sLD[0][tid >> 6][tid & 63] = tl;
    __syncthreads();
    int buf = 0;
    for (int q0 = 0; q0 < len; q0 += 64, buf ^= 1) {
        const bool more = q0 + 64 < len;
        if (more) {
            TILE_LOAD_REGS(tq, Q, ld, q0 + 64, S - 1);
            TILE_LOAD_REGS(td, dO, ldo, q0 + 64, S - 1);
            if (tid < 128) tl = load_ld(q0 + 64);
        }
        const unsigned char* sQ = sQD[buf][0];
        const unsigned char* sdO = sQD[buf][1];
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            if (q0 + qb * 32 >= len) break;
            // ---- S^T = Q K^T and dP^T = dO V^T for 32 queries x my 32 keys: 16 fragment reads, then 16 MFMAs
            bf16x8 fq[8], fd[8];
#pragma unroll
            for (int st = 0; st < 8; ++st) { fq[st] = frag_k(sQ, qb * 32 + fl, st, h2); fd[st] = frag_k(sdO, qb * 32 + fl, st, h2); }
            // this lane's 16 queries: rows crow(r, h2) = (r&3) + 8*(r>>2) + 4*h2 -> four float4 per array
            float4 l4[4], d4[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                l4[g] = *reinterpret_cast<const float4*>(&sLD[buf][0][qb * 32 + 8 * g + 4 * h2]);
                d4[g] = *reinterpret_cast<const float4*>(&sLD[buf][1][qb * 32 + 8 * g + 4 * h2]);
            }
            __builtin_amdgcn_sched_barrier(0);
            f32x16 s, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
            for (int st = 0; st < 8; ++st) {
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fq[st], kf[st], s, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fd[st], vf[st], dp, 0, 0, 0);
            }
            // ---- transposed Q / dO fragments of the second product pair, issued before the exponentials
            bf16x8 tO[2][4], tQ[2][4];
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) {
                    tO[u][nb] = frag_t(sdO, qb * 32 + 16 * u, nb, lane);
                    tQ[u][nb] = frag_t(sQ, qb * 32 + 16 * u, nb, lane);
                }
            __builtin_amdgcn_sched_barrier(0);
            float pv[16], dsv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float lq = reinterpret_cast<const float*>(&l4[r >> 2])[r & 3];
                const float dq_ = reinterpret_cast<const float*>(&d4[r >> 2])[r & 3];
                const float p = __builtin_amdgcn_exp2f(fmaf(s[r], sc2, -lq)) * key_ok;   // lq = +inf for q >= len -> exp2(-inf) = 0
                pv[r] = p;
                dsv[r] = p * (dp[r] - dq_);                                  // (the softmax scale multiplies dK once, at the store)
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                bf16x8 pa = pack8(pv + 8 * u), da = pack8(dsv + 8 * u);
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) {
                    dv[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa, tO[u][nb], dv[nb], 0, 0, 0);
                    dk[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(da, tQ[u][nb], dk[nb], 0, 0, 0);
                }
            }
        }
        if (more) {
            TILE_STORE_REGS(sQD[buf ^ 1][0], tq);
            TILE_STORE_REGS(sQD[buf ^ 1][1], td);
            if (tid < 128) sLD[buf ^ 1][tid >> 6][tid & 63] = tl;
        }
        __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        int key = kbase + crow(r, h2);
        if (key < S) {
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) {
                dK[(size_t)key * ld + nb * 32 + fl] = f32_to_bf16(dk[nb][r] * scale);
                dV[(size_t)key * ld + nb * 32 + fl] = f32_to_bf16(dv[nb][r]);
            }
        }
    }
}

// (also computes delta[q] = sum_d dO[q][d] O[q][d] for its queries - a lane already holds half of its query's dO row - and
// writes it for the dK/dV kernel, which therefore runs AFTER this one: the separate delta launch, 12 us per layer, is gone)
__global__ void __launch_bounds__(256, 2) attn_bwd_dq_bf16_kernel(const bf16_t* __restrict__ qkv, long ld,
                                                                  const bf16_t* __restrict__ ctx,
                                                                  const bf16_t* __restrict__ dctx, long ldo,
                                                                  const float* __restrict__ lse, float* __restrict__ delta,
                                                                  bf16_t* __restrict__ dqkv, const int32_t* __restrict__ lens,
                                                                  int S, int H, float scale) {
    __shared__ __attribute__((aligned(16))) unsigned char sKV[2][2][64 * 256];   // [buffer][K | V]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, fl = lane & 31, h2 = lane >> 5;
    const int qt = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    const int len = lens ? min(lens[b], S) : S;
    const int q0 = qt * 128;
    const size_t rowbase = (size_t)b * S;
    bf16_t* dQ = dqkv + rowbase * ld + h * DK;
    if (q0 >= len) {
        for (int i = tid; i < 128 * 32; i += 256) {
            int r = i >> 5, c = (i & 31) * 4;
            if (q0 + r < S) st4<bf16_t>(dQ + (size_t)(q0 + r) * ld + c, make_float4(0.f, 0.f, 0.f, 0.f));
        }
        if (tid < 128 && q0 + tid < S) delta[((size_t)b * H + h) * S + q0 + tid] = 0.f;      // (never read: queries >= len)
        return;
    }
    const bf16_t* Q = qkv + rowbase * ld + h * DK;
    const bf16_t* K = qkv + rowbase * ld + (size_t)H * DK + h * DK;
    const bf16_t* V = qkv + rowbase * ld + (size_t)2 * H * DK + h * DK;
    const bf16_t* dO = dctx + rowbase * ldo + h * DK;
    const int myq = q0 + w * 32 + fl;
    const bool q_ok = myq < len;
    float my_lse = 0.f, my_del = 0.f;
    if (myq < S) my_lse = lse[((size_t)b * H + h) * S + myq];
    // log2 domain (one fma + one v_exp_f32 per score); a padded query gets lse = +inf -> p = 0 without a mask multiply
    const float sc2 = scale * 1.4426950408889634f;
    const float my_lse2 = q_ok ? my_lse * 1.4426950408889634f : INFINITY;
    bf16x8 qf[8], df[8];
    load_row_frags(qf, Q + (size_t)min(myq, S - 1) * ld, h2, myq < S);
    load_row_frags(df, dO + (size_t)min(myq, S - 1) * ldo, h2, myq < S);
    {   // delta of my query from the dO fragments just loaded (this lane: d = 16 s + 8 h2 + e; the other half-wave has the rest)
        bf16x8 of[8];
        load_row_frags(of, ctx + (rowbase + (size_t)min(myq, S - 1)) * ldo + h * DK, h2, myq < S);
        float acc = 0.f;
#pragma unroll
        for (int st = 0; st < 8; ++st) {
            const uint4 ov = __builtin_bit_cast(uint4, of[st]), dv_ = __builtin_bit_cast(uint4, df[st]);
            const uint32_t* ou = reinterpret_cast<const uint32_t*>(&ov);
            const uint32_t* du = reinterpret_cast<const uint32_t*>(&dv_);
#pragma unroll
            for (int e = 0; e < 4; ++e)
                acc += __uint_as_float(ou[e] << 16) * __uint_as_float(du[e] << 16) + __uint_as_float(ou[e] & 0xffff0000u) * __uint_as_float(du[e] & 0xffff0000u);
        }
        acc += __shfl_xor(acc, 32, 64);
        my_del = acc;                                        // (rows >= S load zeros)
        if (h2 == 0 && myq < S) delta[((size_t)b * H + h) * S + myq] = my_del;
    }
    f32x16 dq[4];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[nb][r] = 0.f;

    uint4 tk0, tk1, tk2, tk3, tv0, tv1, tv2, tv3;
    TILE_LOAD_REGS(tk, K, ld, 0, S - 1);
    TILE_LOAD_REGS(tv, V, ld, 0, S - 1);
    TILE_STORE_REGS(sKV[0][0], tk);
    TILE_STORE_REGS(sKV[0][1], tv);
    __syncthreads();
    int buf = 0;
    for (int k0 = 0; k0 < len; k0 += 64, buf ^= 1) {
        const bool more = k0 + 64 < len;
        if (more) {
            TILE_LOAD_REGS(tk, K, ld, k0 + 64, S - 1);
            TILE_LOAD_REGS(tv, V, ld, k0 + 64, S - 1);
        }
        const unsigned char* sK = sKV[buf][0];
        const unsigned char* sV = sKV[buf][1];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            if (k0 + kb * 32 >= len) break;
            // software pipeline over the 8 d-slices (K and V fragments of slice st+1 in flight while slice st is multiplied);
            // sched_barrier pins it (hoisting the 8 transposed K fragments too spills at 2 waves per SIMD)
            f32x16 s, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
            {
                bf16x8 ck = frag_k(sK, kb * 32 + fl, 0, h2), cv = frag_k(sV, kb * 32 + fl, 0, h2);
#pragma unroll
                for (int st = 0; st < 8; ++st) {
                    bf16x8 nk = ck, nv = cv;
                    if (st < 7) { nk = frag_k(sK, kb * 32 + fl, st + 1, h2); nv = frag_k(sV, kb * 32 + fl, st + 1, h2); }
                    __builtin_amdgcn_sched_barrier(0);
                    s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ck, qf[st], s, 0, 0, 0);
                    dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cv, df[st], dp, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    ck = nk; cv = nv;
                }
            }
            float dsv[16];
            const bool tail = k0 + kb * 32 + 32 > len;                     // block-uniform: the only key block with padded keys
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float p = __builtin_amdgcn_exp2f(fmaf(s[r], sc2, -my_lse2));
                if (tail && k0 + kb * 32 + crow(r, h2) >= len) p = 0.f;
                dsv[r] = p * (dp[r] - my_del);                                  // (the softmax scale multiplies dQ once, at the store)
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                bf16x8 da = pack8(dsv + 8 * u);
#pragma unroll
                for (int nb = 0; nb < 4; ++nb)
                    dq[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(da, frag_t(sK, kb * 32 + 16 * u, nb, lane), dq[nb], 0, 0, 0);
            }
        }
        if (more) {
            TILE_STORE_REGS(sKV[buf ^ 1][0], tk);
            TILE_STORE_REGS(sKV[buf ^ 1][1], tv);
        }
        __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        int q = q0 + w * 32 + crow(r, h2);
        if (q < S) {
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) dQ[(size_t)q * ld + nb * 32 + fl] = f32_to_bf16(dq[nb][r] * scale);
        }
    }
}

// ================================================================== ping-pong kernels (bf16, S >= 256)
// r03m: the kernels above spend ~1000 MFMA cycles and ~1500 VALU cycles per wave and key tile, and the two co-resident waves of a
// SIMD (from two independent workgroups, each barrier-synchronised with its own three siblings) do not overlap them: a tile step
// takes about the SUM (no-load ablation: 53 of 66 us remain when the in-loop tile loads are removed - it is not the staging).
// Here a workgroup is 8 waves = one 256-query block, the two waves of a SIMD belong to the SAME workgroup and run a tile step in
// OPPOSITE phases, held there by one workgroup barrier per phase: while wave w multiplies (P V of the previous tile, then Q K^T
// of the current one: 32 MFMAs) wave w+4 runs the current tile's softmax on the VALU, then they swap.  K/V tiles arrive by
// LDS-DMA (swizzle on the source address) into a 4-slot ring two tiles ahead of their first use: no prefetch registers, no
// ds_write pass, no address arithmetic in the loop beyond one clamp per piece.
typedef unsigned fs2_u32x2 __attribute__((ext_vector_type(2)));
static constexpr int AR_D = 4, AR_TILE = 64 * 256, AR_SLOT = 2 * AR_TILE;
template <int N> __device__ __forceinline__ void ar_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }
__device__ __forceinline__ void ar_glds16(unsigned voff, const void* sbase, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ unsigned ar_lds_addr(const void* p) { return (unsigned)(size_t)(__attribute__((address_space(3))) const void*)p; }

// the per-wave state of the forward kernel and its three pieces of work; G (0 | 1) = the phase group: group 0 multiplies in the
// even barrier phases and runs the softmax in the odd ones, group 1 the other way round
struct FwdPP {
    bf16x8 qf[8], pa[4];
    f32x16 o[4], s[2];
    float m, l;
};
// The multiply phase of one tile step: O^T += V(t-1)^T P(t-1)^T (16 MFMAs; the accumulators are kept TRANSPOSED - lane = query,
// registers = head columns - so that the running-maximum rescale and the final 1/l are per-lane multiplies and the epilogue
// stores 16-byte row pieces), then S^T(t) = K(t) Q^T (16 MFMAs).  One LDS fragment feeds one MFMA.
// r03m, measured on the first two builds of this kernel: a wave issues about one instruction per 5 cycles, so everything
// between two MFMAs (32 cycles) has to fit in ~5 issues.  With swizzled addresses formed per read (~10 VALU each) a 16-MFMA
// product took ~1400 cycles whatever the read look-ahead (one step or a window of six).  Hence: the sixteen lane-dependent
// address parts live in registers as ABSOLUTE LDS addresses of the current ring slot (aK: K fragment of d-slice st; aV: the two
// transposing reads of head-column block nb), every other term is an instruction immediate, and the whole set moves to the next
// slot with one add per register and tile.
typedef const __attribute__((address_space(3))) uint4* lds_u4p;
template <bool PV, bool QK>
__device__ __forceinline__ void fwd_pp_mfma(FwdPP& z, const unsigned (&aV)[8], const unsigned (&aK)[8]) {
    constexpr int W = 8, N0 = PV ? 16 : 0, N = N0 + (QK ? 16 : 0);
    bf16x8 f[N];
    auto load = [&](int i) -> bf16x8 {
        if (i < N0) {
            const int gk = i >> 2, nb = i & 3;
            s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4p)(size_t)(aV[2 * nb] + gk * 4096));
            s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4p)(size_t)(aV[2 * nb + 1] + gk * 4096));
            return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
        }
        const int j = i - N0;
        return __builtin_bit_cast(bf16x8, *(lds_u4p)(size_t)(aK[j >> 1] + (j & 1) * 8192));
    };
    if (QK) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) z.s[kb][r] = 0.f;
    }
#pragma unroll
    for (int i = 0; i < W; ++i) f[i] = load(i);
#pragma unroll
    for (int i = 0; i < N; ++i) {
        __builtin_amdgcn_sched_barrier(0);
        if (i < N0) z.o[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[i], z.pa[i >> 2], z.o[i & 3], 0, 0, 0);
        else z.s[(i - N0) & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[i], z.qf[(i - N0) >> 1], z.s[(i - N0) & 1], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (i + W < N) f[i + W] = load(i + W);
    }
}
// softmax of one 64-key tile in the log2 domain with the lazy running maximum of attn_fwd_bf16_kernel; leaves P packed in z.pa
__device__ __forceinline__ void fwd_pp_softmax(FwdPP& z, int k0, int len, float sc2, int h2) {
    if (k0 + 64 > len) {                                     // (one per-lane threshold against compile-time row constants)
        int thr = len - k0 - 4 * h2;
        asm volatile("" : "+v"(thr));
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (kb * 32 + (r & 3) + 8 * (r >> 2) >= thr) z.s[kb][r] = -INFINITY;
    }
    float mx = -INFINITY;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, z.s[kb][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64)) * sc2;
    if (__builtin_amdgcn_ballot_w64(mx > z.m + 8.f) != 0ull) {
        const float mn = fmaxf(z.m, mx);
        const float alpha = __builtin_amdgcn_exp2f(z.m - mn);
        z.l *= alpha;
        z.m = mn;
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) z.o[nb][r] *= alpha;  // (transposed accumulators: this lane's query)
    }
    float rs0 = 0.f, rs1 = 0.f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            z.s[kb][r] = __builtin_amdgcn_exp2f(fmaf(z.s[kb][r], sc2, -z.m));
            z.s[kb][r + 1] = __builtin_amdgcn_exp2f(fmaf(z.s[kb][r + 1], sc2, -z.m));
            rs0 += z.s[kb][r]; rs1 += z.s[kb][r + 1];
        }
    float rs = rs0 + rs1;
    rs += __shfl_xor(rs, 32, 64);
    z.l += rs;
#pragma unroll
    for (int gk = 0; gk < 4; ++gk) {
        float pv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) pv[e] = z.s[gk >> 1][8 * (gk & 1) + e];
        z.pa[gk] = pack8(pv);
    }
}

__global__ void __launch_bounds__(512) attn_fwd_pp_kernel(const bf16_t* __restrict__ qkv, long ld, bf16_t* __restrict__ ctx, long ldo,
                                                          float* __restrict__ lse, const int32_t* __restrict__ lens, int S, int H,
                                                          float scale, unsigned long long* dbg) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];          // AR_D x (K tile | V tile)
    const int tid = threadIdx.x, lane = tid & 63, fl = lane & 31, h2 = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = w >> 2;                             // phase group: waves w and w + 4 sit on the same SIMD
    const int rb = 2 * (w & 3) + g;               // the wave's 32-query block: both groups get blocks of a partial tile
    const int qt = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    const int len = lens ? min(lens[b], S) : S;
    const int q0 = qt * 256;
    const size_t rowbase = (size_t)b * S;
    bf16_t* out = ctx + rowbase * ldo + h * DK;
    float* lse_o = lse + ((size_t)b * H + h) * S;
    if (q0 >= len) {
        for (int i = tid; i < 256 * 32; i += 512) {
            int r = i >> 5, c = (i & 31) * 4;
            if (q0 + r < S) st4<bf16_t>(out + (size_t)(q0 + r) * ldo + c, make_float4(0.f, 0.f, 0.f, 0.f));
        }
        if (tid < 256 && q0 + tid < S) lse_o[q0 + tid] = 0.f;
        return;
    }
    const bf16_t* Q = qkv + rowbase * ld + h * DK;
    const unsigned char* Kb = reinterpret_cast<const unsigned char*>(qkv + rowbase * ld + (size_t)H * DK + h * DK);
    const unsigned char* Vb = reinterpret_cast<const unsigned char*>(qkv + rowbase * ld + (size_t)2 * H * DK + h * DK);
#ifdef FS2_DEV          // per-wave time stamps of workgroup (0, 0, 0) for tools/bench_attn.py (the product build has none of it)
#define AT_STAMP(i) do { if (dbg && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && lane == 0 && (i) < 64) dbg[w * 64 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define AT_STAMP(i) do { } while (0)
#endif
    AT_STAMP(0);
    const int nt = (len + 63) >> 6;
    const unsigned smem_u = ar_lds_addr(smem);
    // ---- this wave's DMA pieces: tile rows 8w .. 8w+7 of K and of V (a piece = 4 rows x 256 B, lane-linear in LDS)
    const unsigned ldb = (unsigned)(ld * 2);
    const int prow = 8 * w + (lane >> 4);
    const unsigned pcol0 = (unsigned)(((lane & 15) ^ (((prow & 3) << 2) | ((prow >> 2) & 3))) << 4);
    const unsigned pcol1 = (unsigned)(((lane & 15) ^ ((((prow + 4) & 3) << 2) | (((prow + 4) >> 2) & 3))) << 4);
    auto issue = [&](int t) {
        const unsigned dst = smem_u + (unsigned)((t & (AR_D - 1)) * AR_SLOT + 2 * w * 1024);
        const unsigned v0 = (unsigned)min(t * 64 + prow, S - 1) * ldb + pcol0;
        const unsigned v1 = (unsigned)min(t * 64 + prow + 4, S - 1) * ldb + pcol1;
        ar_glds16(v0, Kb, dst);
        ar_glds16(v0, Vb, dst + AR_TILE);
        ar_glds16(v1, Kb, dst + 1024);
        ar_glds16(v1, Vb, dst + AR_TILE + 1024);
    };
    issue(0);
    if (nt > 1) issue(1);

    const int myq = q0 + rb * 32 + fl;
    const bool active = q0 + rb * 32 < len;                 // wave-uniform
    FwdPP z;
    load_row_frags(z.qf, Q + (size_t)min(myq, S - 1) * ld, h2, myq < S);
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) z.o[nb][r] = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) z.pa[i] = __builtin_bit_cast(bf16x8, make_uint4(0, 0, 0, 0));
    z.m = -INFINITY; z.l = 0.f;
    const float sc2 = scale * 1.4426950408889634f;

    // even barrier phase 2t: tile t's pieces have landed (counted wait before the barrier), tile t+2 is issued behind it into the
    // slot whose last reader (group 1's P V of tile t-2, phase 2t-1) is done
    auto even_sync = [&](int t) {
        if (t < nt) { if (t + 1 < nt) ar_wait_vm<4>(); else ar_wait_vm<0>(); }
#ifdef FS2_DEV
        if (dbg && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && lane == 0 && t < 64) dbg[512 + w * 64 + t] = __builtin_amdgcn_s_memtime();
#endif
        __builtin_amdgcn_s_barrier();
        if (t + 2 < nt) issue(t + 2);
    };
    // fragment addresses (see fwd_pp_mfma): aK for the K tile of slot 0, aV for the V tile of slot 2 (= "tile -2": stepped BEFORE use); both step one
    // slot per multiply phase
    unsigned aK[8], aV[8];
    {
        const int li = lane & 15, gq = lane >> 4, hh = gq >> 1;
#pragma unroll
        for (int st = 0; st < 8; ++st) aK[st] = smem_u + (unsigned)swzb(fl, 2 * st + h2);
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
            const int row = 4 * hh + (li >> 2), col = nb * 32 + 16 * (gq & 1) + 4 * (li & 3);
            aV[2 * nb] = smem_u + (unsigned)(2 * AR_SLOT + AR_TILE + swzb(row, col >> 3) + ((col & 7) << 1));
            aV[2 * nb + 1] = smem_u + (unsigned)(2 * AR_SLOT + AR_TILE + swzb(row + 8, col >> 3) + ((col & 7) << 1));
        }
    }
    auto step = [&](unsigned (&a)[8], int from_slot) {       // slot from_slot -> from_slot + 1 (mod AR_D)
        const int d = (from_slot & (AR_D - 1)) == AR_D - 1 ? -(AR_D - 1) * AR_SLOT : AR_SLOT;
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] += (unsigned)d;
    };
    // (the first and the last tile step are peeled so that the loop bodies are straight-line: with the three forms of the
    // multiply phase behind branches inside ONE loop the accumulators were written out of place and copied back every tile)
    if (!active) {
        for (int t = 0; t <= nt; ++t) { even_sync(t); __builtin_amdgcn_s_barrier(); }
        return;
    }
    AT_STAMP(1);
    // (MI355X_MICROARCH "two waves per SIMD", item 4: issue arbitration is priority, then AGE - the second-dispatched half loses
    // every contested slot unless it is given static priority once; measured here: its multiply phase took 2000 cycles next to
    // the older half's softmax, the older half's 1370 next to the younger's)
    if (g == 1) __builtin_amdgcn_s_setprio(1);
    if (g == 0) {
        even_sync(0);
        step(aV, -2); fwd_pp_mfma<false, true>(z, aV, aK); step(aK, 0);
        __builtin_amdgcn_s_barrier();
        fwd_pp_softmax(z, 0, len, sc2, h2);
        for (int t = 1; t < nt; ++t) {
            even_sync(t);
            AT_STAMP(4 * t - 2);
            step(aV, t - 2); fwd_pp_mfma<true, true>(z, aV, aK); step(aK, t);
            AT_STAMP(4 * t - 1);
            __builtin_amdgcn_s_barrier();
            AT_STAMP(4 * t);
            fwd_pp_softmax(z, t * 64, len, sc2, h2);
            AT_STAMP(4 * t + 1);
        }
        even_sync(nt);
        step(aV, nt - 2); fwd_pp_mfma<true, false>(z, aV, aK);
        __builtin_amdgcn_s_barrier();
    } else {
        even_sync(0);
        __builtin_amdgcn_s_barrier();
        step(aV, -2); fwd_pp_mfma<false, true>(z, aV, aK); step(aK, 0);
        for (int t = 1; t < nt; ++t) {
            even_sync(t);
            AT_STAMP(4 * t - 2);
            fwd_pp_softmax(z, (t - 1) * 64, len, sc2, h2);
            AT_STAMP(4 * t - 1);
            __builtin_amdgcn_s_barrier();
            AT_STAMP(4 * t);
            step(aV, t - 2); fwd_pp_mfma<true, true>(z, aV, aK); step(aK, t);
            AT_STAMP(4 * t + 1);
        }
        even_sync(nt);
        fwd_pp_softmax(z, (nt - 1) * 64, len, sc2, h2);
        __builtin_amdgcn_s_barrier();
        step(aV, nt - 2); fwd_pp_mfma<true, false>(z, aV, aK);
    }
    AT_STAMP(62);
    const float linv = z.l > 0.f ? 1.f / z.l : 0.f;
    if (h2 == 0 && myq < S) lse_o[myq] = (z.l > 0.f) ? z.m * 0.6931471805599453f + __logf(z.l) : 0.f;
    // lane (fl, h2) holds, for its query, the columns nb*32 + 8*(r>>2) + 4*h2 + (r&3): one v_permlane32_swap per register pair
    // gives it two 8-column runs per 32-column block (the epilogue of fs2_gemm_epi.h)
    bf16_t* orow = out + (size_t)myq * ldo;
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
        float c[2][8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            fs2_u32x2 s0 = __builtin_amdgcn_permlane32_swap(__float_as_uint(z.o[nb][e]), __float_as_uint(z.o[nb][4 + e]), false, false);
            fs2_u32x2 s1 = __builtin_amdgcn_permlane32_swap(__float_as_uint(z.o[nb][8 + e]), __float_as_uint(z.o[nb][12 + e]), false, false);
            c[0][e] = __uint_as_float(s0[0]); c[0][4 + e] = __uint_as_float(s0[1]);
            c[1][e] = __uint_as_float(s1[0]); c[1][4 + e] = __uint_as_float(s1[1]);
        }
        if (myq < S) {
#pragma unroll
            for (int ch = 0; ch < 2; ++ch) {
                uint4 ov;
                uint32_t* ou = reinterpret_cast<uint32_t*>(&ov);
#pragma unroll
                for (int e = 0; e < 4; ++e) ou[e] = pack_bf16x2(c[ch][2 * e] * linv, c[ch][2 * e + 1] * linv);
                *reinterpret_cast<uint4*>(orow + nb * 32 + ch * 16 + h2 * 8) = ov;
            }
        }
    }
    AT_STAMP(63);
}

extern "C" int fs2_attn_fwd(const void* qkv, void* ctx, float* lse, const int32_t* lens, int B, int S, int H, int dk,
                            float scale, int dtype, hipStream_t stream) {
    FS2_CHECK_ARG(qkv && ctx && lse, "attn_fwd: null pointer");
    FS2_CHECK_ARG(dk == DK, "attn_fwd: only d_k = 128 is supported (got %d)", dk);
    FS2_CHECK_ARG(B >= 0 && S > 0 && H > 0, "attn_fwd: bad shape");
    if (B == 0) return FS2_OK;
    dim3 grid(fs2_cdiv(S, 128), H, B);
    long ld = 3L * H * DK, ldo = (long)H * DK;
    if (dtype == FS2_F32) attn_fwd_kernel<float><<<grid, 256, 0, stream>>>((const float*)qkv, ld, (float*)ctx, ldo, lse, lens, S, H, scale);
    else if (dtype == FS2_BF16) {
        static const int abl = fs2_dev_env("FS2_ATTN_ABL", 0);
        static const int pp = fs2_dev_env("FS2_ATTN_PP", 1);
        if (pp && S >= 256) {
            const int dyn = AR_D * AR_SLOT;
            unsigned long long* dbgp = nullptr;
#ifdef FS2_DEV
            { const char* e = getenv("FS2_ATTN_DBG_PTR"); if (e) dbgp = (unsigned long long*)strtoull(e, nullptr, 0); }
#endif
            static Fs2DevOnce once;
            once.run([&] { (void)hipFuncSetAttribute((const void*)attn_fwd_pp_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, dyn); });
            attn_fwd_pp_kernel<<<dim3(fs2_cdiv(S, 256), H, B), 512, dyn, stream>>>((const bf16_t*)qkv, ld, (bf16_t*)ctx, ldo, lse, lens, S, H, scale, dbgp);
        } else attn_fwd_bf16_kernel<<<grid, 256, 0, stream>>>((const bf16_t*)qkv, ld, (bf16_t*)ctx, ldo, lse, lens, S, H, scale, abl);
    }
    else { fs2_set_error("attn_fwd: dtype"); return FS2_EDTYPE; }
    FS2_CHECK_LAUNCH("attn_fwd");
    return FS2_OK;
}

// ------------------------------------------------------------------ backward, part 0: D[b,h,q] = sum_d dO*O
template <typename T>
__global__ void attn_delta_kernel(const T* __restrict__ ctx, const T* __restrict__ dctx, float* __restrict__ delta, int rows,
                                  int S, int H) {
    int idx = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);  // idx = row*H + h
    if (idx >= rows * H) return;
    int lane = threadIdx.x & 63;
    int row = idx / H, h = idx - row * H;
    size_t off = (size_t)row * H * DK + h * DK + lane * 2;
    float s = Elem<T>::ld(ctx + off) * Elem<T>::ld(dctx + off) + Elem<T>::ld(ctx + off + 1) * Elem<T>::ld(dctx + off + 1);
    s = wave_sum(s);
    if (lane == 0) {
        int b = row / S, q = row - b * S;
        delta[((size_t)b * H + h) * S + q] = s;
    }
}

// ------------------------------------------------------------------ backward, part 1: dK, dV
// One wave owns 32 keys; the block's 4 waves share the streamed Q / dO tiles.
// dynamic LDS: [4 waves][K 16K | V 16K] + Q 16K + dO 16K = 160 KiB.
template <typename T>
__global__ void __launch_bounds__(256, 1) attn_bwd_dkv_kernel(const T* __restrict__ qkv, long ld, const T* __restrict__ dctx,
                                                              long ldo, const float* __restrict__ lse,
                                                              const float* __restrict__ delta, T* __restrict__ dqkv,
                                                              const int32_t* __restrict__ lens, int S, int H, float scale) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, fl = lane & 31, h2 = lane >> 5;
    const int kt = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    const int len = lens ? min(lens[b], S) : S;
    const size_t rowbase = (size_t)b * S;
    const int kbase = kt * 128 + w * 32;
    T* dK = dqkv + rowbase * ld + (size_t)H * DK + h * DK;
    T* dV = dqkv + rowbase * ld + (size_t)2 * H * DK + h * DK;
    if (kt * 128 >= len) {   // all keys of this block are padding: gradients are zero
        for (int i = tid; i < 128 * 32; i += 256) {
            int r = i >> 5, c = (i & 31) * 4;
            if (kt * 128 + r < S) {
                st4<T>(dK + (size_t)(kt * 128 + r) * ld + c, make_float4(0.f, 0.f, 0.f, 0.f));
                st4<T>(dV + (size_t)(kt * 128 + r) * ld + c, make_float4(0.f, 0.f, 0.f, 0.f));
            }
        }
        return;
    }
    unsigned char* sKw = smem + w * 32768;
    unsigned char* sVw = sKw + 16384;
    unsigned char* sQ = smem + 131072;
    unsigned char* sdO = sQ + 16384;
    const T* Q = qkv + rowbase * ld + h * DK;
    const T* K = qkv + rowbase * ld + (size_t)H * DK + h * DK;
    const T* V = qkv + rowbase * ld + (size_t)2 * H * DK + h * DK;
    const T* dO = dctx + rowbase * ldo + h * DK;
    const float* lse_b = lse + ((size_t)b * H + h) * S;
    const float* del_b = delta + ((size_t)b * H + h) * S;
    // each wave stages its own K / V sub-tiles (64 threads)
    stage_tile<T, 64>(sKw, K, ld, kbase, max(0, min(32, S - kbase)), lane);
    stage_tile<T, 64>(sVw, V, ld, kbase, max(0, min(32, S - kbase)), lane);
    const bool key_ok = (kbase + fl) < len;

    f32x16 dk[4], dv[4];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dk[nb][r] = 0.f; dv[nb][r] = 0.f; }

    for (int q0 = 0; q0 < len; q0 += 32) {
        __syncthreads();
        stage_tile<T, 256>(sQ, Q, ld, q0, min(32, S - q0), tid);
        stage_tile<T, 256>(sdO, dO, ldo, q0, min(32, S - q0), tid);
        __syncthreads();
        f32x16 s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
        // S[q][key] : A = Q (lane row = q), B = K^T (lane col = key)
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            float4 qa = *reinterpret_cast<const float4*>(sQ + swz(fl, h2 * 16 + u));
            float4 kb = *reinterpret_cast<const float4*>(sKw + swz(fl, h2 * 16 + u));
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(qa.x, kb.x, s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(qa.y, kb.y, s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(qa.z, kb.z, s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(qa.w, kb.w, s, 0, 0, 0);
        }
        // dP[q][key] : A = dO, B = V^T
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            float4 da = *reinterpret_cast<const float4*>(sdO + swz(fl, h2 * 16 + u));
            float4 vb = *reinterpret_cast<const float4*>(sVw + swz(fl, h2 * 16 + u));
            dp = __builtin_amdgcn_mfma_f32_32x32x2f32(da.x, vb.x, dp, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x2f32(da.y, vb.y, dp, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x2f32(da.z, vb.z, dp, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x2f32(da.w, vb.w, dp, 0, 0, 0);
        }
        // lane = key (col), reg r = query row crow(r,h2)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            int q = q0 + crow(r, h2);
            float p = 0.f, ds = 0.f;
            if (key_ok && q < len) {
                p = __expf(s[r] * scale - lse_b[q]);
                ds = p * (dp[r] - del_b[q]) * scale;
            }
            s[r] = p; dp[r] = ds;
        }
        // dV[key][d] += P^T dO ; dK[key][d] += dS^T Q     (A = regs, lane row = key, k-slot = q)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            int qr = crow(r, h2);
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) {
                int col = nb * 32 + fl;
                float ob = *reinterpret_cast<const float*>(sdO + swz(qr, col >> 2) + (col & 3) * 4);
                float qb = *reinterpret_cast<const float*>(sQ + swz(qr, col >> 2) + (col & 3) * 4);
                dv[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(s[r], ob, dv[nb], 0, 0, 0);
                dk[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(dp[r], qb, dk[nb], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        int key = kbase + crow(r, h2);
        if (key < S) {
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) {
                Elem<T>::st(dK + (size_t)key * ld + nb * 32 + fl, dk[nb][r]);
                Elem<T>::st(dV + (size_t)key * ld + nb * 32 + fl, dv[nb][r]);
            }
        }
    }
}

// ------------------------------------------------------------------ backward, part 2: dQ
template <typename T>
__global__ void __launch_bounds__(256, 1) attn_bwd_dq_kernel(const T* __restrict__ qkv, long ld, const T* __restrict__ dctx,
                                                             long ldo, const float* __restrict__ lse,
                                                             const float* __restrict__ delta, T* __restrict__ dqkv,
                                                             const int32_t* __restrict__ lens, int S, int H, float scale) {
    __shared__ __attribute__((aligned(16))) unsigned char sK[32 * 512];
    __shared__ __attribute__((aligned(16))) unsigned char sV[32 * 512];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, fl = lane & 31, h2 = lane >> 5;
    const int qt = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    const int len = lens ? min(lens[b], S) : S;
    const int q0 = qt * 128;
    const size_t rowbase = (size_t)b * S;
    T* dQ = dqkv + rowbase * ld + h * DK;
    if (q0 >= len) {
        for (int i = tid; i < 128 * 32; i += 256) {
            int r = i >> 5, c = (i & 31) * 4;
            if (q0 + r < S) st4<T>(dQ + (size_t)(q0 + r) * ld + c, make_float4(0.f, 0.f, 0.f, 0.f));
        }
        return;
    }
    const T* Q = qkv + rowbase * ld + h * DK;
    const T* K = qkv + rowbase * ld + (size_t)H * DK + h * DK;
    const T* V = qkv + rowbase * ld + (size_t)2 * H * DK + h * DK;
    const T* dO = dctx + rowbase * ldo + h * DK;
    const int myq = q0 + w * 32 + fl;
    const bool q_ok = myq < len;
    float my_lse = 0.f, my_del = 0.f;
    if (myq < S) { my_lse = lse[((size_t)b * H + h) * S + myq]; my_del = delta[((size_t)b * H + h) * S + myq]; }
    float qf[64], df[64];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f), d = v;
        if (myq < S) { v = ld4<T>(Q + (size_t)myq * ld + h2 * 64 + u * 4); d = ld4<T>(dO + (size_t)myq * ldo + h2 * 64 + u * 4); }
        qf[u * 4 + 0] = v.x; qf[u * 4 + 1] = v.y; qf[u * 4 + 2] = v.z; qf[u * 4 + 3] = v.w;
        df[u * 4 + 0] = d.x; df[u * 4 + 1] = d.y; df[u * 4 + 2] = d.z; df[u * 4 + 3] = d.w;
    }
    f32x16 dq[4];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[nb][r] = 0.f;

    for (int k0 = 0; k0 < len; k0 += 32) {
        __syncthreads();
        stage_tile<T, 256>(sK, K, ld, k0, min(32, S - k0), tid);
        stage_tile<T, 256>(sV, V, ld, k0, min(32, S - k0), tid);
        __syncthreads();
        f32x16 s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
        // S^T[key][q] : A = K (lane row = key), B = Q^T (lane col = q) ; dP^T[key][q] : A = V, B = dO^T
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            float4 ka = *reinterpret_cast<const float4*>(sK + swz(fl, h2 * 16 + u));
            float4 va = *reinterpret_cast<const float4*>(sV + swz(fl, h2 * 16 + u));
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(ka.x, qf[u * 4 + 0], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(ka.y, qf[u * 4 + 1], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(ka.z, qf[u * 4 + 2], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(ka.w, qf[u * 4 + 3], s, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x2f32(va.x, df[u * 4 + 0], dp, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x2f32(va.y, df[u * 4 + 1], dp, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x2f32(va.z, df[u * 4 + 2], dp, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x2f32(va.w, df[u * 4 + 3], dp, 0, 0, 0);
        }
        // lane = q (col), reg r = key row crow(r,h2)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            int key = k0 + crow(r, h2);
            float ds = 0.f;
            if (q_ok && key < len) {
                float p = __expf(s[r] * scale - my_lse);
                ds = p * (dp[r] - my_del) * scale;
            }
            s[r] = ds;
        }
        // dQ[q][d] += dS K   (A = regs: lane row = q, k-slot = key)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            int kr = crow(r, h2);
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) {
                int col = nb * 32 + fl;
                float kb = *reinterpret_cast<const float*>(sK + swz(kr, col >> 2) + (col & 3) * 4);
                dq[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(s[r], kb, dq[nb], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        int q = q0 + w * 32 + crow(r, h2);
        if (q < S) {
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) Elem<T>::st(dQ + (size_t)q * ld + nb * 32 + fl, dq[nb][r]);
        }
    }
}

extern "C" int fs2_attn_bwd(const void* qkv, const void* ctx, const void* dctx, const float* lse, float* delta, void* dqkv,
                            const int32_t* lens, int B, int S, int H, int dk, float scale, int dtype, hipStream_t stream) {
    FS2_CHECK_ARG(qkv && ctx && dctx && lse && delta && dqkv, "attn_bwd: null pointer");
    FS2_CHECK_ARG(dk == DK, "attn_bwd: only d_k = 128 is supported (got %d)", dk);
    if (B == 0) return FS2_OK;
    long ld = 3L * H * DK, ldo = (long)H * DK;
    int rows = B * S;
    dim3 grid(fs2_cdiv(S, 128), H, B);
    const int dyn = 163840;
    if (dtype == FS2_F32) {
        attn_delta_kernel<float><<<fs2_cdiv(rows * H, 4), 256, 0, stream>>>((const float*)ctx, (const float*)dctx, delta, rows, S, H);
        static Fs2DevOnce once;
        once.run([&] { (void)hipFuncSetAttribute((const void*)attn_bwd_dkv_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, dyn); });
        attn_bwd_dkv_kernel<float><<<grid, 256, dyn, stream>>>((const float*)qkv, ld, (const float*)dctx, ldo, lse, delta, (float*)dqkv, lens, S, H, scale);
        attn_bwd_dq_kernel<float><<<grid, 256, 0, stream>>>((const float*)qkv, ld, (const float*)dctx, ldo, lse, delta, (float*)dqkv, lens, S, H, scale);
    } else if (dtype == FS2_BF16) {
        // dQ first: it also produces delta (row sums of dO * O) for the dK/dV kernel
        attn_bwd_dq_bf16_kernel<<<grid, 256, 0, stream>>>((const bf16_t*)qkv, ld, (const bf16_t*)ctx, (const bf16_t*)dctx, ldo, lse, delta, (bf16_t*)dqkv, lens, S, H, scale);
        attn_bwd_dkv_bf16_kernel<<<grid, 256, 0, stream>>>((const bf16_t*)qkv, ld, (const bf16_t*)dctx, ldo, lse, delta, (bf16_t*)dqkv, lens, S, H, scale);
    } else { fs2_set_error("attn_bwd: dtype"); return FS2_EDTYPE; }
    FS2_CHECK_LAUNCH("attn_bwd");
    return FS2_OK;
}
