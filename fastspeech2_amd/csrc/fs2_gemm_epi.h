// fs2_gemm_epi.h - the register epilogue of the transposed-accumulator contraction kernels (fs2_gemm_p.hip: persistent 256x128
// kernel; fs2_gemm_w.hip: wide one-tap kernel): a wave's 64 (M) x 128 (N) tile goes from its accumulators straight to bf16 rows.
#pragma once
#include "fs2_gemm.h"

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// epilogue straight from the transposed accumulators: lane (fl, fh) holds, for row m = mbase + fl, the columns
// nb*32 + 8*g + 4*fh + e (g = r>>2, e = r&3).  v_permlane32_swap pairs (g, g+1) across the two half-waves so that a lane ends
// up with columns nb*32 + fh*8 + [0,8) (from g = 0,1) and nb*32 + 16 + fh*8 + [0,8) (from g = 2,3): two 16-byte bf16 runs.
// MB = 32-row blocks a wave owns (2: the 64 x 128 wave tile; 1: 32 x 128, the twelve-wave form of the persistent kernel); wave wm's rows start at wm * 32 * MB
template <int ACT, int MB>
__device__ __forceinline__ void p_epilogue(const ConvGemmArgs& a, f32x16 (&acc)[MB][4], int m0, int n0, int wm, int fl, int fh,
                                           const int32_t* lens_s, float* bias_s, int lane) {
    bf16_t* Y = reinterpret_cast<bf16_t*>(a.Y);
    const bf16_t* R = reinterpret_cast<const bf16_t*>(a.R);
    const bool gate = a.act == FS2_ACT_GATE;
    // bias line of this tile -> this wave's private LDS line (same wave writes and reads: LDS ops are in order)
    if (a.bias) {
        if (lane < 32) {
            float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
            const int n = n0 + lane * 4;
            if (n + 4 <= a.N) bv = *reinterpret_cast<const float4*>(a.bias + n);
            else { float t[4] = {0.f, 0.f, 0.f, 0.f}; for (int e = 0; e < 4; ++e) if (n + e < a.N) t[e] = a.bias[n + e]; bv = make_float4(t[0], t[1], t[2], t[3]); }
            *reinterpret_cast<float4*>(bias_s + lane * 4) = bv;
        }
    }
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
        const int m = m0 + wm * (32 * MB) + mb * 32 + fl;
        const bool rowok = m < a.M;
        bool padrow = false;
        if (a.lens && rowok) { const int b = m / a.S; padrow = (m - b * a.S) >= lens_s[b]; }
        bf16_t* yrow = Y + (size_t)m * a.ldy;
        const bf16_t* rrow = R ? R + (size_t)m * a.ldr : nullptr;
        // (The residual / old-Y chunks are loaded where they are used: load, wait, add, store, chunk by chunk - the ISA shows a
        // vmcnt(0) behind every load once a store is pending.  Hoisting them ahead of the stores - all 16 of a tile, 8 per row block,
        // 4 per pair of column blocks - spilled 50 / 9 / 0 registers of the 256 and the spill-free form measured SLOWER on one box:
        // batch synthesis 7.30 -> 7.52 ms, train step 8.40 -> 8.48 ms (profiles/r05p_ab_*_epilogue_hoist.log).  Not kept.)
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
            float c[2][8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                u32x2 s0 = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[mb][nb][e]), __float_as_uint(acc[mb][nb][4 + e]), false, false);
                u32x2 s1 = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[mb][nb][8 + e]), __float_as_uint(acc[mb][nb][12 + e]), false, false);
                c[0][e] = __uint_as_float(s0[0]); c[0][4 + e] = __uint_as_float(s0[1]);
                c[1][e] = __uint_as_float(s1[0]); c[1][4 + e] = __uint_as_float(s1[1]);
            }
#pragma unroll
            for (int ch = 0; ch < 2; ++ch) {
                const int nl = nb * 32 + ch * 16 + fh * 8;
                const int n = n0 + nl;
                if (!rowok || n >= a.N) continue;            // N % 8 == 0: a chunk is inside or outside
                float v[8];
                if (a.bias) {
                    const float4 b0 = *reinterpret_cast<const float4*>(bias_s + nl), b1 = *reinterpret_cast<const float4*>(bias_s + nl + 4);
                    const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = act_ct<ACT>(c[ch][e] + bb[e], a.slope);
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = act_ct<ACT>(c[ch][e], a.slope);
                }
                if (rrow) {
                    const uint4 rr = *reinterpret_cast<const uint4*>(rrow + n);
                    const uint32_t* u = reinterpret_cast<const uint32_t*>(&rr);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float r0 = __uint_as_float(u[e] << 16), r1 = __uint_as_float(u[e] & 0xffff0000u);
                        if (a.res_unlrelu > 0.f) { r0 = r0 > 0.f ? r0 : r0 * a.res_unlrelu; r1 = r1 > 0.f ? r1 : r1 * a.res_unlrelu; }
                        v[2 * e] = gate ? (r0 > 0.f ? v[2 * e] : 0.f) : v[2 * e] + r0;
                        v[2 * e + 1] = gate ? (r1 > 0.f ? v[2 * e + 1] : 0.f) : v[2 * e + 1] + r1;
                    }
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) { v[e] *= a.out_scale; if (padrow) v[e] = 0.f; }
                if (a.accumulate) {
                    const uint4 yy = *reinterpret_cast<const uint4*>(yrow + n);
                    const uint32_t* u = reinterpret_cast<const uint32_t*>(&yy);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v[2 * e] += __uint_as_float(u[e] << 16); v[2 * e + 1] += __uint_as_float(u[e] & 0xffff0000u); }
                }
                if (a.post_slope > 0.f) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * a.post_slope;
                }
                uint4 o;
                uint32_t* ou = reinterpret_cast<uint32_t*>(&o);
#pragma unroll
                for (int e = 0; e < 4; ++e) ou[e] = pack_bf16x2(v[2 * e], v[2 * e + 1]);
                *reinterpret_cast<uint4*>(yrow + n) = o;
            }
        }
    }
}

