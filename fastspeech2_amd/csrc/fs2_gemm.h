// fs2_gemm.h - declarations shared by the implicit-GEMM kernel files (fs2_gemm.hip, fs2_gemm_p.hip).
#pragma once
#include "fs2_common.h"
#include <stdlib.h>
#include <type_traits>


typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct ConvGemmArgs {
    const void* X; long ldx;
    const void* W; long ldw;       // packed [N][taps*Cin]
    const float* bias;             // [N] or null
    const void* R; long ldr;       // residual added after activation, or null
    void* Y; long ldy;
    const int32_t* lens;           // per-sequence valid rows (rows t >= lens[b] are written as 0) or null
    int M, N, Cin, S, taps, dil, pad;
    int act; float slope;          // output activation
    int in_act; float in_slope;    // activation applied to X on load (leaky-relu prologue of HiFi-GAN)
    int accumulate; float out_scale;
    // HiFi-GAN's pre-activation chains store leaky-ReLU'd activations (round 5): a convolution then reads its operand without a
    // prologue (the prologue cost 27-35 % of every launch that had one: 13 VALU per MFMA in the consumer's issue stream,
    // profiles/r05l_bench_voc.log), and the two places that need the RAW value undo it:
    float res_unlrelu;             // > 0: the residual operand R holds lrelu(r, slope) - use r > 0 ? r : r * res_unlrelu (= 1 / slope)
    float post_slope;              // > 0: the value stored is lrelu(v, post_slope), applied last (after residual, scale, accumulate)
    int vec_ok;                    // Y / R rows are 16-byte addressable (ld % elems-per-16B == 0, aligned bases)
    int dbg;                       // dev ablations (FS2_GEMM_DBG): 1 = loaders issue no DMA, 2 = consumers issue no MFMA
};

// gemm_res_ln (N == 256: a workgroup owns whole rows): the epilogue of the projection IS the LayerNorm kernel
//   z = dropout(acc + bias) + res  (stored bf16 into Y: saved for backward)    out = mask(LN(z) * gamma + beta)    mean / rstd saved
struct WLn {
    const float* gamma; const float* beta;   // gamma == null: plain contraction
    void* out; long ldo;
    float* mean; float* rstd;
    float eps, p_pre;
    uint64_t seed_pre;
    const uint64_t* seed_dev;
};

__device__ __forceinline__ float apply_act(float v, int act, float slope) {
    switch (act) {
        case FS2_ACT_RELU: return fmaxf(v, 0.f);
        case FS2_ACT_TANH: return tanhf(v);
        case FS2_ACT_LRELU: return v > 0.f ? v : v * slope;
        default: return v;
    }
}

// activation with the kind fixed at COMPILE time.  The epilogues run it on 64-128 accumulators per lane in fully unrolled
// loops; with the run-time switch above every one of those elements carried its own scalar branch ladder plus an inlined
// tanhf (the 256x128 kernel grew to ~17 k instructions, far beyond the instruction cache, and its epilogue cost ~8 us per
// workgroup: r01i ablation 277 us with / 212 us without epilogue on the k=9 FFN conv).  Now the switch runs ONCE per
// epilogue and selects a straight-line instantiation.
template <int ACT> __device__ __forceinline__ float act_ct(float v, float slope) {
    if (ACT == FS2_ACT_RELU) return fmaxf(v, 0.f);
    if (ACT == FS2_ACT_TANH) return tanhf(v);
    if (ACT == FS2_ACT_LRELU) return v > 0.f ? v : v * slope;
    return v;                                            // none / gate (the gate is applied with the residual operand)
}
#define FS2_ACT_DISPATCH(act, CALL) do { switch (act) { \
    case FS2_ACT_RELU: { constexpr int ACT = FS2_ACT_RELU; CALL; } break; \
    case FS2_ACT_TANH: { constexpr int ACT = FS2_ACT_TANH; CALL; } break; \
    case FS2_ACT_LRELU: { constexpr int ACT = FS2_ACT_LRELU; CALL; } break; \
    default: { constexpr int ACT = FS2_ACT_NONE; CALL; } break; } } while (0)

template <typename T> struct MmaTraits;
template <> struct MmaTraits<float> { static constexpr int EPC = 4; };   // elements per 16-B chunk
template <> struct MmaTraits<bf16_t> { static constexpr int EPC = 8; };

__device__ __forceinline__ uint4 act_chunk_f32(uint4 v, float slope) {
    float* f = reinterpret_cast<float*>(&v);
#pragma unroll
    for (int i = 0; i < 4; ++i) f[i] = f[i] > 0.f ? f[i] : f[i] * slope;
    return v;
}
__device__ __forceinline__ uint4 act_chunk_bf16(uint4 v, float slope) {
    uint32_t* u = reinterpret_cast<uint32_t*>(&v);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float lo = __uint_as_float(u[i] << 16), hi = __uint_as_float(u[i] & 0xffff0000u);
        lo = lo > 0.f ? lo : lo * slope;
        hi = hi > 0.f ? hi : hi * slope;
        u[i] = pack_bf16x2(lo, hi);
    }
    return v;
}

__device__ __forceinline__ void glds16_sbase(unsigned voff, const void* sbase, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ unsigned lds_addr(const void* p) {
    return (unsigned)(size_t)(__attribute__((address_space(3))) const void*)p;
}
