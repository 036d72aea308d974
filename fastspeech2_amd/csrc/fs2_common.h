// fs2_common.h — shared device/host helpers for libfs2hip (gfx950 / CDNA4 only).
// Conventions for every entry point (see include/fs2hip.h):
//   * caller owns every buffer; functions never allocate, never synchronise;
//   * launches are ordered on the hipStream_t passed as the last argument;
//   * return 0 on success, a negative FS2_E* code otherwise (message via fs2_last_error()).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <mutex>
#include <stdlib.h>

// Development switches (ablations, forced variants, split-depth sweeps) exist only in -DFS2_DEV builds (`make dev` ->
// libfs2hip_dev.so, loaded through FS2_LIB_PATH by the tools/ scripts).  In the shipped library every switch is a
// compile-time constant: no environment variable can change what the product computes or skip work inside a timed region.
#ifdef FS2_DEV
static inline int fs2_dev_env(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
#define FS2_DEV_DBG(x) (x)
#else
#define fs2_dev_env(name, dflt) (dflt)
#define FS2_DEV_DBG(x) false
#endif

#define FS2_OK 0
#define FS2_EINVAL (-1)    // bad shape / null pointer / unsupported combination
#define FS2_EDTYPE (-2)    // unsupported dtype
#define FS2_ELAUNCH (-3)   // hip launch failure

enum { FS2_F32 = 0, FS2_BF16 = 1 };
enum { FS2_ACT_NONE = 0, FS2_ACT_RELU = 1, FS2_ACT_TANH = 2, FS2_ACT_LRELU = 3, FS2_ACT_GATE = 4 };

void fs2_set_error(const char* fmt, ...);

#define FS2_CHECK_ARG(cond, ...)                \
    do {                                        \
        if (!(cond)) {                          \
            fs2_set_error(__VA_ARGS__);         \
            return FS2_EINVAL;                  \
        }                                       \
    } while (0)

// One-time, PER-DEVICE host-side setup of a kernel (dynamic-LDS opt-in via hipFuncSetAttribute is a per-device property):
//   static Fs2DevOnce once;  once.run([&] { hipFuncSetAttribute(...); });
// Thread-safe (autograd / DDP hook threads call the ABI concurrently); the uncontended cost is one mutex + hipGetDevice.
struct Fs2DevOnce {
    std::mutex mu;
    uint64_t done = 0;
    template <typename F>
    void run(F&& f) {
        int d = 0;
        (void)hipGetDevice(&d);
        const uint64_t bit = 1ull << (d & 63);
        std::lock_guard<std::mutex> g(mu);
        if (!(done & bit)) { f(); done |= bit; }
    }
};

#define FS2_CHECK_LAUNCH(name)                                                        \
    do {                                                                              \
        hipError_t e__ = hipGetLastError();                                           \
        if (e__ != hipSuccess) {                                                      \
            fs2_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));     \
            return FS2_ELAUNCH;                                                       \
        }                                                                             \
    } while (0)

// ---------------------------------------------------------------- bf16 <-> f32
// bf16 is carried as raw uint16_t in HBM; conversions are round-to-nearest-even.
typedef uint16_t bf16_t;

__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
// gfx950 converts in hardware (v_cvt_pk_bf16_f32, round-to-nearest-even, two lanes' worth per instruction): one VALU op
// per PAIR instead of ~6 per element for the integer emulation.
typedef __bf16 fs2_bf16x2 __attribute__((ext_vector_type(2)));
typedef float fs2_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ bf16_t f32_to_bf16(float f) { return __builtin_bit_cast(unsigned short, (__bf16)f); }
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    return __builtin_bit_cast(uint32_t, __builtin_convertvector((fs2_f32x2){lo, hi}, fs2_bf16x2));
}

template <typename T> struct Elem;
template <> struct Elem<float> {
    static __device__ __forceinline__ float ld(const float* p) { return *p; }
    static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct Elem<bf16_t> {
    static __device__ __forceinline__ float ld(const bf16_t* p) { return bf16_to_f32(*p); }
    static __device__ __forceinline__ void st(bf16_t* p, float v) { *p = f32_to_bf16(v); }
};

// 4-element vector load/store as float4 regardless of storage type (16 B or 8 B per access).
template <typename T> __device__ __forceinline__ float4 ld4(const T* p);
template <> __device__ __forceinline__ float4 ld4<float>(const float* p) { return *reinterpret_cast<const float4*>(p); }
template <> __device__ __forceinline__ float4 ld4<bf16_t>(const bf16_t* p) {
    uint2 u = *reinterpret_cast<const uint2*>(p);
    float4 r;
    r.x = __uint_as_float(u.x << 16);
    r.y = __uint_as_float(u.x & 0xffff0000u);
    r.z = __uint_as_float(u.y << 16);
    r.w = __uint_as_float(u.y & 0xffff0000u);
    return r;
}
template <typename T> __device__ __forceinline__ void st4(T* p, float4 v);
template <> __device__ __forceinline__ void st4<float>(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
template <> __device__ __forceinline__ void st4<bf16_t>(bf16_t* p, float4 v) {
    uint2 u;
    u.x = pack_bf16x2(v.x, v.y);
    u.y = pack_bf16x2(v.z, v.w);
    *reinterpret_cast<uint2*>(p) = u;
}

// ---------------------------------------------------------------- wave64 reductions
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// ---------------------------------------------------------------- counter-based dropout RNG
// keep(seed, idx) is a pure function so backward regenerates the mask instead of storing it.
// 32-bit mix (two rounds of a murmur3-style finaliser over seed-keyed counter); uniform in [0,1).
__device__ __forceinline__ uint32_t fs2_hash32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du;
    x ^= x >> 15; x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}
__device__ __forceinline__ float fs2_uniform(uint64_t seed, uint32_t idx) {
    uint32_t h = fs2_hash32(idx ^ (uint32_t)seed);
    h = fs2_hash32(h + (uint32_t)(seed >> 32) + 0x9e3779b9u);
    return (float)(h >> 8) * (1.0f / 16777216.0f);
}
// returns the multiplicative dropout factor: 0 or 1/(1-p); p<=0 -> 1.
__device__ __forceinline__ float fs2_drop_scale(uint64_t seed, uint32_t idx, float p, float inv_keep) {
    return (fs2_uniform(seed, idx) >= p) ? inv_keep : 0.0f;
}

// fs2_conv_gemm_variant() codes (mirrors include/fs2hip.h)
#ifndef FS2_GEMM_PLAIN
#define FS2_GEMM_PLAIN 1
#define FS2_GEMM_DMA 2
#define FS2_GEMM_RING 3
#define FS2_GEMM_SKINNY 4
#define FS2_GEMM_PERSIST 5
#define FS2_GEMM_PERSIST_1TAP 6
#define FS2_GEMM_WIDE_1TAP 7
#define FS2_GEMM_STREAM_K256 9
#endif
static inline int fs2_cdiv(long a, long b) { return (int)((a + b - 1) / b); }
