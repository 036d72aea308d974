// fs2_calib.hip - measurement aid: what the matrix pipes of THIS chip sustain on a back-to-back v_mfma_f32_32x32x16_bf16 stream.
//
// bench.py prices the contraction kernels against the nominal dense bf16 peak (2.5 PFLOP/s = 256 CUs x 4 SIMDs x one 32x32x16
// MFMA per 32 cycles at 2.4 GHz, MI355X_MICROARCH.md).  Under a sustained MFMA load the chip clocks to its power budget (the same
// guide, "DVFS give-back": ~1.9-2.0 GHz on random data), so a kernel that issued nothing but MFMAs would still not see 2.5 PF.
// This kernel is that stream - one wave per SIMD, eight independent accumulators, register operands, nothing else in the loop -
// run for a few hundred microseconds on every CU; bench.py reports its rate beside the nominal peak (`roofline.mfma_sustained`) so
// that a reader can tell how much of a kernel's distance to 2.5 PF is the kernel and how much is the clock.  Round 4's ablations of
// the convolution kernels put "MFMAs + barriers only" at the same place (round 4's tall-tile experiment: DESIGN.md §3, git ff5fda0:fastspeech2_amd/csrc/fs2_gemm_t.hip).
#include "fs2_common.h"

typedef float calib_f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 calib_bf16x8 __attribute__((ext_vector_type(8)));

__global__ void __launch_bounds__(256) mfma_calibrate_kernel(int iters, float* __restrict__ sink) {
    // operands: small non-zero values that differ per lane (zero operands draw less power and clock higher)
    calib_bf16x8 a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        a[i] = (__bf16)(1e-3f * (float)(((threadIdx.x * 7 + i * 3) & 15) - 7));
        b[i] = (__bf16)(1e-3f * (float)(((threadIdx.x * 5 + i * 11) & 15) - 8));
    }
    calib_f32x16 acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) s += acc[j][0] + acc[j][7];
    if (s == 12345.678f) sink[0] = s;                                    // keeps the accumulators live
}

// Launches one 256-thread workgroup (one wave per SIMD) per CU, `iters` x 8 MFMAs per wave; returns the FLOPs of the launch through
// *flops (2 * 32 * 32 * 16 per MFMA) so that the caller only needs the duration.
extern "C" int fs2_mfma_calibrate(int iters, float* sink, double* flops, hipStream_t stream) {
    FS2_CHECK_ARG(iters > 0 && sink && flops, "mfma_calibrate: bad arguments");
    int d = 0, cus = 0;
    (void)hipGetDevice(&d);
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, d) != hipSuccess || cus <= 0) cus = 256;
    mfma_calibrate_kernel<<<cus, 256, 0, stream>>>(iters, sink);
    FS2_CHECK_LAUNCH("mfma_calibrate");
    *flops = (double)cus * 4.0 * (double)iters * 8.0 * (2.0 * 32 * 32 * 16);
    return FS2_OK;
}

// ---- the two memory-side calibrations next to it (the driver's box had MORE sustained MFMA and ran every kernel
// 8-10 % slower than the builder's - the spread is on the memory / fabric side, so the line now says what THIS box streams):
//   fs2_hbm_calibrate     16-byte-per-lane copy of `bytes` (read + write through HBM when the buffers exceed the 256 MB Infinity Cache)
//   fs2_ldsdma_calibrate  every CU streams an L2-resident window through LDS by LDS-DMA (global_load_lds_dwordx4, 1 KiB per
//                         wave-instruction, eight waves, `depth` pieces in flight per wave): the operand path of every contraction
//                         kernel of this library; the persistent kernels sit at ~10 TB/s of it (fs2_gemm_p.hip header)
__global__ void __launch_bounds__(256) hbm_copy_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n) {
    const size_t stride = (size_t)gridDim.x * 256 * 4;
    for (size_t i = (size_t)blockIdx.x * 256 * 4 + threadIdx.x; i < n; i += stride) {
        uint4 v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = (i + j * 256 < n) ? src[i + j * 256] : make_uint4(0, 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (i + j * 256 < n) dst[i + j * 256] = v[j];
    }
}
extern "C" int fs2_hbm_calibrate(const void* src, void* dst, size_t bytes, hipStream_t stream) {
    FS2_CHECK_ARG(src && dst && bytes >= 16 && ((uintptr_t)src & 15) == 0 && ((uintptr_t)dst & 15) == 0, "hbm_calibrate: bad arguments");
    int d = 0, cus = 0;
    (void)hipGetDevice(&d);
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, d) != hipSuccess || cus <= 0) cus = 256;
    hbm_copy_kernel<<<cus * 8, 256, 0, stream>>>((const uint4*)src, (uint4*)dst, bytes / 16);
    FS2_CHECK_LAUNCH("hbm_calibrate");
    return FS2_OK;
}

// window: `window_bytes` of src per XCD-sized group of workgroups (blockIdx % 8 picks the window: the workgroups the dispatcher places
// on one XCD re-read the same 1-2 MB from that XCD's L2).  Each wave keeps 4 pieces (4 KiB) in flight into its own 4 KiB of LDS.
__global__ void __launch_bounds__(512) ldsdma_calibrate_kernel(const unsigned char* __restrict__ src, unsigned window_bytes, int iters,
                                                               float* __restrict__ sink) {
    __shared__ __attribute__((aligned(16))) unsigned char ring[8 * 4096];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)ring + (unsigned)(wave * 4096);
    const unsigned char* base = src + (size_t)(blockIdx.x & 7) * window_bytes;
    // piece p of this wave in iteration it: window offset ((it * 8 + wave) * 4 + p) KiB + a per-workgroup rotation, modulo the window
    unsigned off = (unsigned)((blockIdx.x >> 3) * 37 * 1024) % window_bytes;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const unsigned o = (off + (unsigned)((wave * 4 + p) * 1024)) % window_bytes;
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(o + (unsigned)(lane * 16)), "s"(base),
                         "s"(__builtin_amdgcn_readfirstlane(lds0 + (unsigned)(p * 1024))) : "memory");
        }
        off = (off + 32 * 1024) % window_bytes;
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");                       // the previous iteration's four pieces have landed
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (ring[threadIdx.x * 16] == 0xA5 && ring[threadIdx.x * 16 + 5] == 0x5A && iters < 0) sink[0] = 1.f;      // keeps the ring live
}
// *bytes receives the bytes moved L2 -> LDS by the launch
extern "C" int fs2_ldsdma_calibrate(const void* src, size_t src_bytes, int iters, float* sink, double* bytes, hipStream_t stream) {
    FS2_CHECK_ARG(src && sink && bytes && iters > 0 && src_bytes >= 8u * (1u << 20) && ((uintptr_t)src & 15) == 0, "ldsdma_calibrate: bad arguments");
    int d = 0, cus = 0;
    (void)hipGetDevice(&d);
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, d) != hipSuccess || cus <= 0) cus = 256;
    const unsigned window = 1u << 20;                                            // 1 MiB per XCD: L2-resident (4 MiB per XCD)
    ldsdma_calibrate_kernel<<<cus, 512, 0, stream>>>((const unsigned char*)src, window, iters, sink);
    FS2_CHECK_LAUNCH("ldsdma_calibrate");
    *bytes = (double)cus * 8.0 * 4.0 * 1024.0 * (double)iters;
    return FS2_OK;
}

#ifdef FS2_DEV
// dev aid: which SIMD does wave i of a workgroup of `threads` threads land on?  out[block * 16 + wave] = HW_ID
__global__ void wave_map_kernel(int* __restrict__ out) {
    extern __shared__ unsigned char smem_[];
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = (int)id;
}
extern "C" int fs2_dev_wave_map(int* out, int blocks, int threads, int lds, hipStream_t stream) {
    (void)hipFuncSetAttribute((const void*)wave_map_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    wave_map_kernel<<<blocks, threads, lds, stream>>>(out);
    FS2_CHECK_LAUNCH("wave_map");
    return FS2_OK;
}
#endif
