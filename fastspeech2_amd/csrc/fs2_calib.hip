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
