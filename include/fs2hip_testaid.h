/* fs2hip_testaid.h - TEST AIDS, not part of the product ABI (include/fs2hip.h is what a reference-side binder calls).
 *
 * tests/aids/libfs2_testaid.so is a host-only library built from the same schedule / layout source the kernels compile
 * (fastspeech2_amd/csrc/fs2_sched.h), so that tests/test_schedule_cpu.py can prove properties of code that otherwise only runs
 * on the device.  libfs2hip.so exports none of these symbols. */
#ifndef FS2HIP_TESTAID_H
#define FS2HIP_TESTAID_H
#ifdef __cplusplus
extern "C" {
#endif
/* position of logical column c (0..127) in a 128-float epilogue staging row of the 128x128 / ring / skinny kernels for an output
 * element of elem_bytes (4 = f32, 2 = bf16); negative on bad arguments */
int fs2t_stage_tile_col(int c, int elem_bytes);
/* the persistent kernel's unit list of workgroup b (n_real real M-tiles, ntn N-tiles, G workgroups, tile order 0 / 1, uniform
 * K-split ks over nkc Cin chunks, tail split of at most tks_max, 1 = none).  out: up to max_units x {real-tile index, N-tile,
 * first chunk, chunk count, tail parts}.  Returns the unit count (NOT capped at 64: the caller checks the kernel's table
 * bound); negative on bad arguments. */
int fs2t_conv_gemm_p_units(int n_real, int ntn, int G, int order, int ks, int nkc, int tks_max, int b, int* out, int max_units);
/* the launcher's bound: most units any workgroup holds under tile order `order` */
int fs2t_conv_gemm_p_max_units(int n_real, int ntn, int ks, int G, int order);
#ifdef __cplusplus
}
#endif
#endif
