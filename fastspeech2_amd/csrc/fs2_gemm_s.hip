// fs2_gemm_s.hip - STREAMING kernel for the one-tap bf16 contractions with K = 256 (gfx950, round 4): the weights live in REGISTERS.
//
//   Y[m][n] = epilogue( sum_k X[m][k] * W[n][k] ),  K = Cin = 256, N a multiple of 256
//   nn.Linear / Conv1d(k = 1) of the FFT blocks with d_model on the reduction side: transformer/SubLayers.py:39-41 (w_qs / w_ks /
//   w_vs as one N = 768 matrix), :54 (fc) and its data gradient, and the data gradient of :88 (w_2: N = 1024, ReLU gate).
//
// Why another kernel.  These launches are HBM streams with almost no arithmetic (fc: 22.7 MB in, 22.7 MB out, 5.8 GFLOP), and
// round 3's wide kernel (fs2_gemm_w.hip, 256 x 256 tiles) runs them at 1.4-2.8 TB/s: 174 tiles for 256 CUs, ONE tile per
// workgroup - its operand loads, its MFMAs and its 128 KB of stores happen one after the other, nothing to pipeline against
// (profiles/r03w: 0.37 of the HBM roof, "every workgroup's first tile spends 57 k cycles in store back-pressure").  With K = 256 a
// wave can hold its whole weight slice in registers: 64 output columns x 256 k = 32 fragments of 4 registers = 128 VGPRs.  So:
//   * a workgroup owns 256 output columns (EIGHT consumer waves x 32 columns: 64 registers of weights each) and walks 64-row tiles
//     of X: the weights are fetched ONCE per workgroup, straight into registers, and never touch LDS; X goes through a 4-deep LDS
//     ring (32 KB per tile) by LDS-DMA;
//   * per tile a consumer wave reads 32 activation fragments and issues 32 MFMAs (no weight reads at all), then stores its
//     64 x 32 outputs from registers.  Two consumer waves share a SIMD: one's epilogue (residual / bias loads, stores) runs under
//     the other's MFMAs - the first cut, four 64-column waves alone on their SIMDs, paid every epilogue load's latency on the
//     tile's critical path and was slower than the wide kernel (r04t_bench_w.log);
//   * the LDS-DMA is issued by TWO loader waves that do nothing else: global stores share vmcnt with loads and may retire out
//     of order with them, so a wave that both prefetches with counted vmcnt and stores every tile cannot count (fs2_gemm_t.hip
//     drains once per tile; here a tile is too short for that).  One raw barrier per tile.
// LDS rows are 512 bytes (256 channels); the 32 16-byte chunks of a row are XOR-swizzled with (row & 15), which puts the 16 lanes of
// every ds_read_b128 lane group ({0-3, 12-15, 20-27}, ...) on 16 different 16-byte bank groups.
#include "fs2_gemm.h"
#include "fs2_gemm_epi.h"

static constexpr int S_TM = 64, S_K = 256, S_NBUF = 4;
static constexpr int S_TILE_BYTES = S_TM * S_K * 2;      // 32 KB
static constexpr int S_LDS = S_NBUF * S_TILE_BYTES;      // 128 KB
static constexpr int S_NLOAD = 2;                        // loader waves
static constexpr int S_CONS = 8;                         // consumer waves (32 output columns each)
static constexpr int S_THREADS = (S_CONS + S_NLOAD) * 64;

template <int N> __device__ __forceinline__ void s_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }
__device__ __forceinline__ void s_barrier_mem() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// epilogue of one wave: 64 rows x 32 columns from the transposed accumulators (layout as p_epilogue, one column block)
template <int ACT>
__device__ __forceinline__ void s_epilogue(const ConvGemmArgs& a, f32x16 (&acc)[2][1], int m0, int nbase, int fl, int fh) {
    bf16_t* Y = reinterpret_cast<bf16_t*>(a.Y);
    const bf16_t* R = reinterpret_cast<const bf16_t*>(a.R);
    const bool gate = a.act == FS2_ACT_GATE;
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
        const int m = m0 + mb * 32 + fl;
        const bool rowok = m < a.M;
        bool padrow = false;
        if (a.lens && rowok) { const int b = m / a.S; padrow = (m - b * a.S) >= a.lens[b]; }
        bf16_t* yrow = Y + (size_t)m * a.ldy;
        const bf16_t* rrow = R ? R + (size_t)m * a.ldr : nullptr;
#pragma unroll
        for (int nb = 0; nb < 1; ++nb) {
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
                const int n = nbase + nb * 32 + ch * 16 + fh * 8;
                if (!rowok) continue;
                float v[8];
                if (a.bias) {
                    const float4 b0 = *reinterpret_cast<const float4*>(a.bias + n), b1 = *reinterpret_cast<const float4*>(a.bias + n + 4);
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
                        const float r0 = __uint_as_float(u[e] << 16), r1 = __uint_as_float(u[e] & 0xffff0000u);
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
                uint4 o;
                uint32_t* ou = reinterpret_cast<uint32_t*>(&o);
#pragma unroll
                for (int e = 0; e < 4; ++e) ou[e] = pack_bf16x2(v[2 * e], v[2 * e + 1]);
                *reinterpret_cast<uint4*>(yrow + n) = o;
            }
        }
    }
}

// grid (stripes, N / 256): workgroup (x, y) walks the 64-row tiles x, x + stripes, ... of column group y
__global__ void __launch_bounds__(S_THREADS) conv_gemm_s_kernel(ConvGemmArgs a, int ntiles) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int stripes = gridDim.x;
    const int my_tiles = (ntiles - (int)blockIdx.x + stripes - 1) / stripes;      // >= 1: the launcher keeps stripes <= ntiles
    const unsigned smem_u = lds_addr(smem);

    if (wave >= S_CONS) {
        // ------------------------------------------------------------------ loader waves: LDS-DMA only, counted vmcnt
        const int lw = wave - S_CONS;
        const unsigned char* Xb = reinterpret_cast<const unsigned char*>(a.X);
        const unsigned ldx2 = (unsigned)(a.ldx * 2);
        const int hi = lane >> 5, pos = lane & 31;
        auto issue = [&](int t) {                            // tile t of this workgroup -> buffer t % 3; 16 pieces of 2 rows
            const int m0 = ((int)blockIdx.x + t * stripes) * S_TM;
            const unsigned dst0 = smem_u + (unsigned)((t % S_NBUF) * S_TILE_BYTES);
#pragma unroll
            for (int i = 0; i < 32 / S_NLOAD; ++i) {
                const int j = lw + S_NLOAD * i;              // piece: rows 2j, 2j + 1
                const int row = 2 * j + hi;
                const int g = min(m0 + row, a.M - 1);
                const unsigned voff = (unsigned)g * ldx2 + (unsigned)((pos ^ (row & 15)) << 4);
                glds16_sbase(voff, Xb, __builtin_amdgcn_readfirstlane(dst0 + (unsigned)(j * 1024)));
            }
        };
        issue(0);
        if (my_tiles > 1) issue(1);
        if (my_tiles > 2) issue(2);
        for (int t = 0; t < my_tiles; ++t) {
            // my pieces of tile t have landed: at most the tiles behind it (two, one or none) may still be in flight
            if (t + 2 < my_tiles) s_wait_vm<2 * (32 / S_NLOAD)>();
            else if (t + 1 < my_tiles) s_wait_vm<32 / S_NLOAD>();
            else s_wait_vm<0>();
            s_barrier_mem();                                 // tile t published; the buffer of tile t - 1 released
            if (t + 3 < my_tiles) issue(t + 3);
        }
        return;
    }

    // ------------------------------------------------------------------ consumer waves
    const int fl = lane & 31, fh = lane >> 5;
    const int nbase = (int)blockIdx.y * 256 + wave * 32;
    // the wave's weight slice, once, straight into registers: B fragment ks = rows nbase + fl, k = 16 ks + 8 fh .. + 8
    u32x4 Wf[16];
    {
        const unsigned char* wrow = reinterpret_cast<const unsigned char*>(a.W) + (size_t)(nbase + fl) * (size_t)(a.ldw * 2) + fh * 16;
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) Wf[ks] = *reinterpret_cast<const u32x4*>(wrow + ks * 32);
    }
    const unsigned arow = (unsigned)(fl * 512);              // byte offset of the lane's row (second row block: + 32 rows)
    const unsigned akey = (unsigned)(fl & 15);
    for (int t = 0; t < my_tiles; ++t) {
        s_barrier_mem();                                     // tile t has landed (all loader pieces)
        const unsigned char* buf = smem + (t % S_NBUF) * S_TILE_BYTES;
        f32x16 acc[2][1];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][0][r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            const unsigned coff = ((unsigned)(2 * ks + fh) ^ akey) << 4;
            const u32x4 a0 = *reinterpret_cast<const u32x4*>(buf + arow + coff);
            const u32x4 a1 = *reinterpret_cast<const u32x4*>(buf + arow + 32 * 512 + coff);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, Wf[ks]), __builtin_bit_cast(bf16x8, a0), acc[0][0], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, Wf[ks]), __builtin_bit_cast(bf16x8, a1), acc[1][0], 0, 0, 0);
        }
        const int m0 = ((int)blockIdx.x + t * stripes) * S_TM;
        FS2_ACT_DISPATCH(a.act, (s_epilogue<ACT>(a, acc, m0, nbase, fl, fh)));
    }
}

// ------------------------------------------------------------------------------------------------ launcher
static int s_cu_count() {
    static int cus[64] = {0};
    int d = 0;
    (void)hipGetDevice(&d);
    d &= 63;
    if (!cus[d]) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, d) != hipSuccess || n <= 0) n = 256;
        cus[d] = n;
    }
    return cus[d];
}

// Eligibility (pure function of the launch description): one tap, K = 256 exactly, N a multiple of 256, 16-byte rows; at least as
// many 64-row tiles as would give every CU one (fewer: the small-M kernels keep the launch)
bool fs2_conv_gemm_s_ok(const ConvGemmArgs& a, int dtype) {
    if (dtype != FS2_BF16 || a.taps != 1 || a.in_act != FS2_ACT_NONE) return false;
    if (a.Cin != S_K || a.N % 256 != 0 || a.N > 256 * 64 || !a.vec_ok || a.ldx % 8 != 0) return false;
    if ((double)a.M * a.ldx * 2 >= 4.0e9) return false;
    static const int on = fs2_dev_env("FS2_GEMM_S", 1);               // dev A/B: 0 = off
    if (!on) return false;
    return (long)fs2_cdiv(a.M, S_TM) * (a.N / 256) >= s_cu_count();
}

void fs2_conv_gemm_s_launch(const ConvGemmArgs& a, hipStream_t stream) {
    static Fs2DevOnce once;
    once.run([&] { (void)hipFuncSetAttribute((const void*)conv_gemm_s_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, S_LDS); });
    const int ntiles = fs2_cdiv(a.M, S_TM), groups = a.N / 256;
    int stripes = s_cu_count() / groups;                    // one workgroup per CU (96 KB of LDS + 256 registers per wave)
    if (stripes < 1) stripes = 1;
    if (stripes > ntiles) stripes = ntiles;
    conv_gemm_s_kernel<<<dim3((unsigned)stripes, (unsigned)groups), S_THREADS, S_LDS, stream>>>(a, ntiles);
}
