// fs2_gemm_w.hip - WIDE-tile persistent kernel for the one-tap bf16 contractions with N a multiple of 256 (gfx950, round 3).
//
//   Y[m][n] = epilogue( sum_k X[m][k] * W[n][k] )      nn.Linear / Conv1d(k = 1) forward and data gradient:
//   transformer/SubLayers.py:39-41,54 (w_qs / w_ks / w_vs as one N = 768 matrix, fc), :88 (w_2), and their autograd transposes.
//
// Why another kernel.  These launches are not MFMA-bound: at K = 256 .. 1024 a 256 x 128 output tile of the persistent kernel
// (fs2_gemm_p.hip) needs 192 .. 768 KB of operands for 1.7 .. 6.8 us of MFMA work, and what a CU can pull through LDS-DMA is set by
// the bytes it keeps in flight (Little's law: ~60 KB of ring at ~2 us = ~30 GB/s per CU; r02 PMC: 10 % MFMA busy, waves parked 61 %,
// 2.9 TB/s = 0.36 of the HBM roof).  Two things follow, and this kernel does both:
//   * FEWER BYTES PER OUTPUT: a 256 x 256 tile fetches (256 + 256) K per 65 536 outputs where 256 x 128 fetches (256 + 128) K per
//     32 768 - one third less L2 -> LDS traffic for the same result (QKV: the activation tile is fetched 3x instead of 6x);
//   * MORE BYTES IN FLIGHT: no loader waves and no per-wave accumulator limit on the ring - all 8 waves are consumers (64 x 128
//     each, the same transposed accumulators and register epilogue as the persistent kernel) and every wave issues its own four
//     1 KiB pieces of the K-step THREE steps ahead into a ring of four 32 KiB buffers: 96 KB in flight per CU.
// K-steps are 32 deep (64-byte rows in LDS, two MFMA k-slices); the 16-byte chunks of a row are XOR-swizzled with ((row >> 2) & 3)
// so that the 16 lanes of a ds_read_b128 service group ({0-3, 12-15, 20-27}, ...) hit 16 distinct bank quads; LDS-DMA writes
// lane-linearly, so the permutation is applied to each lane's SOURCE chunk.  One raw barrier per K-step, counted vmcnt, the
// K-step sequence runs on across tile boundaries (the next tile's operands are in flight during the epilogue).
// Tiles are dealt statically, per XCD and N-fastest (the workgroups of one XCD walk the N-tiles of the same M-tile together);
// fully padded M-tiles (fs2_tile_map) are left out of the deal and zero-filled.
#include "fs2_gemm.h"
#include "fs2_gemm_epi.h"

static constexpr int W_STEP_BYTES = 2 * 256 * 64;        // A [256 rows][64 B] + B [256 rows][64 B]
static constexpr int W_NBUF = 4;                        // (the pipeline code assumes 4: (buf + 3) & 3)
static constexpr int W_MAXB = 1024;                      // lens[] staged in LDS
static constexpr int W_AUX = W_NBUF * W_STEP_BYTES;      // lens (4 KB) + one 128-float bias line per wave (4 KB)
static constexpr int W_LDS = W_AUX + W_MAXB * 4 + 8 * 512;

struct WSched {
    int G;                 // workgroups (a multiple of 8)
    int ntm, ntn;          // 256-row M-tiles (all), 256-column N-tiles
    const int32_t* tmap;   // fs2_tile_map(rows = 256): [0] = n_real, [1..] real M-tiles then padded ones; null = all real
};

template <int N> __device__ __forceinline__ void w_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

struct WPos { int k, s, mt, nt; };                       // k-th unit of this workgroup (k < 0: none), K-step s of it, its tile

template <int ACT_UNUSED>
__device__ __forceinline__ void conv_gemm_w_body(const ConvGemmArgs& a, const WSched& sc, unsigned char* smem) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;             // wave tile: rows 64 wm .., columns 128 wn ..
    const int fl = lane & 31, fh = lane >> 5;
    const unsigned smem_u = lds_addr(smem);
    int32_t* lens_s = reinterpret_cast<int32_t*>(smem + W_AUX);
    float* bias_s = reinterpret_cast<float*>(smem + W_AUX + W_MAXB * 4) + wave * 128;
    const int b = blockIdx.x, x = b & 7, j = b >> 3, Gg = sc.G >> 3;
    const int n_real = sc.tmap ? __builtin_amdgcn_readfirstlane(sc.tmap[0]) : sc.ntm;
    const int nx = n_real > x ? (n_real - x + 7) >> 3 : 0;                  // real M-tiles of this XCD's share (mi = x mod 8)
    const int units = nx * sc.ntn;                                           // ... times N-tiles: dealt to the XCD's Gg workgroups
    const int nk = a.Cin >> 5;                                               // K-steps per tile

    if (a.lens) {
        const int B = a.M / a.S;
        for (int i = tid; i < B; i += 512) lens_s[i] = a.lens[i];
    }
    // fully padded M-tiles: zeros (every workgroup takes its share)
    if (sc.tmap && !a.accumulate) {
        bf16_t* Y = reinterpret_cast<bf16_t*>(a.Y);
        const int n_pad = sc.ntm - n_real;
        const int cpr = a.N >> 3;                                            // 16-byte chunks per row
        for (int p = b; p < n_pad; p += sc.G) {
            const int m0 = sc.tmap[1 + n_real + p] * 256;
            for (int i = tid; i < 256 * cpr; i += 512) {
                const int m = m0 + i / cpr, n = (i % cpr) * 8;
                if (m < a.M) *reinterpret_cast<uint4*>(Y + (size_t)m * a.ldy + n) = make_uint4(0, 0, 0, 0);
            }
        }
    }
    __syncthreads();                                                         // lens staged (read in the epilogues)

    auto tile_of = [&](int k, int& mt, int& nt) {                            // k-th unit of this workgroup: position j + k Gg of the XCD's list
        const int p = j + k * Gg;
        const int mil = p / sc.ntn;
        nt = p - mil * sc.ntn;
        const int mi = x + 8 * mil;
        mt = sc.tmap ? __builtin_amdgcn_readfirstlane(sc.tmap[1 + mi]) : mi;
    };
    auto first_pos = [&]() -> WPos {
        WPos q = {-1, 0, 0, 0};
        if (j < units) { q.k = 0; tile_of(0, q.mt, q.nt); }
        return q;
    };
    auto next_pos = [&](WPos q) -> WPos {
        if (q.k < 0) return q;
        if (++q.s < nk) return q;
        q.s = 0;
        q.k += 1;
        if (j + q.k * Gg < units) tile_of(q.k, q.mt, q.nt); else q.k = -1;
        return q;
    };

    // ---- DMA: a 1 KiB piece = 16 rows x 64 B; lane l writes (row l >> 2, chunk position l & 3) which holds source chunk
    // (l & 3) ^ ((row >> 2) & 3) = (l & 3) ^ ((l >> 4) & 3).  Wave w issues A pieces w, w + 8 and B pieces w, w + 8.
    const int prow = lane >> 2;
    const unsigned pchunk = (unsigned)(((lane & 3) ^ ((lane >> 4) & 3)) << 4);
    const unsigned char* Xb = reinterpret_cast<const unsigned char*>(a.X);
    const unsigned char* Wb = reinterpret_cast<const unsigned char*>(a.W);
    const unsigned x_rs = (unsigned)a.ldx * 2u, w_rs = (unsigned)a.ldw * 2u;
    auto issue = [&](const WPos& q, int buf) {
        const unsigned dst = __builtin_amdgcn_readfirstlane(smem_u + (unsigned)(buf * W_STEP_BYTES + wave * 1024));
        const unsigned char* xa = Xb + (size_t)q.s * 64;                     // (uniform: scalar base of the K-step)
        const unsigned char* wa = Wb + (size_t)q.s * 64;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = min(q.mt * 256 + 16 * (wave + 8 * i) + prow, a.M - 1);
            glds16_sbase((unsigned)m * x_rs + pchunk, xa, dst + (unsigned)(i * 8192));
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int n = min(q.nt * 256 + 16 * (wave + 8 * i) + prow, a.N - 1);
            glds16_sbase((unsigned)n * w_rs + pchunk, wa, dst + (unsigned)(16384 + i * 8192));
        }
    };

    // ---- fragments: lane (fl, fh) reads 16-byte chunk c = 2 slice + fh of its row; chunk position = c ^ key, key = (fl >> 2) & 3
    const unsigned key = (unsigned)((fl >> 2) & 3);
    const unsigned aoff = (unsigned)((wm * 64 + fl) * 64) + (((unsigned)fh ^ key) << 4);                  // slice 0; slice 1 = ^ 32
    const unsigned boff = 16384u + (unsigned)((wn * 128 + fl) * 64) + (((unsigned)fh ^ key) << 4);
    typedef __attribute__((address_space(3))) const u32x4* lds_v4;
    auto ldsr = [&](unsigned off) -> u32x4 { return *(lds_v4)(size_t)(smem_u + off); };

    f32x16 acc[2][4];
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int jn = 0; jn < 4; ++jn)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][jn][r] = 0.f;
    };
    zero_acc();
    auto kstep = [&](int buf) {
        const unsigned bufoff = (unsigned)(buf * W_STEP_BYTES);
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) {
            const unsigned ao = (aoff ^ (unsigned)(sl * 32)) + bufoff, bo = (boff ^ (unsigned)(sl * 32)) + bufoff;
            u32x4 af[2], bf[4];
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) af[mb] = ldsr(ao + mb * 2048);
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) bf[nb] = ldsr(bo + nb * 2048);
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int nb = 0; nb < 4; ++nb)
                    acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, bf[nb]), __builtin_bit_cast(bf16x8, af[mb]),
                                                                         acc[mb][nb], 0, 0, 0);
        }
    };

    // ---- pipeline: step i lives in buffer i % 4, its DMA is issued 3 steps ahead.  (The buffer index is a run-time value: the
    // loop body exists ONCE - four unrolled copies would each carry the epilogue's instantiations.)
    WPos q[3];
    q[0] = first_pos();
    q[1] = next_pos(q[0]);
    q[2] = next_pos(q[1]);
#pragma unroll
    for (int i = 0; i < 3; ++i)
        if (q[i].k >= 0) issue(q[i], i);
    int buf = 0;
    while (q[0].k >= 0) {
        const WPos nxt = next_pos(q[2]);
        const int ahead = (q[1].k >= 0 ? 1 : 0) + (q[2].k >= 0 ? 1 : 0);
        if (ahead == 2) w_wait_vm<8>(); else if (ahead == 1) w_wait_vm<4>(); else w_wait_vm<0>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (nxt.k >= 0) issue(nxt, (buf + 3) & 3);
        kstep(buf);
        if (q[0].s == nk - 1) {                                              // the tile's last K-step: registers -> bf16 rows
            FS2_ACT_DISPATCH(a.act, (p_epilogue<ACT>(a, acc, q[0].mt * 256, q[0].nt * 256 + wn * 128, wm, fl, fh, lens_s, bias_s, lane)));
            zero_acc();
        }
        q[0] = q[1]; q[1] = q[2]; q[2] = nxt;
        buf = (buf + 1) & 3;
    }
}

__global__ void __launch_bounds__(512, 2) conv_gemm_w_kernel(ConvGemmArgs a, WSched sc) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    conv_gemm_w_body<0>(a, sc, smem);
}

static int w_cu_count() {
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

// Eligibility (a pure function of the launch description, shared with fs2_conv_gemm_variant)
bool fs2_conv_gemm_w_ok(const ConvGemmArgs& a, bool has_map, int dtype) {
    (void)has_map;
    if (dtype != FS2_BF16 || a.taps != 1 || a.in_act != FS2_ACT_NONE) return false;
    if (a.N % 256 != 0 || a.Cin % 32 != 0 || a.Cin < 128 || !a.vec_ok) return false;
    if ((double)a.M * a.ldx * 2 >= 4.0e9 || (double)a.N * a.ldw * 2 >= 4.0e9) return false;
    if (a.lens && a.M / a.S > W_MAXB) return false;
    const long tiles = (long)fs2_cdiv(a.M, 256) * (a.N / 256);
    return tiles >= 96;                                       // fewer: the 128^2 / persistent kernels fill more CUs
}

void fs2_conv_gemm_w_launch(const ConvGemmArgs& a, const int32_t* tile_map, hipStream_t stream) {
    static Fs2DevOnce once;
    once.run([&] { (void)hipFuncSetAttribute((const void*)conv_gemm_w_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, W_LDS); });
    WSched sc;
    sc.ntm = fs2_cdiv(a.M, 256);
    sc.ntn = a.N / 256;
    sc.tmap = a.lens ? tile_map : nullptr;
    int G = w_cu_count() & ~7;
    const long tiles = (long)sc.ntm * sc.ntn;
    if (tiles < G) G = (int)((tiles + 7) & ~7L);
    sc.G = G;
    conv_gemm_w_kernel<<<(unsigned)G, 512, W_LDS, stream>>>(a, sc);
}
