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

// gemm_res_ln (N == 256: a workgroup owns whole rows): the epilogue of the projection IS the LayerNorm kernel
//   z = dropout(acc + bias) + res  (stored bf16 into Y: saved for backward)    out = mask(LN(z) * gamma + beta)    mean / rstd saved
// == fs2_conv_gemm followed by fs2_ln_fwd (transformer/SubLayers.py:54-55, 90-93 + Layers.py:25,28) without the projection's
// output making a round trip through HBM (one bf16 rounding less: y is never stored) and without the second launch.
// (struct WLn: fs2_gemm.h - shared with the streaming kernel's LayerNorm epilogue)

// Layout after the half-wave swaps: lane (fl, fh) of wave (wm, wn) holds, for its two rows m = m0 + 64 wm + 32 mb + fl, the
// columns 128 wn + 32 nb + 16 ch + 8 fh + [0, 8) as v[mb][nb][8 ch + e].  A row is spread over the lane pair (fh = 0, 1) and the
// wave pair (wn = 0, 1): row sums = lane sum + one cross-half shuffle + one exchange through LDS (red: [2][8 waves][64 rows]).
__device__ __forceinline__ void w_epilogue_resln(const ConvGemmArgs& a, const WLn& ln, f32x16 (&acc)[2][4], int m0, int wave, int wm, int wn,
                                                 int fl, int fh, const int32_t* lens_s, float* red) {
    bf16_t* Z = reinterpret_cast<bf16_t*>(a.Y);
    bf16_t* O = reinterpret_cast<bf16_t*>(ln.out);
    const bf16_t* R = reinterpret_cast<const bf16_t*>(a.R);
    const uint64_t seed = ln.seed_pre + (ln.seed_dev ? *ln.seed_dev : 0ull);
    const float ik = ln.p_pre > 0.f ? 1.f / (1.f - ln.p_pre) : 1.f;
    float rsum[2] = {0.f, 0.f};
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
        const int m = m0 + wm * 64 + mb * 32 + fl;
        const bool rowok = m < a.M;
        const int mc = rowok ? m : a.M - 1;
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
            float v[16];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                u32x2 s0 = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[mb][nb][e]), __float_as_uint(acc[mb][nb][4 + e]), false, false);
                u32x2 s1 = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[mb][nb][8 + e]), __float_as_uint(acc[mb][nb][12 + e]), false, false);
                v[e] = __uint_as_float(s0[0]); v[4 + e] = __uint_as_float(s0[1]);
                v[8 + e] = __uint_as_float(s1[0]); v[12 + e] = __uint_as_float(s1[1]);
            }
#pragma unroll
            for (int ch = 0; ch < 2; ++ch) {
                const int n = wn * 128 + nb * 32 + ch * 16 + fh * 8;
                float x[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) x[e] = v[ch * 8 + e];
                if (a.bias) {
                    const float4 b0 = *reinterpret_cast<const float4*>(a.bias + n), b1 = *reinterpret_cast<const float4*>(a.bias + n + 4);
                    x[0] += b0.x; x[1] += b0.y; x[2] += b0.z; x[3] += b0.w; x[4] += b1.x; x[5] += b1.y; x[6] += b1.z; x[7] += b1.w;
                }
                if (ln.p_pre > 0.f) {
                    const uint32_t e0 = (uint32_t)m * 256u + (uint32_t)n;
#pragma unroll
                    for (int e = 0; e < 8; ++e) x[e] *= fs2_drop_scale(seed, e0 + e, ln.p_pre, ik);
                }
                if (R) {
                    const uint4 rr = *reinterpret_cast<const uint4*>(R + (size_t)mc * a.ldr + n);
                    const uint32_t* u = reinterpret_cast<const uint32_t*>(&rr);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { x[2 * e] += __uint_as_float(u[e] << 16); x[2 * e + 1] += __uint_as_float(u[e] & 0xffff0000u); }
                }
                uint4 zq;
                uint32_t* zu = reinterpret_cast<uint32_t*>(&zq);
#pragma unroll
                for (int e = 0; e < 4; ++e) zu[e] = pack_bf16x2(x[2 * e], x[2 * e + 1]);
                if (rowok) *reinterpret_cast<uint4*>(Z + (size_t)m * a.ldy + n) = zq;
#pragma unroll
                for (int e = 0; e < 4; ++e) {            // statistics on the values as stored: backward sees the same z
                    const float lo = __uint_as_float(zu[e] << 16), hi = __uint_as_float(zu[e] & 0xffff0000u);
                    acc[mb][nb][ch * 8 + 2 * e] = lo; acc[mb][nb][ch * 8 + 2 * e + 1] = hi;
                    rsum[mb] += lo + hi;
                }
            }
        }
    }
    // ---- row means: lane pair, then wave pair through LDS
    float mean[2], rstd[2];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
        rsum[mb] += __shfl_xor(rsum[mb], 32, 64);
        if (fh == 0) red[wave * 64 + mb * 32 + fl] = rsum[mb];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) mean[mb] = (rsum[mb] + red[(wave ^ 1) * 64 + mb * 32 + fl]) * (1.f / 256.f);
    float rsq[2] = {0.f, 0.f};
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) { const float d = acc[mb][nb][r] - mean[mb]; rsq[mb] += d * d; }
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
        rsq[mb] += __shfl_xor(rsq[mb], 32, 64);
        if (fh == 0) red[512 + wave * 64 + mb * 32 + fl] = rsq[mb];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) rstd[mb] = rsqrtf((rsq[mb] + red[512 + (wave ^ 1) * 64 + mb * 32 + fl]) * (1.f / 256.f) + ln.eps);
    // ---- normalise, scale, mask, store
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
        const int m = m0 + wm * 64 + mb * 32 + fl;
        if (m >= a.M) continue;
        bool padrow = false;
        if (a.lens) { const int b = m / a.S; padrow = (m - b * a.S) >= lens_s[b]; }
        if (wn == 0 && fh == 0) { ln.mean[m] = mean[mb]; ln.rstd[m] = rstd[mb]; }
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
#pragma unroll
            for (int ch = 0; ch < 2; ++ch) {
                const int n = wn * 128 + nb * 32 + ch * 16 + fh * 8;
                const float4 g0 = *reinterpret_cast<const float4*>(ln.gamma + n), g1 = *reinterpret_cast<const float4*>(ln.gamma + n + 4);
                const float4 b0 = *reinterpret_cast<const float4*>(ln.beta + n), b1 = *reinterpret_cast<const float4*>(ln.beta + n + 4);
                const float gm[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
                const float bt[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
                float o[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = padrow ? 0.f : (acc[mb][nb][ch * 8 + e] - mean[mb]) * rstd[mb] * gm[e] + bt[e];
                uint4 oq;
                uint32_t* ou = reinterpret_cast<uint32_t*>(&oq);
#pragma unroll
                for (int e = 0; e < 4; ++e) ou[e] = pack_bf16x2(o[2 * e], o[2 * e + 1]);
                *reinterpret_cast<uint4*>(O + (size_t)m * ln.ldo + n) = oq;
            }
    }
}

template <bool RESLN>
__device__ __forceinline__ void conv_gemm_w_body(const ConvGemmArgs& a, const WSched& sc, const WLn& ln, unsigned char* smem) {
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
                if (m < a.M) {
                    *reinterpret_cast<uint4*>(Y + (size_t)m * a.ldy + n) = make_uint4(0, 0, 0, 0);
                    if (RESLN) *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(ln.out) + (size_t)m * ln.ldo + n) = make_uint4(0, 0, 0, 0);
                }
            }
            // (gemm_res_ln: FINITE statistics for the rows nobody normalises - LayerNorm's backward multiplies them by a zero
            // gradient, and 0 x whatever torch.empty left there would put NaNs into padded rows that the k = 9 data gradient reads)
            if (RESLN)
                for (int i = tid; i < 256; i += 512)
                    if (m0 + i < a.M) { ln.mean[m0 + i] = 0.f; ln.rstd[m0 + i] = 0.f; }
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
            if (RESLN) {
                w_epilogue_resln(a, ln, acc, q[0].mt * 256, wave, wm, wn, fl, fh, lens_s, reinterpret_cast<float*>(smem + W_AUX + W_MAXB * 4));
            } else {
                FS2_ACT_DISPATCH(a.act, (p_epilogue<ACT, 2>(a, acc, q[0].mt * 256, q[0].nt * 256 + wn * 128, wm, fl, fh, lens_s, bias_s, lane)));
            }
            zero_acc();
        }
        q[0] = q[1]; q[1] = q[2]; q[2] = nxt;
        buf = (buf + 1) & 3;
    }
}

template <bool RESLN>
__global__ void __launch_bounds__(512, 2) conv_gemm_w_kernel(ConvGemmArgs a, WSched sc, WLn ln) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    conv_gemm_w_body<RESLN>(a, sc, ln, smem);
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

static void w_launch(const ConvGemmArgs& a, const int32_t* tile_map, const WLn& ln, hipStream_t stream) {
    static Fs2DevOnce once;
    once.run([&] {
        (void)hipFuncSetAttribute((const void*)conv_gemm_w_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, W_LDS);
        (void)hipFuncSetAttribute((const void*)conv_gemm_w_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, W_LDS);
    });
    WSched sc;
    sc.ntm = fs2_cdiv(a.M, 256);
    sc.ntn = a.N / 256;
    sc.tmap = a.lens ? tile_map : nullptr;
    int G = w_cu_count() & ~7;
    const long tiles = (long)sc.ntm * sc.ntn;
    if (tiles < G) G = (int)((tiles + 7) & ~7L);
    sc.G = G;
    if (ln.gamma) conv_gemm_w_kernel<true><<<(unsigned)G, 512, W_LDS, stream>>>(a, sc, ln);
    else conv_gemm_w_kernel<false><<<(unsigned)G, 512, W_LDS, stream>>>(a, sc, ln);
}

void fs2_conv_gemm_w_launch(const ConvGemmArgs& a, const int32_t* tile_map, hipStream_t stream) {
    WLn ln = {};
    w_launch(a, tile_map, ln, stream);
}

// defined in fs2_gemm_s.hip
bool fs2_conv_gemm_s_ln_ok(const ConvGemmArgs& a, int dtype);
void fs2_conv_gemm_s_ln_launch(const ConvGemmArgs& a, const WLn& ln, hipStream_t stream);

// ---- gemm_res_ln: Linear (N = 256) + dropout + residual + LayerNorm + pad-row zero in one launch (SURVEY §8(b) export list)
extern "C" int fs2_gemm_res_ln_supported(int M, int N, int Cin, int S, int dtype) {
    if (dtype != FS2_BF16 || N != 256 || Cin % 32 != 0 || Cin < 128 || M <= 0 || S <= 0 || M % S != 0) return 0;
    if ((double)M * Cin * 2 >= 4.0e9 || M / S > W_MAXB) return 0;
    return fs2_cdiv(M, 256) >= 16 ? 1 : 0;                // fewer tiles: the separate small-tile kernels keep more CUs busy
}

// 1 when fs2_gemm_res_ln_fwd would run this shape on the streaming K = 256 kernel (fs2_gemm_s.hip) - the form that is FASTER than
// the two launches at the train step's decoder shapes; 0: the wide-tile form (or unsupported).  A flag, not a status.
extern "C" int fs2_gemm_res_ln_streams(int M, int N, int Cin, int S, int dtype) {
    if (!fs2_gemm_res_ln_supported(M, N, Cin, S, dtype)) return 0;
    ConvGemmArgs a = {};
    a.ldx = Cin; a.ldw = Cin; a.ldy = N; a.ldr = N; a.M = M; a.N = N; a.Cin = Cin; a.S = S; a.taps = 1; a.dil = 1; a.act = FS2_ACT_NONE;
    a.in_act = FS2_ACT_NONE; a.out_scale = 1.f; a.vec_ok = 1;
    a.lens = reinterpret_cast<const int32_t*>(1);          // (a launch with lens: the batch-size bound of the LDS lengths table applies)
    return fs2_conv_gemm_s_ln_ok(a, dtype) ? 1 : 0;
}

extern "C" int fs2_gemm_res_ln_fwd(const void* X, long ldx, const void* Wpacked, const float* bias, const void* R, long ldr, void* Z, long ldz,
                                   void* out, long ldo, const float* gamma, const float* beta, float* mean, float* rstd, const int32_t* lens,
                                   const int32_t* tile_map, int M, int N, int Cin, int S, float eps, float p_pre, uint64_t seed_pre,
                                   const uint64_t* seed_dev, int dtype, hipStream_t stream) {
    FS2_CHECK_ARG(X && Wpacked && Z && out && gamma && beta && mean && rstd, "gemm_res_ln: null pointer");
    FS2_CHECK_ARG(fs2_gemm_res_ln_supported(M, N, Cin, S, dtype), "gemm_res_ln: unsupported shape M=%d N=%d Cin=%d S=%d dtype=%d", M, N, Cin, S, dtype);
    FS2_CHECK_ARG(ldx % 8 == 0 && ldz % 8 == 0 && ldo % 8 == 0 && (!R || ldr % 8 == 0) && p_pre >= 0.f && p_pre < 1.f, "gemm_res_ln: bad strides / p");
    FS2_CHECK_ARG((((uintptr_t)X | (uintptr_t)Wpacked | (uintptr_t)Z | (uintptr_t)out | (uintptr_t)R | (uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)bias) & 15) == 0,
                  "gemm_res_ln: operands must be 16-byte aligned");
    ConvGemmArgs a = {};
    a.X = X; a.ldx = ldx; a.W = Wpacked; a.ldw = Cin; a.bias = bias; a.R = R; a.ldr = ldr; a.Y = Z; a.ldy = ldz; a.lens = lens;
    a.M = M; a.N = N; a.Cin = Cin; a.S = S; a.taps = 1; a.dil = 1; a.pad = 0; a.act = FS2_ACT_NONE; a.slope = 0.f; a.in_act = FS2_ACT_NONE;
    a.in_slope = 0.f; a.accumulate = 0; a.out_scale = 1.f; a.vec_ok = 1; a.dbg = 0;
    WLn ln;
    ln.gamma = gamma; ln.beta = beta; ln.out = out; ln.ldo = ldo; ln.mean = mean; ln.rstd = rstd; ln.eps = eps; ln.p_pre = p_pre;
    ln.seed_pre = seed_pre; ln.seed_dev = seed_dev;
    // K = 256 with enough 64-row tiles to give every CU one: the streaming kernel (weights in registers) with the same epilogue
    if (fs2_conv_gemm_s_ln_ok(a, dtype)) fs2_conv_gemm_s_ln_launch(a, ln, stream);
    else w_launch(a, tile_map, ln, stream);
    FS2_CHECK_LAUNCH("gemm_res_ln");
    return FS2_OK;
}
