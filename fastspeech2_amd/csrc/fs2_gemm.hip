// fs2_gemm.hip — the MFMA contraction core of the hot path (gfx950).
//
// ONE implicit-GEMM kernel family covers every dense contraction of FastSpeech2 / HiFi-GAN / STFT:
//   Linear (taps=1), Conv1d k=3/5/7/9/11 with dilation (taps=k), their data gradients (same kernel,
//   tap-flipped packed weights), ConvTranspose1d (polyphase -> 3-tap conv with N = stride*Cout) and the
//   framed DFT (ldx = hop).  Reference call sites: transformer/SubLayers.py:39-41,54,87-88;
//   model/modules.py:209-240; model/fastspeech2.py:95; transformer/Layers.py:129-137;
//   hifigan/models.py:96-165; audio/stft.py:60-72.
//
// Layout: activations are time-major rows X[M][Cin] (M = B*S, row m = b*S + t), so a conv tap is a ROW
// shift and no transpose(1,2).contiguous() ever runs.  Weights are pre-packed K-contiguous W[N][taps][Cin]
// in the compute dtype ("NT" GEMM: both operands K-contiguous, identical staging for A and B).
//
// Tile: 128x128 per 256-thread workgroup (4 waves as 2x2, each wave 64x64 = 2x2 MFMA 32x32 blocks);
// K-tile = 128 bytes per row (32 f32 / 64 bf16).  LDS rows are 128 B, 16-B chunks XOR-swizzled with
// ((row>>1)&7) so both the ds_write_b128 staging and the ds_read_b128 fragment reads are conflict-free
// (guide §5.5 T2).  f32 path: v_mfma_f32_32x32x2_f32 (exact f32, 157 TF roof); bf16 path:
// v_mfma_f32_32x32x16_bf16 (2.5 PF roof), fp32 accumulate in both.
// Register-staged double buffering: global loads of K-tile t+1 are issued before the MFMAs of tile t and
// written to the other LDS buffer afterwards (one barrier per K-tile).
#include "fs2_common.h"
#include <stdlib.h>
#include <type_traits>

#include "fs2_gemm.h"
#include "fs2_sched.h"
#include "fs2_wgrad.h"

// final stage of the epilogue (shared by both GEMM kernels): the f32 tile staged in LDS is written out in whole
// 16-byte row segments; residual / ReLU-gate operand and the accumulate operand are read the same way.
// bf16 outputs, 128-float-wide staging tiles: a lane reads 8 consecutive floats as two ds_read_b128, lanes 32 bytes apart - with
// the natural layout every 16-lane service group of the read ({0-3, 12-15, 20-27}, ... - MI355X_MICROARCH.md, LDS table) covers
// half of the 16 bank quads twice (r02 PMC: 16.9 % of conv_gemm_kernel's LDS cycles, 6.9 % of the DMA kernel's).  The tile is
// therefore stored GRANULE-PERMUTED (granule = 4 floats): even granule 2i at position i, odd granule 2i+1 at 16 + ((i + 4) & 15):
// each read then touches 16 distinct quads, and a half-wave's ds_write_b32 of 32 consecutive floats still lands on 32 distinct
// banks (positions distinct mod 8).  Checked exhaustively against the guide's lane groups on the host.  fp32 outputs (one
// ds_read_b128 per lane, 16 bytes apart) are conflict-free in the natural layout and keep it.
template <typename T> __host__ __device__ __forceinline__ int fs2_tile_col128(int c) { return fs2_tile_col128_bytes((int)sizeof(T), c); }
template <typename T, int TW = 128, int NT = 256, int ROWS = 128>
__device__ __forceinline__ void gemm_store_tile(const ConvGemmArgs& a, const float* tile, int m0, int n0, int tid) {
    constexpr int EPC = MmaTraits<T>::EPC;
    T* Y = reinterpret_cast<T*>(a.Y);
    const T* R = reinterpret_cast<const T*>(a.R);
    constexpr int EPT = EPC;                 // elements per 16-byte global store
    constexpr int CPR = TW / EPT;            // 16-byte chunks per tile row ([128][TW] f32 tile, NT threads)
    const bool gate = a.act == FS2_ACT_GATE;
#pragma unroll 4
    for (int it = 0; it < (ROWS * CPR) / NT; ++it) {
        int idx = tid + NT * it;
        int rl = idx / CPR, cc = (idx % CPR) * EPT;
        int m = m0 + rl, n = n0 + cc;
        if (m >= a.M || n >= a.N) continue;
        float v[EPT];
#pragma unroll
        for (int e = 0; e < EPT; e += 4) {
            float4 t = *reinterpret_cast<const float4*>(tile + rl * TW + (TW == 128 ? fs2_tile_col128<T>(cc + e) : cc + e));
            v[e] = t.x; v[e + 1] = t.y; v[e + 2] = t.z; v[e + 3] = t.w;
        }
        bool padrow = false;
        if (a.lens) { int b = m / a.S; padrow = (m - b * a.S) >= a.lens[b]; }
        T* yp = Y + (size_t)m * a.ldy + n;
        const T* rp = R ? R + (size_t)m * a.ldr + n : nullptr;
        if (a.vec_ok && n + EPT <= a.N) {
            if (rp) {
                float rv[EPT];
                uint4 rr = *reinterpret_cast<const uint4*>(rp);
                if constexpr (sizeof(T) == 4) { const float* f = reinterpret_cast<const float*>(&rr); for (int e = 0; e < 4; ++e) rv[e] = f[e]; }
                else { const uint32_t* u = reinterpret_cast<const uint32_t*>(&rr); for (int e = 0; e < 4; ++e) { rv[2 * e] = __uint_as_float(u[e] << 16); rv[2 * e + 1] = __uint_as_float(u[e] & 0xffff0000u); } }
#pragma unroll
                for (int e = 0; e < EPT; ++e) {
                    if (a.res_unlrelu > 0.f) rv[e] = rv[e] > 0.f ? rv[e] : rv[e] * a.res_unlrelu;
                    v[e] = gate ? (rv[e] > 0.f ? v[e] : 0.f) : v[e] + rv[e];
                }
            }
#pragma unroll
            for (int e = 0; e < EPT; ++e) { v[e] *= a.out_scale; if (padrow) v[e] = 0.f; }
            if (a.accumulate) {
                uint4 yy = *reinterpret_cast<const uint4*>(yp);
                if constexpr (sizeof(T) == 4) { const float* f = reinterpret_cast<const float*>(&yy); for (int e = 0; e < 4; ++e) v[e] += f[e]; }
                else { const uint32_t* u = reinterpret_cast<const uint32_t*>(&yy); for (int e = 0; e < 4; ++e) { v[2 * e] += __uint_as_float(u[e] << 16); v[2 * e + 1] += __uint_as_float(u[e] & 0xffff0000u); } }
            }
            if (a.post_slope > 0.f) {
#pragma unroll
                for (int e = 0; e < EPT; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * a.post_slope;
            }
            uint4 o;
            if constexpr (sizeof(T) == 4) { float* f = reinterpret_cast<float*>(&o); for (int e = 0; e < 4; ++e) f[e] = v[e]; }
            else { uint32_t* u = reinterpret_cast<uint32_t*>(&o); for (int e = 0; e < 4; ++e) u[e] = pack_bf16x2(v[2 * e], v[2 * e + 1]); }
            *reinterpret_cast<uint4*>(yp) = o;
        } else {
            for (int e = 0; e < EPT && n + e < a.N; ++e) {
                float x = v[e];
                if (rp) {
                    float rv = Elem<T>::ld(rp + e);
                    if (a.res_unlrelu > 0.f) rv = rv > 0.f ? rv : rv * a.res_unlrelu;
                    x = gate ? (rv > 0.f ? x : 0.f) : x + rv;
                }
                x *= a.out_scale;
                if (padrow) x = 0.f;
                if (a.accumulate) x += Elem<T>::ld(yp + e);
                if (a.post_slope > 0.f) x = x > 0.f ? x : x * a.post_slope;
                Elem<T>::st(yp + e, x);
            }
        }
    }
}

template <typename T>
__global__ void __launch_bounds__(256, 2) conv_gemm_kernel(ConvGemmArgs a) {
    constexpr int EPC = MmaTraits<T>::EPC;
    constexpr int BK = 8 * EPC;
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * 2 * 128 * 128];  // [buf][A|B][128 rows][128 B]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    // M-fastest rasterisation: consecutive workgroups (which the dispatcher deals round-robin to the 8 XCDs) share
    // one weight slice (stays hot in every XCD's L2) and XCD x always sees M-tiles == x (mod 8), so its slice of the
    // activations is also L2-resident across the N sweep.
    const int ntm = (a.M + 127) >> 7;
    const int tile_m = blockIdx.x % ntm, tile_n = blockIdx.x / ntm;
    const int m0 = tile_m * 128, n0 = tile_n * 128;
    const T* X = reinterpret_cast<const T*>(a.X);
    const T* W = reinterpret_cast<const T*>(a.W);
    T* Y = reinterpret_cast<T*>(a.Y);
    const T* R = reinterpret_cast<const T*>(a.R);

    // fully padded M-tile inside one sequence: write zeros and leave.
    if (a.lens) {
        int mlast = min(m0 + 127, a.M - 1);
        int b0 = m0 / a.S, b1 = mlast / a.S;
        if (b0 == b1 && (m0 - b0 * a.S) >= a.lens[b0]) {
            if (!a.accumulate) {
                for (int i = tid; i < 128 * 128; i += 256) {
                    int r = i >> 7, c = i & 127;
                    int m = m0 + r, n = n0 + c;
                    if (m < a.M && n < a.N) Elem<T>::st(Y + (size_t)m * a.ldy + n, 0.f);
                }
            }
            return;
        }
    }

    // staging assignment: thread -> 16-B chunk sc of rows sr + 32 i
    const int sc = tid & 7, sr = tid >> 3;
    int a_t[4];   // time index of the A row within its sequence, or a large negative if m >= M
    size_t a_off[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int m = m0 + sr + 32 * i;
        a_t[i] = (m < a.M) ? (m % a.S) : -(1 << 28);
        a_off[i] = (size_t)m * a.ldx;
    }
    const int nkc = (a.Cin + BK - 1) / BK;
    const int nk = a.taps * nkc;

    auto load_tile = [&](int kt, uint4 (&ra)[4], uint4 (&rb)[4]) {
        int tap = kt / nkc, kc = kt - tap * nkc;
        int shift = tap * a.dil - a.pad;
        int col = kc * BK + sc * EPC;
        bool colok = col < a.Cin;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int ts = a_t[i] + shift;
            bool ok = colok && ts >= 0 && ts < a.S;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (ok) v = *reinterpret_cast<const uint4*>(X + (a_off[i] + (long)shift * a.ldx + col));
            ra[i] = v;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int n = n0 + sr + 32 * i;
            bool ok = colok && n < a.N;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (ok) v = *reinterpret_cast<const uint4*>(W + ((size_t)n * a.ldw + (size_t)tap * a.Cin + col));
            rb[i] = v;
        }
        if (a.in_act == FS2_ACT_LRELU) {
#pragma unroll
            for (int i = 0; i < 4; ++i) ra[i] = (sizeof(T) == 4) ? act_chunk_f32(ra[i], a.in_slope) : act_chunk_bf16(ra[i], a.in_slope);
        }
    };
    auto store_tile = [&](int buf, const uint4 (&ra)[4], const uint4 (&rb)[4]) {
        unsigned char* As = smem + buf * 32768;
        unsigned char* Bs = As + 16384;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int r = sr + 32 * i;
            int off = r * 128 + ((sc ^ ((r >> 1) & 7)) << 4);
            *reinterpret_cast<uint4*>(As + off) = ra[i];
            *reinterpret_cast<uint4*>(Bs + off) = rb[i];
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int fl = lane & 31, fh = lane >> 5;
    auto compute = [&](int buf) {
        const unsigned char* As = smem + buf * 32768;
        const unsigned char* Bs = As + 16384;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            uint4 af[2], bf[2];
            int c = fh * 4 + j;
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
                int r = wm * 64 + mb * 32 + fl;
                af[mb] = *reinterpret_cast<const uint4*>(As + r * 128 + ((c ^ ((r >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                int r = wn * 64 + nb * 32 + fl;
                bf[nb] = *reinterpret_cast<const uint4*>(Bs + r * 128 + ((c ^ ((r >> 1) & 7)) << 4));
            }
            if constexpr (sizeof(T) == 4) {
#pragma unroll
                for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                        for (int nb = 0; nb < 2; ++nb)
                            acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(
                                reinterpret_cast<const float*>(&af[mb])[jj], reinterpret_cast<const float*>(&bf[nb])[jj],
                                acc[mb][nb], 0, 0, 0);
            } else {
#pragma unroll
                for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                    for (int nb = 0; nb < 2; ++nb)
                        acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                            __builtin_bit_cast(bf16x8, af[mb]), __builtin_bit_cast(bf16x8, bf[nb]), acc[mb][nb], 0, 0, 0);
            }
        }
    };

    // 3-stage software pipeline: two register sets hold K-tiles t+1 / t+2 in flight while tile t is consumed from LDS
    // (global loads get two tile-times to land; one barrier per K-tile).
    uint4 ra0[4], rb0[4], ra1[4], rb1[4];
    load_tile(0, ra0, rb0);
    if (nk > 1) load_tile(1, ra1, rb1);
    store_tile(0, ra0, rb0);
    __syncthreads();
    for (int kt = 0; kt < nk; kt += 2) {
        if (kt + 2 < nk) load_tile(kt + 2, ra0, rb0);
        compute(0);
        if (kt + 1 < nk) store_tile(1, ra1, rb1);
        __syncthreads();
        if (kt + 1 >= nk) break;
        if (kt + 3 < nk) load_tile(kt + 3, ra1, rb1);
        compute(1);
        if (kt + 2 < nk) store_tile(0, ra0, rb0);
        __syncthreads();
    }

    // ---- epilogue: bias + activation in registers, tile staged through LDS as f32 [128][128], then whole 16-byte
    // row segments are written (and the residual / gate operand read) coalesced.
    // C layout: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
    float* tile = reinterpret_cast<float*>(smem);
    auto stage = [&](auto actc) {
        constexpr int ACT = decltype(actc)::value;
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            int cl = wn * 64 + nb * 32 + fl;
            int n = n0 + cl;
            float bv = (a.bias && n < a.N) ? a.bias[n] : 0.f;
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    int rl = wm * 64 + mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                    tile[rl * 128 + fs2_tile_col128<T>(cl)] = act_ct<ACT>(acc[mb][nb][r] + bv, a.slope);
                }
        }
    };
    FS2_ACT_DISPATCH(a.act, stage(std::integral_constant<int, ACT>{}));
    __syncthreads();
    gemm_store_tile<T>(a, tile, m0, n0, tid);
}

// =====================================================================================================
// LDS-DMA variant (in_act == none, halo (taps-1)*dil <= 16): operands go HBM/L2 -> LDS with
// global_load_lds_dwordx4 (no VGPR round trip, no ds_write pass).  The DMA destination is wave-uniform base +
// lane*16 (lane-linear), so the bank swizzle is applied to the per-lane SOURCE chunk (guide rule 21): LDS slot
// (row r, chunk c) receives source chunk c ^ ((r>>1)&7) and fragment reads use the same XOR.
// For convolutions the activation tile is staged ONCE per Cin-chunk with a halo of (taps-1)*dil rows and every tap
// reads it at a row offset; only the weight tile changes per tap.  Taps that fall outside the row's own sequence
// are zeroed on the fragment (per-lane tap-validity bitmask, VALU work hidden under the MFMAs); rows outside
// [0, M) / columns outside Cin / N are fetched from a zero line.
__device__ __attribute__((aligned(128))) unsigned int fs2_zero_line[32];

// KS = 2: in-workgroup split-K.  512 threads = two 4-wave groups that walk the even / odd Cin chunks of the SAME 128x128
// tile through their own operand buffers (2 waves per SIMD) and add their accumulators through LDS before the epilogue.
// For long reductions with few output tiles - the encoder's k=9 data gradient is 96 tiles x 144 K-steps on 256 CUs,
// 185 us at one 4-wave workgroup per tile - this halves the time a tile occupies its CU.
template <typename T, int KS>
__global__ void __launch_bounds__(256 * KS, KS == 1 ? 2 : 1) conv_gemm_dma_kernel(ConvGemmArgs a) {
    constexpr int EPC = MmaTraits<T>::EPC;
    constexpr int BK = 8 * EPC;
    constexpr int A_ROWS = 160;                                  // 128 + halo (<= 16), rounded to 20 wave-loads
    constexpr int A_BYTES = A_ROWS * 128, B_BYTES = 128 * 128;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_all[];   // per group: [A0 | A1 | B0 | B1]
    typedef __attribute__((address_space(3))) void* lptr;
    typedef __attribute__((address_space(1))) const void* gptr;

    const int tid = threadIdx.x & 255, lane = tid & 63;       // thread / wave index inside the 4-wave group
    const int grp = KS == 1 ? 0 : __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8));
    unsigned char* smem = smem_all + grp * (2 * A_BYTES + 2 * B_BYTES);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int ntm = (a.M + 127) >> 7;
    const int tile_m = blockIdx.x % ntm, tile_n = blockIdx.x / ntm;
    const int m0 = tile_m * 128, n0 = tile_n * 128;
    const T* X = reinterpret_cast<const T*>(a.X);
    const T* W = reinterpret_cast<const T*>(a.W);
    T* Y = reinterpret_cast<T*>(a.Y);
    const T* R = reinterpret_cast<const T*>(a.R);

    if (a.lens) {
        int mlast = min(m0 + 127, a.M - 1);
        int b0 = m0 / a.S, b1 = mlast / a.S;
        if (b0 == b1 && (m0 - b0 * a.S) >= a.lens[b0]) {
            if (!a.accumulate) {
                for (int i = threadIdx.x; i < 128 * 128; i += 256 * KS) {
                    int r = i >> 7, c = i & 127;
                    int m = m0 + r, n = n0 + c;
                    if (m < a.M && n < a.N) Elem<T>::st(Y + (size_t)m * a.ldy + n, 0.f);
                }
            }
            return;
        }
    }
    const int nkc_all = (a.Cin + BK - 1) / BK;
    const int nkc = nkc_all / KS;                            // chunks of this group (host guarantees divisibility): kc_g = kc * KS + grp
    const int nsteps = a.taps * nkc;
    const T* zline = reinterpret_cast<const T*>(fs2_zero_line);
    // DMA lane geometry: a wave-load covers 8 rows x 128 B; lane -> (row lr, linear chunk lc), source chunk lc ^ f(row)
    const int lr = lane >> 3, lc = lane & 7;

    auto load_A = [&](int kc, int buf) {            // 20 wave-loads, 5 per wave
        unsigned char* dst = smem + buf * A_BYTES;
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            int wl = wave * 5 + j;
            if (wl * 8 >= 128 + (a.taps - 1) * a.dil) continue;   // wave-uniform: rows beyond the needed halo
            int r = wl * 8 + lr;                     // halo-tile row
            int g = m0 - a.pad + r;                  // global activation row
            int col = (kc * KS + grp) * BK + ((lc ^ ((r >> 1) & 7)) * EPC);
            const T* src = (g >= 0 && g < a.M && col < a.Cin) ? X + ((size_t)g * a.ldx + col) : zline;
            __builtin_amdgcn_global_load_lds((gptr)src, (lptr)(dst + wl * 1024), 16, 0, 0);
        }
    };
    auto load_B = [&](int step, int buf) {          // 16 wave-loads, 4 per wave
        int kc = step / a.taps, tap = step - kc * a.taps;
        unsigned char* dst = smem + 2 * A_BYTES + buf * B_BYTES;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int wl = wave * 4 + j;
            int r = wl * 8 + lr;
            int n = n0 + r;
            int col = (kc * KS + grp) * BK + ((lc ^ ((r >> 1) & 7)) * EPC);
            const T* src = (n < a.N && col < a.Cin) ? W + ((size_t)n * a.ldw + (size_t)tap * a.Cin + col) : zline;
            __builtin_amdgcn_global_load_lds((gptr)src, (lptr)(dst + wl * 1024), 16, 0, 0);
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int fl = lane & 31, fh = lane >> 5;
    // tap-validity bitmask of this lane's two A rows (bit j: tap j stays inside the row's sequence)
    unsigned vmask[2];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
        int m = m0 + wm * 64 + mb * 32 + fl;
        unsigned msk = 0;
        if (m < a.M) {
            int t = m % a.S;
            for (int j = 0; j < a.taps; ++j) {
                int ts = t + j * a.dil - a.pad;
                if (ts >= 0 && ts < a.S) msk |= 1u << j;
            }
        }
        vmask[mb] = msk;
    }
    const bool need_mask = a.taps > 1;

    load_A(0, 0);
    load_B(0, 0);
    __syncthreads();                                 // compiler drains vmcnt(0) before the barrier
    for (int step = 0; step < nsteps; ++step) {
        const int kc = step / a.taps, tap = step - kc * a.taps;
        if (step + 1 < nsteps) {
            if (tap == a.taps - 1) load_A(kc + 1, (kc + 1) & 1);
            load_B(step + 1, (step + 1) & 1);
        }
        const unsigned char* As = smem + (kc & 1) * A_BYTES + tap * a.dil * 128;
        const unsigned char* Bs = smem + 2 * A_BYTES + (step & 1) * B_BYTES;
        const int roff = tap * a.dil;                // physical halo row = logical row + roff (swizzle uses the physical row)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            uint4 af[2], bf[2];
            int c = fh * 4 + j;
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
                int r = wm * 64 + mb * 32 + fl;
                int pr = r + roff;
                af[mb] = *reinterpret_cast<const uint4*>(As + r * 128 + ((c ^ ((pr >> 1) & 7)) << 4));
                if (need_mask && !((vmask[mb] >> tap) & 1u)) af[mb] = make_uint4(0, 0, 0, 0);
            }
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                int r = wn * 64 + nb * 32 + fl;
                bf[nb] = *reinterpret_cast<const uint4*>(Bs + r * 128 + ((c ^ ((r >> 1) & 7)) << 4));
            }
            if constexpr (sizeof(T) == 4) {
#pragma unroll
                for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                        for (int nb = 0; nb < 2; ++nb)
                            acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(
                                reinterpret_cast<const float*>(&af[mb])[jj], reinterpret_cast<const float*>(&bf[nb])[jj],
                                acc[mb][nb], 0, 0, 0);
            } else {
#pragma unroll
                for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                    for (int nb = 0; nb < 2; ++nb)
                        acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                            __builtin_bit_cast(bf16x8, af[mb]), __builtin_bit_cast(bf16x8, bf[nb]), acc[mb][nb], 0, 0, 0);
            }
        }
        __syncthreads();
    }

    if (KS == 2) {                                         // group 1 hands its partial tile to group 0 through LDS
        __syncthreads();                                   // (every wave has left the K loop: operand buffers are dead)
        float* part = reinterpret_cast<float*>(smem_all);
        if (grp == 1) {
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) part[((mb * 2 + nb) * 16 + r) * 256 + tid] = acc[mb][nb][r];
        }
        __syncthreads();
        if (grp == 0) {
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[mb][nb][r] += part[((mb * 2 + nb) * 16 + r) * 256 + tid];
        }
        __syncthreads();
    }
    float* tile = reinterpret_cast<float*>(smem_all);
    auto stage = [&](auto actc) {
        constexpr int ACT = decltype(actc)::value;
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            int cl = wn * 64 + nb * 32 + fl;
            int n = n0 + cl;
            float bv = (a.bias && n < a.N) ? a.bias[n] : 0.f;
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    int rl = wm * 64 + mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                    tile[rl * 128 + fs2_tile_col128<T>(cl)] = act_ct<ACT>(acc[mb][nb][r] + bv, a.slope);
                }
        }
    };
    if (grp == 0) FS2_ACT_DISPATCH(a.act, stage(std::integral_constant<int, ACT>{}));
    __syncthreads();
    gemm_store_tile<T, 128, 256 * KS>(a, tile, m0, n0, threadIdx.x);
}


// =====================================================================================================
// Ring-buffered, wave-specialised variant for the large bf16 contractions.
// 256(M) x 128(N) tile, ONE 768-thread workgroup per CU = three waves per SIMD (r01j: was 512 threads = one consumer +
// one loader wave per SIMD with 128x64 consumer tiles; a lone consumer wave cannot overlap its own ds_read bursts with
// its MFMAs - ablation: ~1000 cycles of MFMA + ~800 cycles of fragment-read issue per K-step, serial - two 64x64 consumer
// waves per SIMD interleave them for free):
//   waves 0-7  CONSUMERS (4(M) x 2(N), each a 64x64 sub-tile = 2x2 MFMA 32x32x16 blocks, 64 accumulator registers):
//              ds_read_b128 fragment loads + MFMA only;
//   waves 8-11 LOADERS: LDS-DMA of the activation / weight tiles into rings of LDS slots, D-1 K-steps ahead.
// What the measurements said (w_1 k=9 forward, M=43200 N=1024 K=2304; PMC + s_memtime phase timers):
//   * 128^2 kernels: waves parked 43-63 % of their cycles in vmcnt(0)+barrier at the end of every K-step, ~14
//     non-MFMA instructions issued per MFMA, MFMA pipe 25-29 % busy;
//   * issuing one 1-KB LDS-DMA op costs the ISSUING wave 120-260 cycles under load (616 cycles per K-step for 5
//     ops) wherever it is placed — burst after the barrier or interleaved between MFMA groups — and an in-order
//     wave cannot issue MFMAs meanwhile.  So the DMA must come from a different wave than the MFMAs: the loader
//     wave's issue stalls overlap the consumer's matrix work on the same SIMD (guide §5: producer-consumer wave
//     specialisation as the way past the 128^2 structure's ceiling);
//   * with the builtin, hipcc drains vmcnt(0) before the first ds_read of every step (it cannot tell which slot a
//     pending DMA writes), so the DMA is issued from inline asm (guide §5.7) and tracked by hand: each loader waits
//     with a COUNTED s_waitcnt vmcnt(N) for its own ops of the current step, then the (raw) s_barrier publishes the
//     slot to the consumers; loads of the next D-2 steps stay in flight across the barrier.
//   * per-lane source offsets are computed once (32-bit VGPR offset, row clamped into the array); the K/tap advance is
//     a scalar base (SGPR pair): a DMA op is one s_mov m0 + one global_load_lds, no VALU.
//   * consumers double-buffer their fragments: the reads of MFMA group g+1 are issued before the MFMAs of group g,
//     and the first group of the next K-step is fetched right behind that step's barrier while the last group of the
//     current step is still executing.
//   conv (taps >= 3): activation halo tile double-buffered, fetched once per Cin-chunk a whole tap-loop early
//                     (A[kc+1] is issued at tap 0 of chunk kc); weight ring D = 4.      LDS = 2*34 + 1 + 4*16 = 133 KB
//   taps == 1       : A and B both in rings of D = 3.                                   LDS = 3*32 + 3*16     = 144 KB
// vmcnt bookkeeping: every loader issues the SAME number of ops per step (A: 9 per conv chunk — loaders that own fewer
// live 8-row groups pad with dummy loads into a scratch slot — or 8 per step when taps == 1; B: 4 per step), so the
// number of ops younger than the slot being waited for is a compile-time constant per case.
// Requires Cin % 64 == 0 and operand footprints < 2 GB (32-bit offsets); the host falls back to the 128^2 kernels
// otherwise.  Rows outside [0, M) / [0, N) are CLAMPED instead of zero-filled: they only ever feed outputs that are
// not stored, or taps that the boundary mask zeroes (tiles touching a sequence end always run the MASK variant).

template <bool ONE_TAP> struct RingCfg;
template <> struct RingCfg<false> { static constexpr int D = 4, A_BYTES = 272 * 128, NA = 2, SCRATCH = 2 * 272 * 128, B_OFF = 2 * 272 * 128 + 1024, NJA = 9; };
template <> struct RingCfg<true>  { static constexpr int D = 3, A_BYTES = 256 * 128, NA = 3, SCRATCH = 0, B_OFF = 3 * 256 * 128, NJA = 8; };
static constexpr int RING_B_BYTES = 128 * 128;

// ---- loader waves (lw = 0..3): all DMA of the tile
template <bool ONE_TAP>
__device__ __forceinline__ void ring_loader(const ConvGemmArgs& a, unsigned char* smem, int m0, int n0, int lane, int lw) {
    typedef RingCfg<ONE_TAP> C;
    constexpr int D = C::D, NJA = C::NJA;
    const unsigned char* Xb = reinterpret_cast<const unsigned char*>(a.X);
    const unsigned char* Wb = reinterpret_cast<const unsigned char*>(a.W);
    const int nkc = a.Cin >> 6;
    const int taps = ONE_TAP ? 1 : a.taps;
    const int nsteps = taps * nkc;
    const int lr = lane >> 3, lc = lane & 7;
    const int arows = 256 + (taps - 1) * a.dil;
    const unsigned smem_base = lds_addr(smem);

    unsigned offA[NJA], offB[4], ldsA[NJA];                // per-lane source offsets (bytes); wave-uniform LDS offsets
#pragma unroll
    for (int j = 0; j < NJA; ++j) {
        int wl = lw + 4 * j;
        const bool live = wl * 8 < arows;                 // wave-uniform
        int r = wl * 8 + lr;
        int g = min(max(m0 - a.pad + r, 0), a.M - 1);
        offA[j] = live ? (unsigned)g * (unsigned)(a.ldx * 2) + (unsigned)((lc ^ ((r >> 1) & 7)) << 4) : 0u;
        ldsA[j] = live ? (unsigned)(wl * 1024) : 0xffffffffu;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int r = (lw * 4 + j) * 8 + lr;
        int n = min(n0 + r, a.N - 1);
        offB[j] = (unsigned)n * (unsigned)(a.ldw * 2) + (unsigned)((lc ^ ((r >> 1) & 7)) << 4);
    }
    const bool nodma = FS2_DEV_DBG(a.dbg & 1);
    auto load_A = [&](int kc, int buf) {
        if (nodma) return;
        const unsigned char* base = Xb + (size_t)kc * 128;
#pragma unroll
        for (int j = 0; j < NJA; ++j) {
            unsigned d = (ldsA[j] == 0xffffffffu) ? smem_base + C::SCRATCH : smem_base + buf * C::A_BYTES + ldsA[j];
            glds16_sbase(offA[j], base, __builtin_amdgcn_readfirstlane(d));
        }
    };
    auto load_B = [&](int kc, int tap, int slot) {
        if (nodma) return;
        const unsigned char* base = Wb + ((size_t)tap * a.Cin + (size_t)kc * 64) * 2;
        const unsigned d0 = smem_base + C::B_OFF + slot * RING_B_BYTES + lw * 4096;
#pragma unroll
        for (int j = 0; j < 4; ++j) glds16_sbase(offB[j], base, d0 + j * 1024);
    };
    // issue pointer (ikc, itap, islot) = the step being filled, D-1 ahead of the step being consumed
    int ikc = 0, itap = 0, islot = 0;
    auto advance_issue = [&]() {
        if (++itap == taps) { itap = 0; ++ikc; }
        if (++islot == D) islot = 0;
    };
    if (!ONE_TAP) load_A(0, 0);
#pragma unroll
    for (int p = 0; p < D - 1; ++p) {
        if (p < nsteps) {
            if (ONE_TAP) load_A(ikc, islot);
            load_B(ikc, itap, islot);
        }
        advance_issue();
    }
    int step = 0;
    for (int kc = 0; kc < nkc; ++kc) {
        for (int tap = 0; tap < taps; ++tap, ++step) {
            // my ops of this step have landed: everything but the ops of the younger in-flight steps
            if (nsteps - 1 - step < D - 2) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            } else if (ONE_TAP) {
                asm volatile("s_waitcnt vmcnt(12)" ::: "memory");                  // (D-2) x (8 A + 4 B)
            } else {
                const bool a_young = (tap >= 1 && tap <= D - 2) && (kc + 1 < nkc);  // A[kc+1] was issued <= D-2 steps ago
                if (a_young) asm volatile("s_waitcnt vmcnt(17)" ::: "memory");     // (D-2) x 4 B + 9 A
                else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();                  // publishes slot(step); consumers are done with slot(step-1)
            if (!ONE_TAP && tap == 0 && kc + 1 < nkc) load_A(kc + 1, (kc + 1) & 1);
            if (step + D - 1 < nsteps) {
                if (ONE_TAP) load_A(ikc, islot);
                load_B(ikc, itap, islot);
            }
            advance_issue();
        }
    }
}

// ---- consumer waves (wave = 0..3 as 2 x 2): fragment reads + MFMA
// INACT: leaky-ReLU applied to the activation fragments after they land (HiFi-GAN's pre-activation convs: 7 VALU per
// dword - widen both halves by shift/mask, max(x, slope*x), hardware bf16 pack - which fits under the wave's MFMAs;
// the LDS-DMA path cannot transform data on the way in)
template <bool ONE_TAP, bool MASK, bool INACT>
__device__ __forceinline__ void ring_consumer(const ConvGemmArgs& a, unsigned char* smem, f32x16 (&acc)[2][2], int m0, int lane,
                                              int wm, int wn) {
    typedef RingCfg<ONE_TAP> C;
    constexpr int D = C::D;
    const int nkc = a.Cin >> 6;
    const int taps = ONE_TAP ? 1 : a.taps;
    const int nsteps = taps * nkc;
    const int fl = lane & 31, fh = lane >> 5;
    unsigned boff[4];                                      // B fragment byte offsets inside a slot (nb = 0; nb = 1 is +4096)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int r = wn * 64 + fl;
        boff[j] = r * 128 + (((fh * 4 + j) ^ ((r >> 1) & 7)) << 4);
    }
    unsigned vmask[2] = {0, 0};
    if (MASK) {
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
            int m = m0 + wm * 64 + mb * 32 + fl;
            unsigned msk = 0;
            if (m < a.M) {
                int t = m % a.S;
                for (int j = 0; j < taps; ++j) {
                    int ts = t + j * a.dil - a.pad;
                    if (ts >= 0 && ts < a.S) msk |= 1u << j;
                }
            }
            vmask[mb] = msk;
        }
    }
    // Fragments are fetched in HALF-STEP groups (two 16-wide k-slices = 8 ds_read_b128, 8 MFMAs per wave): the reads of the
    // next group are issued before the MFMAs of the current one.
    // The reads are issued from inline asm and waited for with a hand-counted s_waitcnt, fenced by sched_barrier(0):
    //   * left to itself the machine scheduler sinks every ds_read down to just above the MFMA that consumes it and the
    //     software prefetch is gone (r01i ISA: s_waitcnt lgkmcnt(2)/(1)/(0) right behind freshly issued reads inside the
    //     MFMA runs; ablations: 86 us fixed + 95 us MFMA + 120 us reads ADD UP to the 304 us kernel, nothing overlaps);
    //   * with the order pinned by sched_barrier alone, the waitcnt-insertion pass still emits lgkmcnt(0) in front of
    //     every MFMA group (the older group's reads were issued across the loop back edge / the barrier branch), which
    //     waits for the 12 reads that were JUST issued - the same serialisation.
    // lgkmcnt counts LDS operations in order: with 8 younger reads allowed in flight, lgkmcnt(8) == "the older group
    // has landed".  Nothing else in this loop touches lgkmcnt (no SMEM, no LDS stores in consumer waves).
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define FS2_DS_READ_B128(dst, addr, OFF) asm volatile("ds_read_b128 %0, %1 offset:" #OFF : "=v"(dst) : "v"(addr))
    u32x4 af[2][2][2], bf[2][2][2];
    const unsigned smem_u = lds_addr(smem);
    auto read_group = [&](int set, int kc, int tap, int slot, int g) {
        const int roff = tap * a.dil;
        const unsigned a_base = (ONE_TAP ? slot : (kc & 1)) * C::A_BYTES + (wm * 64 + roff) * 128;
        const unsigned x = ((fl + roff) >> 1) & 7;         // swizzle key of the physical halo row
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int j = 2 * g + q;
            const unsigned aaddr = smem_u + a_base + fl * 128 + ((((unsigned)(fh * 4 + j)) ^ x) << 4);
            const unsigned baddr = smem_u + C::B_OFF + slot * RING_B_BYTES + boff[j];
            FS2_DS_READ_B128(bf[set][q][0], baddr, 0);
            FS2_DS_READ_B128(bf[set][q][1], baddr, 4096);
            FS2_DS_READ_B128(af[set][q][0], aaddr, 0);
            FS2_DS_READ_B128(af[set][q][1], aaddr, 4096);
        }
    };
    auto mfma_group = [&](int set, int tap) {
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
                u32x4 av = af[set][q][mb];
                if (INACT) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float lo = __uint_as_float(av[e] << 16), hi = __uint_as_float(av[e] & 0xffff0000u);
                        av[e] = pack_bf16x2(fmaxf(lo, lo * a.in_slope), fmaxf(hi, hi * a.in_slope));       // 0 < slope < 1
                    }
                }
                if (MASK && !((vmask[mb] >> tap) & 1u)) av = u32x4{0u, 0u, 0u, 0u};     // tap leaves the row's sequence
#pragma unroll
                for (int nb = 0; nb < 2; ++nb)
                    acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                        __builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, bf[set][q][nb]), acc[mb][nb], 0, 0, 0);
            }
    };
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // kernel-argument loads etc. retired: lgkmcnt is ours from here
    __builtin_amdgcn_s_barrier();                          // slot(0) published
    read_group(0, 0, 0, 0, 0);
    int slot = 0, step = 0;
    for (int kc = 0; kc < nkc; ++kc) {
        for (int tap = 0; tap < taps; ++tap, ++step) {
            read_group(1, kc, tap, slot, 1);
            asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");       // first half (8 older reads) has landed
            __builtin_amdgcn_sched_barrier(0);
            mfma_group(0, tap);
            __builtin_amdgcn_sched_barrier(0);
            // next step's first half is fetched behind ITS barrier while this step's second half runs
            int nslot = slot + 1; if (nslot == D) nslot = 0;
            int ntap = tap + 1, nkc_ = kc; if (ntap == taps) { ntap = 0; ++nkc_; }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // second half landed; my reads of slot(step) are done
            if (step + 1 < nsteps) {
                __builtin_amdgcn_s_barrier();
                read_group(0, nkc_, ntap, nslot, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            mfma_group(1, tap);
            __builtin_amdgcn_sched_barrier(0);
            slot = nslot;
        }
    }
#undef FS2_DS_READ_B128
}

template <bool ONE_TAP, bool INACT>
__global__ void __launch_bounds__(768, 3) conv_gemm_ring_kernel(ConvGemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = (wave >> 1) & 3, wn = wave & 1;          // consumer wave grid 4 (M) x 2 (N), 64 x 64 each
    const int ntm = (a.M + 255) >> 8;
    const int tile_m = blockIdx.x % ntm, tile_n = blockIdx.x / ntm;
    const int m0 = tile_m * 256, n0 = tile_n * 128;
    bf16_t* Y = reinterpret_cast<bf16_t*>(a.Y);

    if (a.lens) {
        int mlast = min(m0 + 255, a.M - 1);
        int b0 = m0 / a.S, b1 = mlast / a.S;
        if (b0 == b1 && (m0 - b0 * a.S) >= a.lens[b0]) {
            if (!a.accumulate) {
                for (int i = tid; i < 256 * 16; i += 768) {              // 16-byte zero stores where the row allows it
                    int r = i >> 4, c = (i & 15) * 8;
                    int m = m0 + r, n = n0 + c;
                    if (m >= a.M || n >= a.N) continue;
                    bf16_t* yp = Y + (size_t)m * a.ldy + n;
                    if (a.vec_ok && n + 8 <= a.N) *reinterpret_cast<uint4*>(yp) = make_uint4(0, 0, 0, 0);
                    else for (int e = 0; e < 8 && n + e < a.N; ++e) yp[e] = 0;
                }
            }
            return;
        }
    }
    if (FS2_DEV_DBG(a.dbg & 16)) return;
    f32x16 acc[2][2];
    if (wave >= 8) {
        ring_loader<ONE_TAP>(a, smem, m0, n0, lane, wave - 8);
    } else {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        if (ONE_TAP) {
            ring_consumer<true, false, INACT>(a, smem, acc, m0, lane, wm, wn);
        } else {
            int t0 = m0 % a.S;
            int rows = min(256, a.M - m0);
            const bool need_mask = (t0 < a.pad) || (t0 + rows - 1 + (a.taps - 1) * a.dil - a.pad >= a.S);
            if (need_mask) ring_consumer<false, true, INACT>(a, smem, acc, m0, lane, wm, wn);
            else ring_consumer<false, false, INACT>(a, smem, acc, m0, lane, wm, wn);
        }
    }
    __syncthreads();                                       // every consumer is done reading the operand slots
    if (FS2_DEV_DBG(a.dbg & 8)) { if (tid == 0 && acc[0][0][0] == 12345.f) Y[0] = 0; return; }

    // epilogue: the whole 256x128 f32 tile through LDS (128 KB), written by the consumers, stored by all 8 waves
    float* tile = reinterpret_cast<float*>(smem);
    if (wave < 8) {
        const int fl = lane & 31, fh = lane >> 5;
        auto stage = [&](auto actc) {
            constexpr int ACT = decltype(actc)::value;
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                int cl = wn * 64 + nb * 32 + fl;
                int n = n0 + cl;
                float bv = (a.bias && n < a.N) ? a.bias[n] : 0.f;
#pragma unroll
                for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        int rl = wm * 64 + mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                        tile[rl * 128 + fs2_tile_col128<bf16_t>(cl)] = act_ct<ACT>(acc[mb][nb][r] + bv, a.slope);
                    }
            }
        };
        FS2_ACT_DISPATCH(a.act, stage(std::integral_constant<int, ACT>{}));
    }
    __syncthreads();
    if (FS2_DEV_DBG(a.dbg & 32)) { if (tile[tid] == 12345.f) Y[0] = 0; return; }
    if (tid < 512) {                                       // (the shared store path walks 128 x 16 chunks in whole strides of its thread count)
        gemm_store_tile<bf16_t, 128, 512>(a, tile, m0, n0, tid);
        gemm_store_tile<bf16_t, 128, 512>(a, tile + 128 * 128, m0 + 128, n0, tid);
    }
}

// =====================================================================================================
// "Skinny" convolution: Cin == Cout == C in {32, 64, 128}, bf16 - the HiFi-GAN residual blocks after the first
// upsampling stage (hifigan/models.py:96-103): M = B * T * up is huge (1.8 M rows per 8 utterances at the last stage),
// C is tiny.  These layers are HBM-bound by construction (C=32, k=3: 48 FLOP per byte; every conv moves M*C*2 bytes in
// and out = 236 MB at every stage), and the general kernels waste them: a 128x128 tile computes 4x / 2x more columns
// than exist, the K-chunk of 64 is half padding at C=32, and the A tile is re-staged per tap (r01j trace: 470 us per
// C=32 conv = 0.5 TB/s, 39 % of the whole synthesis step).
// Here one workgroup owns 128 rows x all C columns: the activation tile is staged ONCE with its full dilation halo
// (leaky-ReLU applied while staging), the weights of a group of taps sit in LDS next to it, each wave multiplies
// 32 rows x C columns tap by tap from those two tiles, and the shared epilogue (bias, activation, residual,
// accumulate, 1/3 scale) writes whole rows.  X is read once, Y written once: ~24 KB of HBM traffic per 128 rows.
// LDS rows are C*2 bytes; 16-byte chunk c of row r is stored at chunk c ^ ((r / RPW) % CPR) (CPR = chunks per row,
// RPW = rows per 256-byte bank window), which makes the ds_read_b128 of one chunk by 16 consecutive rows conflict-free
// for all three widths.
// (r01j: fusing a residual block's conv1 -> leaky-ReLU -> conv2 + residual into one kernel, intermediate in LDS, was tried
// for C = 32 / 64: 2.5x less HBM traffic on paper, but two weight-group loops with a barrier pair per group (24 barriers at
// k = 11) and 1-2 workgroups per CU made the synthesis step 24 % SLOWER than two skinny launches.  Not kept.)
template <int C> struct SkinnyCfg {
    static constexpr int CPR = C / 8, RPW = 16 / CPR, XR_MAX = 192;
    static constexpr int G = C == 32 ? 8 : (C == 64 ? 2 : 1);          // taps per weight group (16 / 16 / 32 KB)
    static constexpr int X_BYTES = XR_MAX * C * 2, W_BYTES = G * C * C * 2;
    static constexpr int EPI_BYTES = 128 * C * 4;
    static constexpr int LDS = (X_BYTES + W_BYTES) > EPI_BYTES ? (X_BYTES + W_BYTES) : EPI_BYTES;
};
template <int C> __device__ __forceinline__ unsigned skinny_swz(int row, int c) {
    typedef SkinnyCfg<C> K;
    return (unsigned)(row * (C * 2) + ((c ^ ((row / K::RPW) % K::CPR)) << 4));
}

template <int C>
__global__ void __launch_bounds__(256) conv_skinny_kernel(ConvGemmArgs a) {
    typedef SkinnyCfg<C> K;
    constexpr int CPR = K::CPR, NB = C / 32, NS = C / 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sX = smem;
    unsigned char* sW = smem + K::X_BYTES;
    const int tid = threadIdx.x, lane = tid & 63, fl = lane & 31, fh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m0 = blockIdx.x * 128;
    const bf16_t* X = reinterpret_cast<const bf16_t*>(a.X);
    const bf16_t* W = reinterpret_cast<const bf16_t*>(a.W);
    const int halo = (a.taps - 1) * a.dil;
    const int xr = 128 + halo;                              // staged rows: global rows m0 - pad .. m0 - pad + xr - 1
    // ---- stage X (all loads first, then activation + swizzled stores)
    constexpr int MAXI = (K::XR_MAX * CPR + 255) / 256;
    uint4 xv[MAXI];
#pragma unroll
    for (int i = 0; i < MAXI; ++i) {
        const int idx = tid + 256 * i;
        const int row = idx / CPR, c = idx % CPR;
        const int g = m0 - a.pad + row;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (row < xr && g >= 0 && g < a.M) v = *reinterpret_cast<const uint4*>(X + (size_t)g * a.ldx + c * 8);
        xv[i] = v;
    }
#pragma unroll
    for (int i = 0; i < MAXI; ++i) {
        const int idx = tid + 256 * i;
        const int row = idx / CPR, c = idx % CPR;
        if (row < xr) {
            uint4 v = xv[i];
            if (a.in_act == FS2_ACT_LRELU) v = act_chunk_bf16(v, a.in_slope);
            *reinterpret_cast<uint4*>(sX + skinny_swz<C>(row, c)) = v;
        }
    }
    // tap-validity bitmask of this lane's output row (bit j: tap j stays inside the row's own sequence)
    unsigned vmask = 0;
    {
        const int m = m0 + wave * 32 + fl;
        if (m < a.M) {
            const int t = m % a.S;
            for (int j = 0; j < a.taps; ++j) {
                const int ts = t + j * a.dil - a.pad;
                if (ts >= 0 && ts < a.S) vmask |= 1u << j;
            }
        }
    }
    f32x16 acc[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;

    constexpr int WI = (K::G * C * CPR) / 256;              // 16-byte chunks of a weight group per thread (4 / 4 / 8)
    for (int g0 = 0; g0 < a.taps; g0 += K::G) {
        __syncthreads();                                   // X tile visible / previous weight group consumed
#pragma unroll
        for (int i = 0; i < WI; ++i) {
            const int idx = tid + 256 * i;
            const int tl = idx / (C * CPR), rem = idx % (C * CPR);
            const int n = rem / CPR, c = rem % CPR;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (g0 + tl < a.taps) v = *reinterpret_cast<const uint4*>(W + ((size_t)n * a.taps + (g0 + tl)) * C + c * 8);
            *reinterpret_cast<uint4*>(sW + tl * (C * C * 2) + skinny_swz<C>(n, c)) = v;
        }
        __syncthreads();
#pragma unroll
        for (int tl = 0; tl < K::G; ++tl) {
            const int tap = g0 + tl;
            if (tap >= a.taps) break;                      // block-uniform
            const int r = wave * 32 + fl + tap * a.dil;    // my row inside the halo tile
            const bool live = (vmask >> tap) & 1u;
            const unsigned char* wt = sW + tl * (C * C * 2);
#pragma unroll
            for (int sl = 0; sl < NS; ++sl) {
                const int c = 2 * sl + fh;
                uint4 av = *reinterpret_cast<const uint4*>(sX + skinny_swz<C>(r, c));
                if (!live) av = make_uint4(0, 0, 0, 0);
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    uint4 bv = *reinterpret_cast<const uint4*>(wt + skinny_swz<C>(nb * 32 + fl, c));
                    acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, bv), acc[nb], 0, 0, 0);
                }
            }
        }
    }
    __syncthreads();                                       // operand tiles dead: LDS becomes the f32 staging tile
    float* tile = reinterpret_cast<float*>(smem);
    auto stage = [&](auto actc) {
        constexpr int ACT = decltype(actc)::value;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const int cl = nb * 32 + fl;
            const float bv = a.bias ? a.bias[cl] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rl = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                tile[rl * C + (C == 128 ? fs2_tile_col128<bf16_t>(cl) : cl)] = act_ct<ACT>(acc[nb][r] + bv, a.slope);
            }
        }
    };
    FS2_ACT_DISPATCH(a.act, stage(std::integral_constant<int, ACT>{}));
    __syncthreads();
    gemm_store_tile<bf16_t, C, 256, 128>(a, tile, m0, 0, tid);
}

template <int C>
static void launch_skinny(const ConvGemmArgs& a, hipStream_t stream) {
    static Fs2DevOnce once;
    once.run([&] { (void)hipFuncSetAttribute((const void*)conv_skinny_kernel<C>, hipFuncAttributeMaxDynamicSharedMemorySize, SkinnyCfg<C>::LDS); });
    conv_skinny_kernel<C><<<(unsigned)fs2_cdiv(a.M, 128), 256, SkinnyCfg<C>::LDS, stream>>>(a);
}

// defined in fs2_gemm_w.hip
bool fs2_conv_gemm_w_ok(const ConvGemmArgs& a, bool has_map, int dtype);
void fs2_conv_gemm_w_launch(const ConvGemmArgs& a, const int32_t* tile_map, hipStream_t stream);
// defined in fs2_gemm_p.hip
bool fs2_conv_gemm_p_ok(const ConvGemmArgs& a, bool has_map, int dtype, int ks);
void fs2_conv_gemm_p_launch(const ConvGemmArgs& a, const int32_t* tile_map, hipStream_t stream, int abl, int ks, float* ws,
                            float* tail_ws);

// Which kernel a launch description dispatches to: a PURE function of the description (no state, no environment in the
// shipped build), shared by fs2_conv_gemm and by the query entry point fs2_conv_gemm_variant that bench.py uses to attribute
// its HIP-event durations to the kernel names rocprofv3 reports.
// defined in fs2_gemm_s.hip
bool fs2_conv_gemm_s_ok(const ConvGemmArgs& a, int dtype);
void fs2_conv_gemm_s_launch(const ConvGemmArgs& a, hipStream_t stream);
struct GemmPick { int variant; bool ring_inact, ks2; };
static GemmPick conv_gemm_pick(const ConvGemmArgs& a, int dtype, bool has_map, bool has_tail_ws = true) {
    const int M = a.M, N = a.N, Cin = a.Cin, taps = a.taps, dil = a.dil, in_act = a.in_act;
    GemmPick p;
    const long grid = (long)fs2_cdiv(M, 128) * fs2_cdiv(N, 128);
    // taps == 1 keeps the register-staged kernel (its 3-stage pipeline wins when there is no halo to reuse)
    static const int dma1 = fs2_dev_env("FS2_GEMM_DMA1", 1);            // dev A/B: one-tap launches of few tiles and K >= 768 on the DMA kernel's in-workgroup K split
    const bool dma = in_act == FS2_ACT_NONE && (taps > 1 || (dma1 && grid <= 160 && Cin % 128 == 0 && Cin >= 768 && dtype == FS2_BF16)) && (taps - 1) * dil <= 16 && taps <= 32;
    // ring-buffered 256x128 tiles once there are enough of them to fill the chip (FS2_GEMM_TILE=128|256: dev A/B only)
    static const int force_tile = fs2_dev_env("FS2_GEMM_TILE", 0);
    const long big_tiles = (long)fs2_cdiv(M, 256) * fs2_cdiv(N, 128);
    p.ring_inact = in_act == FS2_ACT_LRELU && a.in_slope > 0.f && a.in_slope < 1.f;
    const bool ring_ok = dtype == FS2_BF16 && (in_act == FS2_ACT_NONE || p.ring_inact) && (taps == 1 || (taps >= 3 && (taps - 1) * dil <= 16 && taps <= 32)) &&
                         Cin % 64 == 0 && (double)M * a.ldx * 2 < 2.0e9 && (double)N * taps * Cin * 2 < 2.0e9;
    // taps == 1 contractions with a short K (4-12 steps) do not amortise the ring's fill: measured faster on the 128^2 kernel
    bool big = ring_ok && big_tiles >= 170 && (taps > 1 || Cin >= 1024);
    if (force_tile == 128) big = false;
    if (force_tile == 256 && ring_ok) big = true;
    // C = 128 stays on the 256x128 / 128^2 kernels: it is MFMA-bound there (k=11: 172 GFLOP per conv) and one tap per
    // weight group leaves only 32 MFMAs per wave between barriers (r01j A/B: 13.1 ms with, 12.0 ms without bit 4|8)
    static const int skinny_mask = fs2_dev_env("FS2_GEMM_SKINNY", 3);    // dev A/B bits
    const int skinny_bit = Cin == 32 ? 1 : (Cin == 64 ? 2 : (in_act == FS2_ACT_NONE ? 4 : 8));
    const bool skinny = (skinny_mask & skinny_bit) && dtype == FS2_BF16 && N == Cin && (Cin == 32 || Cin == 64 || Cin == 128) && taps <= 16 &&
                        (taps - 1) * dil <= 64 && (in_act == FS2_ACT_NONE || in_act == FS2_ACT_LRELU) && a.vec_ok;
    // persistent 256x128 kernel (fs2_gemm_p.hip): every shape the ring kernel takes plus the short-K one-tap contractions
    // (its run-ahead loaders hide the per-tile ring fill those could not amortise).  FS2_GEMM_P=0: dev A/B against the ring.
    static const int p_on = fs2_dev_env("FS2_GEMM_P", 1);
    // HiFi-GAN's stored-leaky-ReLU chain launches WITH a residual / accumulate operand and a short reduction (conv2 of a block: 6 - 44
    // K-steps): the ring kernel's LDS-staged epilogue reads the operand four chunks at a time, the persistent kernel's register
    // epilogue one chunk at a time behind its own stores (a vmcnt(0) each) - same box, `tools/bench_voc.py`: 116.7 -> 97.2 us (C = 128,
    // k = 3), 146.3 -> 132.7 (k = 7), 89.8 -> 85.3 (C = 256, k = 11); without such an operand the persistent kernel is the faster one
    // at every shape (profiles/r05t_bench_voc_ring.log).  Scoped to the launches it was measured on.
    const bool res_short = (a.res_unlrelu > 0.f || a.post_slope > 0.f) && (a.R || a.accumulate) && a.act != FS2_ACT_GATE &&
                           (long)taps * ((Cin + 63) / 64) <= 48 && big;
    const bool persist = p_on && !skinny && !res_short && fs2_conv_gemm_p_ok(a, has_map, dtype, 1) && (taps > 1 || Cin >= 256);
    // few tiles, long reduction: split the Cin chunks over two wave groups of one workgroup (FS2_GEMM_KSPLIT=0: off)
    static const int ksplit_on = fs2_dev_env("FS2_GEMM_KSPLIT", 1);
    static const int ks2_min = fs2_dev_env("FS2_GEMM_KS2_MIN", 12);
    p.ks2 = ksplit_on && dtype == FS2_BF16 && grid <= 160 && Cin % 128 == 0 && (long)taps * (Cin / 64) >= ks2_min;
    // wide-tile one-tap kernel (fs2_gemm_w.hip): N a multiple of 256 - the Linear layers of the FFT blocks and their data gradients
    static const int w_on = fs2_dev_env("FS2_GEMM_W", 1);
    // streaming kernel with the weights in registers (fs2_gemm_s.hip): the K = 256 Linear layers and data gradients
    const bool lrelu_io = a.res_unlrelu > 0.f || a.post_slope > 0.f;     // (the streaming kernel's own epilogue does not carry them)
    if (!skinny && !lrelu_io && fs2_conv_gemm_s_ok(a, dtype)) { p.variant = FS2_GEMM_STREAM_K256; return p; }
    if (w_on && !skinny && fs2_conv_gemm_w_ok(a, has_map, dtype)) { p.variant = FS2_GEMM_WIDE_1TAP; return p; }
    p.variant = skinny ? FS2_GEMM_SKINNY : (persist ? (taps == 1 ? FS2_GEMM_PERSIST_1TAP : FS2_GEMM_PERSIST) : (big ? FS2_GEMM_RING : (dma ? FS2_GEMM_DMA : FS2_GEMM_PLAIN)));
    return p;
}

static void conv_gemm_fill(ConvGemmArgs& a, const void* X, long ldx, const void* W, const float* bias, const void* R, long ldr, void* Y,
                           long ldy, const int32_t* lens, int M, int N, int Cin, int S, int taps, int dil, int pad, int act, float slope,
                           int in_act, float in_slope, int accumulate, float out_scale, int dtype) {
    const int epc = dtype == FS2_F32 ? 4 : 8;
    a.X = X; a.ldx = ldx; a.W = W; a.ldw = (long)taps * Cin; a.bias = bias; a.R = R; a.ldr = ldr; a.Y = Y; a.ldy = ldy;
    a.lens = lens; a.M = M; a.N = N; a.Cin = Cin; a.S = S; a.taps = taps; a.dil = dil; a.pad = pad; a.act = act;
    a.slope = slope; a.in_act = in_act; a.in_slope = in_slope; a.accumulate = accumulate; a.out_scale = out_scale;
    a.dbg = 0;
    a.res_unlrelu = 0.f; a.post_slope = 0.f;
    a.vec_ok = (ldy % epc == 0) && (((uintptr_t)Y & 15) == 0) && (!R || ((ldr % epc == 0) && (((uintptr_t)R & 15) == 0)));
}

// stateless query: the FS2_GEMM_* code fs2_conv_gemm would dispatch this description to (aligned 16-byte-addressable
// operands assumed, as every tensor the engine passes is).  has_lens / has_map: whether lens and tile_map are non-null.
extern "C" int fs2_conv_gemm_variant(long ldx, long ldy, long ldr, int has_lens, int has_map, int M, int N, int Cin, int S, int taps,
                                     int dil, int in_act, float in_slope, int dtype) {
    ConvGemmArgs a;
    conv_gemm_fill(a, nullptr, ldx, nullptr, nullptr, ldr ? (const void*)16 : nullptr, ldr, nullptr, ldy, has_lens ? (const int32_t*)16 : nullptr,
                   M, N, Cin, S > 0 ? S : 1, taps, dil, 0, 0, 0.f, in_act, in_slope, 0, 1.f, dtype);
    return conv_gemm_pick(a, dtype, has_map != 0).variant;
}

// the same query for a fs2_conv_gemm_lrelu_io launch (stored-leaky-ReLU chains: their dispatch also depends on the epilogue operands)
extern "C" int fs2_conv_gemm_lrelu_io_variant(long ldx, long ldy, long ldr, int accumulate, int M, int N, int Cin, int S, int taps, int dil,
                                              int act, float res_unlrelu, float post_slope, int dtype) {
    ConvGemmArgs a;
    conv_gemm_fill(a, nullptr, ldx, nullptr, nullptr, ldr ? (const void*)16 : nullptr, ldr, nullptr, ldy, nullptr, M, N, Cin, S > 0 ? S : 1,
                   taps, dil, 0, act, 0.f, FS2_ACT_NONE, 0.f, accumulate, 1.f, dtype);
    a.res_unlrelu = res_unlrelu; a.post_slope = post_slope;
    return conv_gemm_pick(a, dtype, false).variant;
}

static int conv_gemm_impl(const void* X, long ldx, const void* W, const float* bias, const void* R, long ldr, void* Y,
                          long ldy, const int32_t* lens, const int32_t* tile_map, float* tail_ws, int M, int N, int Cin, int S, int taps,
                          int dil, int pad, int act, float slope, int in_act, float in_slope, int accumulate, float out_scale,
                          int dtype, hipStream_t stream, float res_unlrelu = 0.f, float post_slope = 0.f) {
    FS2_CHECK_ARG(X && W && Y, "conv_gemm: null pointer");
    FS2_CHECK_ARG(M >= 0 && N > 0 && Cin > 0 && S > 0 && taps > 0 && dil > 0, "conv_gemm: bad shape M=%d N=%d Cin=%d S=%d taps=%d", M, N, Cin, S, taps);
    FS2_CHECK_ARG(M % S == 0, "conv_gemm: M=%d not a multiple of S=%d", M, S);
    int epc = dtype == FS2_F32 ? 4 : 8;
    FS2_CHECK_ARG(dtype == FS2_F32 || dtype == FS2_BF16, "conv_gemm: dtype %d", dtype);
    FS2_CHECK_ARG(Cin % epc == 0 && ldx % epc == 0, "conv_gemm: Cin=%d / ldx=%ld must be multiples of %d", Cin, ldx, epc);
    FS2_CHECK_ARG(((uintptr_t)X & 15) == 0 && ((uintptr_t)W & 15) == 0, "conv_gemm: X/W must be 16-byte aligned");
    if (M == 0) return FS2_OK;
    ConvGemmArgs a;
    conv_gemm_fill(a, X, ldx, W, bias, R, ldr, Y, ldy, lens, M, N, Cin, S, taps, dil, pad, act, slope, in_act, in_slope, accumulate, out_scale, dtype);
    static const int gemm_dbg = fs2_dev_env("FS2_GEMM_DBG", 0);
    a.dbg = gemm_dbg;
    FS2_CHECK_ARG(res_unlrelu >= 0.f && post_slope >= 0.f && post_slope < 1.f && (res_unlrelu == 0.f || (R && act != FS2_ACT_GATE)),
                  "conv_gemm: res_unlrelu needs an additive residual operand, post_slope in [0, 1)");
    a.res_unlrelu = res_unlrelu; a.post_slope = post_slope;
    const long grid = (long)fs2_cdiv(M, 128) * fs2_cdiv(N, 128);
    const long big_tiles = (long)fs2_cdiv(M, 256) * fs2_cdiv(N, 128);
    const GemmPick pk = conv_gemm_pick(a, dtype, tile_map != nullptr, tail_ws != nullptr);
    if (pk.variant == FS2_GEMM_STREAM_K256) {
        fs2_conv_gemm_s_launch(a, stream);
    } else if (pk.variant == FS2_GEMM_SKINNY) {
        if (Cin == 32) launch_skinny<32>(a, stream);
        else if (Cin == 64) launch_skinny<64>(a, stream);
        else launch_skinny<128>(a, stream);
    } else if (pk.variant == FS2_GEMM_WIDE_1TAP) {
        fs2_conv_gemm_w_launch(a, tile_map, stream);
    } else if (pk.variant == FS2_GEMM_PERSIST || pk.variant == FS2_GEMM_PERSIST_1TAP) {
        static const int abl = fs2_dev_env("FS2_GEMM_ABL", 0);
        fs2_conv_gemm_p_launch(a, tile_map, stream, abl, 1, nullptr, tail_ws);
    } else if (pk.variant == FS2_GEMM_RING) {
        static Fs2DevOnce ring_once;
        const int dyn1 = RingCfg<true>::B_OFF + RingCfg<true>::D * RING_B_BYTES;
        const int dynk = RingCfg<false>::B_OFF + RingCfg<false>::D * RING_B_BYTES;
        ring_once.run([&] {
            (void)hipFuncSetAttribute((const void*)conv_gemm_ring_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, dyn1);
            (void)hipFuncSetAttribute((const void*)conv_gemm_ring_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, dynk);
            (void)hipFuncSetAttribute((const void*)conv_gemm_ring_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, dyn1);
            (void)hipFuncSetAttribute((const void*)conv_gemm_ring_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, dynk);
        });
        if (pk.ring_inact) {
            if (taps == 1) conv_gemm_ring_kernel<true, true><<<(unsigned)big_tiles, 768, dyn1, stream>>>(a);
            else conv_gemm_ring_kernel<false, true><<<(unsigned)big_tiles, 768, dynk, stream>>>(a);
        } else if (taps == 1) conv_gemm_ring_kernel<true, false><<<(unsigned)big_tiles, 768, dyn1, stream>>>(a);
        else conv_gemm_ring_kernel<false, false><<<(unsigned)big_tiles, 768, dynk, stream>>>(a);
    } else if (pk.variant == FS2_GEMM_DMA) {
        const int dyn = 2 * 160 * 128 + 2 * 128 * 128;
        static Fs2DevOnce dma_once;
        dma_once.run([&] {
            (void)hipFuncSetAttribute((const void*)conv_gemm_dma_kernel<float, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, dyn);
            (void)hipFuncSetAttribute((const void*)conv_gemm_dma_kernel<bf16_t, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, dyn);
            (void)hipFuncSetAttribute((const void*)conv_gemm_dma_kernel<bf16_t, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * dyn);
        });
        if (dtype == FS2_F32) conv_gemm_dma_kernel<float, 1><<<(unsigned)grid, 256, dyn, stream>>>(a);
        else if (pk.ks2) conv_gemm_dma_kernel<bf16_t, 2><<<(unsigned)grid, 512, 2 * dyn, stream>>>(a);
        else conv_gemm_dma_kernel<bf16_t, 1><<<(unsigned)grid, 256, dyn, stream>>>(a);
    } else if (dtype == FS2_F32) conv_gemm_kernel<float><<<(unsigned)grid, 256, 0, stream>>>(a);
    else conv_gemm_kernel<bf16_t><<<(unsigned)grid, 256, 0, stream>>>(a);
    FS2_CHECK_LAUNCH("conv_gemm");
    return FS2_OK;
}

extern "C" int fs2_conv_gemm(const void* X, long ldx, const void* W, const float* bias, const void* R, long ldr, void* Y,
                             long ldy, const int32_t* lens, const int32_t* tile_map, int M, int N, int Cin, int S, int taps, int dil,
                             int pad, int act, float slope, int in_act, float in_slope, int accumulate, float out_scale, int dtype,
                             hipStream_t stream) {
    return conv_gemm_impl(X, ldx, W, bias, R, ldr, Y, ldy, lens, tile_map, nullptr, M, N, Cin, S, taps, dil, pad, act, slope, in_act,
                          in_slope, accumulate, out_scale, dtype, stream);
}

// fs2_conv_gemm with a scratch workspace for the persistent kernel's tail split (fs2_gemm_p.hip, PSched::tws): when the output
// tiles do not fill a whole number of rounds over the CUs, the last partial round is K-split so that it costs a fraction of a
// round.  tail_ws: fs2_conv_gemm_tail_ws_bytes() bytes of f32 scratch (any contents; must not be shared by launches that can
// run concurrently).  Launches the persistent kernel does not take ignore it.  Same results as fs2_conv_gemm up to the fp32
// summation order of the split tiles.
extern "C" int fs2_conv_gemm_tail(const void* X, long ldx, const void* W, const float* bias, const void* R, long ldr, void* Y,
                                  long ldy, const int32_t* lens, const int32_t* tile_map, float* tail_ws, int M, int N, int Cin,
                                  int S, int taps, int dil, int pad, int act, float slope, int in_act, float in_slope,
                                  int accumulate, float out_scale, int dtype, hipStream_t stream) {
    FS2_CHECK_ARG(!tail_ws || ((uintptr_t)tail_ws & 15) == 0, "conv_gemm_tail: tail_ws must be 16-byte aligned");
    return conv_gemm_impl(X, ldx, W, bias, R, ldr, Y, ldy, lens, tile_map, tail_ws, M, N, Cin, S, taps, dil, pad, act, slope, in_act,
                          in_slope, accumulate, out_scale, dtype, stream);
}

// fs2_conv_gemm_tail for chains that STORE leaky-ReLU'd activations (hifigan/models.py:96-103 as fastspeech2_amd/hifigan.py runs it):
// res_unlrelu = 1 / slope when R holds lrelu(r) (the raw r is added), post_slope = slope when lrelu(value) is to be stored.
extern "C" int fs2_conv_gemm_lrelu_io(const void* X, long ldx, const void* W, const float* bias, const void* R, long ldr, void* Y,
                                      long ldy, float* tail_ws, int M, int N, int Cin, int S, int taps, int dil, int pad, int act,
                                      float slope, int accumulate, float out_scale, float res_unlrelu, float post_slope, int dtype,
                                      hipStream_t stream) {
    FS2_CHECK_ARG(!tail_ws || ((uintptr_t)tail_ws & 15) == 0, "conv_gemm_lrelu_io: tail_ws must be 16-byte aligned");
    return conv_gemm_impl(X, ldx, W, bias, R, ldr, Y, ldy, nullptr, nullptr, tail_ws, M, N, Cin, S, taps, dil, pad, act, slope,
                          FS2_ACT_NONE, 0.f, accumulate, out_scale, dtype, stream, res_unlrelu, post_slope);
}

// K-split form of fs2_conv_gemm for contractions with few output tiles and a long reduction (the encoder's k=9 data gradient:
// 48 tiles x 144 K-steps on 256 CUs): `ksplit` workgroups share one output tile, each reduces a contiguous range of Cin
// chunks and stores its partial tile into its own slab of `ws` (f32 scratch, ksplit x M x N, any contents); one more launch
// sums the slabs, applies bias / activation / residual and writes Y.  Only shapes the persistent kernel takes (bf16, Cin % (64 ksplit) == 0, N % 8 == 0,
// 16-byte rows, lens only together with tile_map); FS2_EINVAL otherwise - the caller falls back to fs2_conv_gemm.
extern "C" int fs2_conv_gemm_splitk(const void* X, long ldx, const void* W, const float* bias, const void* R, long ldr, void* Y,
                                    long ldy, const int32_t* lens, const int32_t* tile_map, float* ws, int ksplit, int M, int N,
                                    int Cin, int S, int taps, int dil, int pad, int act, float slope, float out_scale, int dtype,
                                    hipStream_t stream) {
    FS2_CHECK_ARG(X && W && Y && ws, "conv_gemm_splitk: null pointer");
    FS2_CHECK_ARG(M > 0 && N > 0 && Cin > 0 && S > 0 && taps > 0 && dil > 0 && M % S == 0 && ksplit >= 2, "conv_gemm_splitk: bad shape");
    FS2_CHECK_ARG(((uintptr_t)X & 15) == 0 && ((uintptr_t)W & 15) == 0 && ((uintptr_t)ws & 15) == 0 && ldx % 8 == 0, "conv_gemm_splitk: alignment");
    ConvGemmArgs a;
    conv_gemm_fill(a, X, ldx, W, bias, R, ldr, Y, ldy, lens, M, N, Cin, S, taps, dil, pad, act, slope, FS2_ACT_NONE, 0.f, 0, out_scale, dtype);
    FS2_CHECK_ARG(fs2_conv_gemm_p_ok(a, tile_map != nullptr, dtype, ksplit), "conv_gemm_splitk: shape not supported (M=%d N=%d Cin=%d taps=%d ksplit=%d)", M, N, Cin, taps, ksplit);
    fs2_conv_gemm_p_launch(a, tile_map, stream, 0, ksplit, ws, nullptr);
    FS2_CHECK_LAUNCH("conv_gemm_splitk");
    return FS2_OK;
}

// ------------------------------------------------------------------ weight packing
// Master conv weights are STORED tap-major, W[n][j][c] (the nn.Parameter the user sees is the permuted view
// (Cout, Cin, k) of that storage), i.e. already in the K-contiguous order the forward GEMM wants:
//   forward pack  Wf[n][j][c] = W[n][j][c]           (a dtype cast; not needed at all for f32 compute)
//   dgrad pack    Wd[c][j][n] = W[n][k-1-j][c]       (tap flip + transpose)
template <typename T>
__global__ void pack_weight_kernel(const float* __restrict__ w, T* __restrict__ wf, T* __restrict__ wd, int Cout, int Cin, int k) {
    size_t total = (size_t)Cout * Cin * k;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        int c = (int)(i % Cin);
        size_t q = i / Cin;
        int j = (int)(q % k);
        int n = (int)(q / k);
        float v = w[i];
        if (wf) Elem<T>::st(wf + i, v);
        if (wd) Elem<T>::st(wd + ((size_t)c * k + (k - 1 - j)) * Cout + n, v);
    }
}
extern "C" int fs2_pack_weight(const float* w, void* wf, void* wd, int Cout, int Cin, int k, int dtype, hipStream_t stream) {
    FS2_CHECK_ARG(w && (wf || wd), "pack_weight: null pointer");
    size_t total = (size_t)Cout * Cin * k;
    if (total == 0) return FS2_OK;
    int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    if (dtype == FS2_F32) pack_weight_kernel<float><<<grid, 256, 0, stream>>>(w, (float*)wf, (float*)wd, Cout, Cin, k);
    else if (dtype == FS2_BF16) pack_weight_kernel<bf16_t><<<grid, 256, 0, stream>>>(w, (bf16_t*)wf, (bf16_t*)wd, Cout, Cin, k);
    else { fs2_set_error("pack_weight: dtype"); return FS2_EDTYPE; }
    FS2_CHECK_LAUNCH("pack_weight");
    return FS2_OK;
}

// Every data-gradient pack of the model in one launch: block = one 64(n) x 64(c) tile of one tap of one weight,
// transposed through LDS so both the f32 reads (along c) and the packed writes (along n) are contiguous.
template <typename T>
__global__ void __launch_bounds__(256) pack_dgrad_multi_kernel(const float* __restrict__ flat, T* __restrict__ wd_base,
                                                               const int64_t* __restrict__ table, int n_entries) {
    __shared__ float tile[64][65];
    int e = 0;
    while (e + 1 < n_entries && (int)table[(e + 1) * 6 + 5] <= (int)blockIdx.x) ++e;   // <= ~50 entries: linear scan
    const int64_t* d = table + e * 6;
    const float* w = flat + d[0];
    T* wd = wd_base + d[1];
    const int Cout = (int)d[2], Cin = (int)d[3], k = (int)d[4];
    int t = (int)blockIdx.x - (int)d[5];
    const int tn = (Cout + 63) >> 6, tc = (Cin + 63) >> 6;
    const int in_ = t % tn; t /= tn;
    const int ic = t % tc; t /= tc;
    const int j = t;
    const int n0 = in_ * 64, c0 = ic * 64;
    const int lx = threadIdx.x & 63, ly = threadIdx.x >> 6;
#pragma unroll 4
    for (int r = ly; r < 64; r += 4) {                   // read W[n0+r][j][c0+lx]
        int n = n0 + r, c = c0 + lx;
        tile[r][lx] = (n < Cout && c < Cin) ? w[((size_t)n * k + j) * Cin + c] : 0.f;
    }
    __syncthreads();
#pragma unroll 4
    for (int r = ly; r < 64; r += 4) {                   // write Wd[c0+r][k-1-j][n0+lx]
        int c = c0 + r, n = n0 + lx;
        if (c < Cin && n < Cout) Elem<T>::st(wd + ((size_t)c * k + (k - 1 - j)) * Cout + n, tile[lx][r]);
    }
}
extern "C" int fs2_pack_dgrad_multi(const float* flat, void* wd_base, const int64_t* table, int n_entries, int total_tiles,
                                    int dtype, hipStream_t stream) {
    FS2_CHECK_ARG(flat && wd_base && table, "pack_dgrad_multi: null pointer");
    if (n_entries <= 0 || total_tiles <= 0) return FS2_OK;
    if (dtype == FS2_F32) pack_dgrad_multi_kernel<float><<<total_tiles, 256, 0, stream>>>(flat, (float*)wd_base, table, n_entries);
    else if (dtype == FS2_BF16) pack_dgrad_multi_kernel<bf16_t><<<total_tiles, 256, 0, stream>>>(flat, (bf16_t*)wd_base, table, n_entries);
    else { fs2_set_error("pack_dgrad_multi: dtype"); return FS2_EDTYPE; }
    FS2_CHECK_LAUNCH("pack_dgrad_multi");
    return FS2_OK;
}

// ------------------------------------------------------------------ weight gradient (TN contraction over rows)
// dW[n][j][c] (tap-major master layout, fp32, atomically accumulated; lanes run along c -> coalesced atomics)
//   = sum_m dY[m][n] * X[m + j*dil - pad][c]
// Block: 128 (n) x 128 (c) output tile for one tap, split-K over row ranges; K-tile = 32 rows.
// f32: LDS tiles [32 rows][128] f32, fragments by ds_read_b32 (lanes along n / c: conflict-free),
//      v_mfma_f32_32x32x2_f32.
// bf16 inputs are widened to f32 when staged (first version; a bf16-MFMA wgrad with transposing
// LDS reads replaces this in a later revision).

template <typename T>
__global__ void __launch_bounds__(256, 2) conv_wgrad_kernel(WgradArgs a) {
    __shared__ __attribute__((aligned(16))) float sA[2][32][128];  // dY tile  [row][n]
    __shared__ __attribute__((aligned(16))) float sB[2][32][128];  // X tile   [row][c]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int ntn = (a.N + 127) >> 7, ntc = (a.Cin + 127) >> 7;
    int bx = blockIdx.x;
    const int tile_n = bx % ntn; bx /= ntn;
    const int tile_c = bx % ntc; bx /= ntc;
    const int tap = bx;
    const int n0 = tile_n * 128, c0 = tile_c * 128;
    const int shift = tap * a.dil - a.pad;
    const int mbeg = blockIdx.y * a.rows_per_split;
    const int mend = min(a.M, mbeg + a.rows_per_split);
    if (mbeg >= mend) return;
    const T* dY = reinterpret_cast<const T*>(a.dY);
    const T* X = reinterpret_cast<const T*>(a.X);

    // staging: tile 32 rows x 128 cols = 1024 float4 -> 4 per thread: row = (tid>>5) + 8 i, col4 = tid&31
    const int sr = tid >> 5, sc4 = (tid & 31) * 4;
    float4 ra[4], rb[4];
    auto load_tile = [&](int mt) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int m = mt + sr + 8 * i;
            float4 va = make_float4(0.f, 0.f, 0.f, 0.f), vb = va;
            if (m < mend) {
                int n = n0 + sc4;
                if (n < a.N) va = ld4<T>(dY + (size_t)m * a.lddy + n);
                int t = m % a.S, ts = t + shift;
                int c = c0 + sc4;
                if (ts >= 0 && ts < a.S && c < a.Cin) vb = ld4<T>(X + (size_t)(m + shift) * a.ldx + c);
            }
            ra[i] = va; rb[i] = vb;
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<float4*>(&sA[buf][sr + 8 * i][sc4]) = ra[i];
            *reinterpret_cast<float4*>(&sB[buf][sr + 8 * i][sc4]) = rb[i];
        }
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int fl = lane & 31, fh = lane >> 5;
    int nt = (mend - mbeg + 31) / 32;
    load_tile(mbeg);
    store_tile(0);
    __syncthreads();
    for (int it = 0; it < nt; ++it) {
        if (it + 1 < nt) load_tile(mbeg + (it + 1) * 32);
        int buf = it & 1;
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            float af[2], bf[2];
            int row = ks * 2 + fh;
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) af[mb] = sA[buf][row][wm * 64 + mb * 32 + fl];
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) bf[nb] = sB[buf][row][wn * 64 + nb * 32 + fl];
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int nb = 0; nb < 2; ++nb)
                    acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mb], bf[nb], acc[mb][nb], 0, 0, 0);
        }
        if (it + 1 < nt) store_tile((it + 1) & 1);
        __syncthreads();
    }
    // acc[mb][nb][r]: row (n index) = wm*64+mb*32+(r&3)+8*(r>>2)+4*fh ; col (c index) = wn*64+nb*32+fl
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
        int c = c0 + wn * 64 + nb * 32 + fl;
        if (c >= a.Cin) continue;
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int n = n0 + wm * 64 + mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                if (n >= a.N) continue;
                atomicAdd(a.dW + ((size_t)n * a.taps + tap) * a.Cin + c, acc[mb][nb][r]);
            }
    }
}


// ---- bf16 weight gradient on v_mfma_f32_32x32x16_bf16 ------------------------------------------------------
// Both operands are ROW(m)-major in HBM (the reduction index is the slow dimension), so the MFMA fragments
// (8 consecutive k = 8 consecutive ROWS for one column) are read with gfx950's transposing LDS load
// ds_read_b64_tr_b16: a 16-lane group supplies 16 x 8-byte addresses forming a [4 rows][16 cols] block and
// lane i receives column i of the 4 rows.  Tiles are staged untransposed ([64 rows][128 cols] bf16, 256-B rows)
// with the 64-B column chunk XOR-swizzled by (row & 3) so the 4 rows of one transposing read hit 4 different
// bank quarters.
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ uint4 and4(uint4 v, unsigned m) { return make_uint4(v.x & m, v.y & m, v.z & m, v.w & m); }
__device__ __forceinline__ int wg_swz(int row, int c16) { return row * 256 + ((((c16 >> 2) ^ (row & 3))) << 6) + ((c16 & 3) << 4); }

// One workgroup = 128 (n) x 128 (c) outputs for a group of EXACTLY NT adjacent taps (compile-time, so the whole K-tile
// body is straight-line code), over a range of K-tiles.  K-tiles are 64 rows and never straddle a sequence, so the X
// tile is staged ONCE with a halo of (NT-1)*dil rows (rows outside the sequence zero-filled) and every tap of the
// group reads it at a different row offset: the dY and X tiles are fetched once per tap GROUP instead of once per tap.
// What the ISA of the first version showed (r01h): a runtime `tap < ntap` branch around every tap's 4 transposing
// reads + 4 MFMAs, so each MFMA quartet (128 cycles) waited out a full LDS round trip with nothing in flight ->
// ~6600 cycles per K-tile against 1536 cycles of MFMA work.  Now: fragments are double-buffered across the four
// 16-row sub-steps (reads of sub-step s+1 issued before the MFMAs of s), global loads are branch-free (clamped
// address + select) and issued before the K-tile's MFMAs, stored to the other LDS buffer after them.
// NW = waves per workgroup: 4 (each 64 n x 64 c) or 8 (each 64 n x 32 c: two waves per SIMD - the tap-group kernels hold
// 3 x 64 accumulators per 64x64 wave tile, which pins a 4-wave workgroup to ONE wave per SIMD with nothing to switch to
// while a wave waits on LDS or the barrier; halving the wave tile lets two waves share a SIMD)
template <int NT, int NW>
__device__ __forceinline__ void wgrad_bf16_body(const WgradArgs& a, unsigned char* smem, int tile_n, int tile_c, int tap0) {
    constexpr int XROWS = 64 + 8;                       // halo capacity: (NT-1)*dil <= 8
    constexpr int A_BYTES = 64 * 256, X_BYTES = XROWS * 256;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int NBC = NW == 4 ? 2 : 1;                  // 32-column blocks per wave
    constexpr int RS = NW * 4, NI = 64 / RS;             // staging: rows per pass, passes per 64-row tile
    const int wm = NW == 4 ? (wave >> 1) : (wave >> 2), wn = NW == 4 ? (wave & 1) : (wave & 3);
    const int n0 = tile_n * 128, c0 = tile_c * 128;
    const int shift0 = tap0 * a.dil - a.pad;
    const int tps = (a.S + 63) >> 6;                    // K-tiles per sequence
    const int nunits = (a.M / a.S) * tps;
    const int ubeg = blockIdx.y * a.rows_per_split;     // units per split
    const int uend = min(nunits, ubeg + a.rows_per_split);
    if (ubeg >= uend) return;
    const bf16_t* dY = reinterpret_cast<const bf16_t*>(a.dY);
    const bf16_t* X = reinterpret_cast<const bf16_t*>(a.X);

    const int sr = tid >> 4, sc = tid & 15;
    const bool ncol_ok = (n0 + sc * 8) < a.N, ccol_ok = (c0 + sc * 8) < a.Cin;
    const int ncol = ncol_ok ? n0 + sc * 8 : 0, ccol = ccol_ok ? c0 + sc * 8 : 0;     // clamped: always addressable
    const bool x5 = wave < 2;                            // halo rows 64..71 are staged by the first 128 threads
    // TWO register stages (prefetch distance 2 K-tiles: a tile's loads get two tile-times - ~2-4 us - to arrive; with
    // one stage every K-tile waited out the L2/HBM round trip: 2 us per tile against 0.77 us of MFMA work, at ANY split count)
    uint4 ra[2][NI], rb[2][NI], rb4[2] = {make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0)};
    bool live_t[2] = {false, false};
    const bool do_bias = a.dbias != nullptr && tile_c == 0 && tap0 == 0;     // block-uniform
    float bsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    // INTERIOR K-tiles (all 64 rows valid, whole halo inside the sequence, full column tiles: 13 of 15 at T = 925) take a
    // fast path: one 32-bit multiply for the thread's first row, scalar strides for the rest, no masks (r01h PMC: 6.9
    // VALU instructions per MFMA in the first version - address arithmetic and selects - made the loop ISSUE-bound).
    // EDGE tiles clamp the row index and zero invalid rows when storing.
    const bool col_full = (n0 + 128 <= a.N) && (c0 + 128 <= a.Cin);
    const unsigned char* dYb = reinterpret_cast<const unsigned char*>(dY);
    const unsigned char* Xb = reinterpret_cast<const unsigned char*>(X);
    const unsigned dy_rs = (unsigned)a.lddy * 2u, x_rs = (unsigned)a.ldx * 2u;      // row strides in bytes
    unsigned okmask_t[2] = {0, 0};                       // edge tiles: bit i: ra[i] valid, bit 4+i: rb[i] valid, bit 8: rb4
    bool edge_t[2] = {false, false};
    // ONE straight-line load sequence per K-tile, issued unconditionally (a tile index past the end re-reads the first
    // tile and is marked dead).  With the loads inside `if (more)` / `if (interior) ... else ...` blocks, hipcc's waitcnt
    // insertion put an s_waitcnt vmcnt(0) at the control-flow merge in front of the MFMA block (r01i ISA) - the prefetch
    // was waited for immediately and every K-tile paid the full L2/HBM round trip (2 us per tile at any split depth).
    auto load_tile = [&](auto setc, int u) {
        constexpr int SET = decltype(setc)::value;
        const bool valid = u < uend;
        const int uu = valid ? u : ubeg;
        const int seq = uu / tps, t0 = (uu - seq * tps) * 64;
        const int tend = a.lens ? min(a.lens[seq], a.S) : a.S;      // rows >= tend carry zero gradient
        live_t[SET] = valid && t0 < tend;                            // a dead K-tile is fetched (cheaply, once) but never multiplied
        edge_t[SET] = !col_full || (t0 + 64 > tend) || (t0 + shift0 < 0) || (t0 + shift0 + XROWS > a.S);
        const unsigned base = (unsigned)(seq * a.S);                 // rows < M: 32-bit
        unsigned ok = 0;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int t = t0 + sr + RS * i;
            const size_t o = (size_t)(base + (unsigned)min(t, a.S - 1)) * dy_rs + (unsigned)ncol * 2u;
            ra[SET][i] = *reinterpret_cast<const uint4*>(dYb + o);
            ok |= (ncol_ok && t < tend) ? (1u << i) : 0u;
        }
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int t = t0 + shift0 + sr + RS * i;
            const size_t o = (size_t)(base + (unsigned)min(max(t, 0), a.S - 1)) * x_rs + (unsigned)ccol * 2u;
            rb[SET][i] = *reinterpret_cast<const uint4*>(Xb + o);
            ok |= (ccol_ok && t >= 0 && t < a.S) ? (16u << i) : 0u;
        }
        {
            const int t = t0 + shift0 + sr + 64;                     // halo rows 64..71: needed from the first 128 threads only,
            const size_t o = (size_t)(base + (unsigned)min(max(t, 0), a.S - 1)) * x_rs + (unsigned)ccol * 2u;
            rb4[SET] = *reinterpret_cast<const uint4*>(Xb + o);      // loaded by everyone to keep the sequence branch-free
            ok |= (ccol_ok && t >= 0 && t < a.S) ? 256u : 0u;
        }
        okmask_t[SET] = ok;
    };
    auto store_tile = [&](auto setc, int buf) {
        constexpr int SET = decltype(setc)::value;
        const unsigned okmask = okmask_t[SET];
        unsigned char* As = smem + buf * (A_BYTES + X_BYTES);
        unsigned char* Bs = As + A_BYTES;
        if (edge_t[SET]) {                               // block-uniform
#pragma unroll
            for (int i = 0; i < NI; ++i) ra[SET][i] = and4(ra[SET][i], 0u - ((okmask >> i) & 1u));
#pragma unroll
            for (int i = 0; i < NI; ++i) rb[SET][i] = and4(rb[SET][i], 0u - ((okmask >> (4 + i)) & 1u));
            rb4[SET] = and4(rb4[SET], 0u - ((okmask >> 8) & 1u));
        }
#pragma unroll
        for (int i = 0; i < NI; ++i) *reinterpret_cast<uint4*>(As + wg_swz(sr + RS * i, sc)) = ra[SET][i];
#pragma unroll
        for (int i = 0; i < NI; ++i) *reinterpret_cast<uint4*>(Bs + wg_swz(sr + RS * i, sc)) = rb[SET][i];
        if (x5) *reinterpret_cast<uint4*>(Bs + wg_swz(sr + 64, sc)) = rb4[SET];
        if (do_bias) {                                   // bias gradient rides on the dY tile already in registers
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const uint32_t* u32 = reinterpret_cast<const uint32_t*>(&ra[SET][i]);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    bsum[2 * e] += __uint_as_float(u32[e] << 16);
                    bsum[2 * e + 1] += __uint_as_float(u32[e] & 0xffff0000u);
                }
            }
        }
    };
    f32x16 acc[NT][2][NBC];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NBC; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][i][j][r] = 0.f;
    const int li = lane & 15, g = lane >> 4, h = g >> 1;
    const int rrow = 8 * h + (li >> 2);
    const int acol = wm * 64 + 16 * (g & 1) + 4 * (li & 3);
    const int bcol = wn * (32 * NBC) + 16 * (g & 1) + 4 * (li & 3);
    typedef __attribute__((address_space(3))) s16x4* lds_s4;
    // per-lane LDS byte offsets of the transposing reads, computed ONCE: sub-step ks adds 16 rows = 4096 B and the
    // buffer index adds a constant, both of which (row & 3 unchanged) fold into the ds_read immediate offset.
    auto tr_off = [&](int row, int col) -> unsigned { return (unsigned)(row * 256 + ((((col >> 5) ^ (row & 3))) << 6) + ((col & 31) << 1)); };
    unsigned offA[2][2], offB[NT][NBC][2];               // [blk][lo/hi], [tap][blk][lo/hi]
#pragma unroll
    for (int blk = 0; blk < 2; ++blk)
#pragma unroll
        for (int hl = 0; hl < 2; ++hl) {
            offA[blk][hl] = tr_off(rrow + 4 * hl, acol + blk * 32);
            if (blk < NBC) {
#pragma unroll
                for (int t = 0; t < NT; ++t) offB[t][blk][hl] = A_BYTES + tr_off(rrow + 4 * hl + t * a.dil, bcol + blk * 32);
            }
        }
    const unsigned smem_u = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
    auto tr_read = [&](unsigned off) -> s16x4 {
        return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(size_t)(smem_u + off));
    };
    s16x8 af[2][2], bf[2][NT][NBC];                      // [set][..]: fragment double buffer
    auto read_frags = [&](int set, int bufoff, int ks) {
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
            s16x4 lo = tr_read(offA[blk][0] + bufoff + ks * 4096);
            s16x4 hi = tr_read(offA[blk][1] + bufoff + ks * 4096);
            af[set][blk] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        }
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int blk = 0; blk < NBC; ++blk) {
                s16x4 lo = tr_read(offB[t][blk][0] + bufoff + ks * 4096);
                s16x4 hi = tr_read(offB[t][blk][1] + bufoff + ks * 4096);
                bf[set][t][blk] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            }
    };
    auto mma = [&](int set) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int nb = 0; nb < NBC; ++nb)
                    acc[t][mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                        __builtin_bit_cast(bf16x8, af[set][mb]), __builtin_bit_cast(bf16x8, bf[set][t][nb]), acc[t][mb][nb], 0, 0, 0);
    };
    typedef std::integral_constant<int, 0> I0;
    typedef std::integral_constant<int, 1> I1;
    // tile u lives in LDS buffer (u - ubeg) & 1 while it is multiplied; tile u+1 waits in register stage ((u+1-ubeg) & 1),
    // tile u+2 is being fetched into stage ((u-ubeg) & 1) (whose previous content, tile u, went to LDS one step earlier).
    load_tile(I0{}, ubeg);
    load_tile(I1{}, ubeg + 1);
    bool live_lds[2] = {live_t[0], false};
    if (live_t[0]) store_tile(I0{}, 0);
    __syncthreads();
    // one K-tile out of LDS buffer BUF (compile-time, so every ds_read offset is an immediate)
    auto ktile = [&](auto bufc, int u) {
        constexpr int BUF = decltype(bufc)::value;
        constexpr int bufoff = BUF * (A_BYTES + X_BYTES);
        const bool live = live_lds[BUF] && !FS2_DEV_DBG(a.dbg & 2);
        if (live) read_frags(0, bufoff, 0);
        load_tile(std::integral_constant<int, BUF>{}, u + 2);
        if (live) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                if (ks < 3) read_frags((ks + 1) & 1, bufoff, ks + 1);
                mma(ks & 1);
            }
        }
        // tile u+1 (stage BUF^1, fetched one step ago) -> the other LDS buffer
        const bool nxt = (u + 1 < uend) && live_t[BUF ^ 1];
        live_lds[BUF ^ 1] = nxt;
        if (nxt) store_tile(std::integral_constant<int, BUF ^ 1>{}, BUF ^ 1);
        __syncthreads();
    };
    for (int u = ubeg; u < uend; u += 2) {
        ktile(I0{}, u);
        if (u + 1 < uend) ktile(I1{}, u + 1);
    }
    if (FS2_DEV_DBG(a.dbg & 1)) return;
    const int fl = lane & 31, fh = lane >> 5;
    // split-K: with a slab, this split's partial tile goes to its OWN copy of dW (plain stores, a half-wave writes 32 consecutive
    // c = one full 128-byte line per store) and wgrad_finalize_kernel sums the splits in a fixed order (bit-reproducible
    // gradients); without one, fp32 atomics straight into the master gradient (round-1/2 path: a full-tile burst of ~2 TB/s
    // device-scope atomics per split - the ~80 us tail of the k=9 launch)
    float* const outW = a.slab ? a.slab + (size_t)blockIdx.y * a.slab_stride : a.dW;
    if (do_bias) {                                       // reduce the 16 row-threads of every column group through LDS
        float* red = reinterpret_cast<float*>(smem);     // [16][128]; the operand tiles are dead (last loop barrier passed)
#pragma unroll
        for (int e = 0; e < 8; ++e) red[sr * 128 + sc * 8 + e] = bsum[e];
        __syncthreads();
        if (tid < 128 && n0 + tid < a.N) {
            float t = 0.f;
#pragma unroll
            for (int r = 0; r < RS; ++r) t += red[r * 128 + tid];
            if (a.slab) outW[(size_t)a.N * a.taps * a.Cin + n0 + tid] = t;
            else atomicAdd(a.dbias + n0 + tid, t);
        }
    }
#pragma unroll
    for (int nb = 0; nb < NBC; ++nb) {
        int c = c0 + wn * (32 * NBC) + nb * 32 + fl;
        if (c >= a.Cin) continue;
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int n = n0 + wm * 64 + mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                if (n >= a.N) continue;
                float* dst = outW + ((size_t)n * a.taps + tap0) * a.Cin + c;
                if (a.slab) {
#pragma unroll
                    for (int t = 0; t < NT; ++t) dst[(size_t)t * a.Cin] = acc[t][mb][nb][r];
                } else {
#pragma unroll
                    for (int t = 0; t < NT; ++t) atomicAdd(dst + (size_t)t * a.Cin, acc[t][mb][nb][r]);
                }
            }
    }
}

// dW[i] += sum_s slab[s][i] over the nw = N*taps*Cin weight entries; dbias[n] += sum_s slab[s][nw + n] (splits in index order: the
// result does not depend on which workgroup finished first).  One thread = 4 consecutive floats.
__global__ void __launch_bounds__(256) wgrad_finalize_kernel(float* __restrict__ dW, float* __restrict__ dbias, const float* __restrict__ slab,
                                                             long slab_stride, int nsplit, long nw, int N) {
    const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i < nw) {
        float4 v = *reinterpret_cast<const float4*>(dW + i);
        for (int sp = 0; sp < nsplit; ++sp) {
            const float4 x = *reinterpret_cast<const float4*>(slab + (size_t)sp * slab_stride + i);
            v.x += x.x; v.y += x.y; v.z += x.z; v.w += x.w;
        }
        *reinterpret_cast<float4*>(dW + i) = v;
    } else if (dbias && i - nw < N) {                    // (nw % 4 == 0, N % 4 == 0)
        const long n = i - nw;
        float4 v = *reinterpret_cast<const float4*>(dbias + n);
        for (int sp = 0; sp < nsplit; ++sp) {
            const float4 x = *reinterpret_cast<const float4*>(slab + (size_t)sp * slab_stride + nw + n);
            v.x += x.x; v.y += x.y; v.z += x.z; v.w += x.w;
        }
        *reinterpret_cast<float4*>(dbias + n) = v;
    }
}

// grid.x = (n-tile, c-tile, tap group): the first a.g3 groups take NT taps each, one more group (if REM) the remaining
// REM taps - k = 9: 3+3+3, k = 5: 3+2, k = 3: 3, k = 1: <1,0> - all in ONE launch (one split-K depth, one set of atomics).
template <int NT, int REM, int NW>
__global__ void __launch_bounds__(64 * NW, NW == 4 ? 1 : 2) conv_wgrad_bf16_kernel(WgradArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // [buf][dY 64 rows | X 72 rows][256 B]
    const int ntn = (a.N + 127) >> 7, ntc = (a.Cin + 127) >> 7;
    int bx = blockIdx.x;
    const int tile_n = bx % ntn; bx /= ntn;
    const int tile_c = bx % ntc; bx /= ntc;
    if (REM == 0 || bx < a.g3) wgrad_bf16_body<NT, NW>(a, smem, tile_n, tile_c, bx * NT);
    else wgrad_bf16_body<(REM ? REM : 1), NW>(a, smem, tile_n, tile_c, a.g3 * NT);
}

// the largest workspace any launch plans for (one slab beyond it: no workspace, the atomic path): callers allocate this once
extern "C" int fs2_conv_wgrad_ws_cap(void) { return (int)FS2_WGRAD_WS_CAP; }
extern "C" int fs2_conv_wgrad_ws_bytes(int M, int N, int Cin, int S, int taps, int dil, int has_lens, int dtype) {
    if (dtype != FS2_BF16 || M <= 0 || N <= 0 || Cin <= 0 || S <= 0 || taps <= 1) return 0;   // (one-tap launches: see conv_wgrad_impl)
    const WgradPlan p = wgrad_plan(M, N, Cin, S, taps, dil, has_lens != 0, true);
    const long b = (long)p.splits * ((long)N * taps * Cin + N) * 4;
    return b > 0x7fffffffL ? 0 : (int)b;                 // (a single slab beyond 2 GB: no workspace -> the atomic path)
}

template <int NT, int REM, int NW>
static void launch_wgrad_bf16(WgradArgs a, int S_eff, int g3, hipStream_t stream, const WgradPlan* plan = nullptr) {
    a.S = S_eff;
    a.g3 = g3;
    static const int dbg = fs2_dev_env("FS2_WGRAD_DBG", 0);
    a.dbg = dbg;
    const int dyn = 2 * (64 * 256 + 72 * 256);
    static Fs2DevOnce once;
    once.run([&] { (void)hipFuncSetAttribute((const void*)conv_wgrad_bf16_kernel<NT, REM, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, dyn); });
    const int groups = g3 + (REM ? 1 : 0);
    long tiles = (long)fs2_cdiv(a.N, 128) * fs2_cdiv(a.Cin, 128) * groups;
    int tps = (a.S + 63) / 64;
    long units = (long)(a.M / a.S) * tps;
    if (plan) {                                           // slab split-K: depth from wgrad_plan (the workspace was sized by it)
        a.rows_per_split = plan->ups;
        dim3 grid((unsigned)tiles, (unsigned)plan->splits);
        conv_wgrad_bf16_kernel<NT, REM, NW><<<grid, 64 * NW, dyn, stream>>>(a);
        return;
    }
    // Split-K depth (ATOMIC path).  Every split ends with a full-tile burst of fp32 atomics (~2 TB/s, all CUs at once) and every
    // workgroup pays ~10 us of prologue/epilogue, so the depth is bounded from both sides (r01h sweep, tools/bench_wgrad.py):
    //   short reductions (encoder, 96 K-tiles)  -> ~192 workgroups;  1-tap GEMMs (2 workgroups per CU) -> ~384;  else ~768;
    //   and at least 8 (1-tap) / 16 (few-tile conv) K-tiles per workgroup when the reduction is long.
    static const int wg_env = fs2_dev_env("FS2_WGRAD_WGS", 0);
    static const int ups_env = fs2_dev_env("FS2_WGRAD_MINUPS", 0);
    // r02 same-box A/B of the WHOLE step (tools/ab_env.py): these launches run on the side stream next to the data-gradient
    // chain, and fewer, longer workgroups disturb it less than the split depth that is fastest in isolation: tap-group kernels
    // at 192 workgroups, one-tap at 192 (profiles/r02y_ab_env*.log)
    int wg_target = units < 256 ? 192 : (NT == 1 ? 192 : (NW == 8 ? 192 : 768));
    int min_ups = NT == 1 ? 8 : ((tiles <= 8 && units >= 512) ? 16 : 4);
    if (wg_env) wg_target = wg_env;
    static const int wg_env1 = fs2_dev_env("FS2_WGRAD_WGS1", 0), wg_env3 = fs2_dev_env("FS2_WGRAD_WGS3", 0);
    if (NT == 1 && wg_env1) wg_target = wg_env1;
    if (NT > 1 && wg_env3) wg_target = wg_env3;
    if (ups_env) min_ups = ups_env;
    long want = (NW == 8 && NT > 1) ? (wg_target / tiles) : (wg_target + tiles - 1) / tiles;   // floor: never spill into a second round
    if (want < 1) want = 1;
    long ups = (units + want - 1) / want;
    if (ups < min_ups) ups = min_ups;
    a.rows_per_split = (int)ups;
    dim3 grid((unsigned)tiles, (unsigned)fs2_cdiv(units, ups));
    conv_wgrad_bf16_kernel<NT, REM, NW><<<grid, 64 * NW, dyn, stream>>>(a);
}

extern "C" int fs2_colsum(const void* x, long ldx, float* out, int M, int N, int dtype, hipStream_t stream);

static int conv_wgrad_impl(const void* dY, long lddy, const void* X, long ldx, float* dW, float* dbias, const int32_t* lens,
                           int M, int N, int Cin, int S, int taps, int dil, int pad, int dtype, float* ws, long ws_bytes,
                           hipStream_t stream) {
    FS2_CHECK_ARG(dY && X && dW, "conv_wgrad: null pointer");
    FS2_CHECK_ARG(M >= 0 && N > 0 && Cin > 0 && S > 0 && taps > 0, "conv_wgrad: bad shape");
    FS2_CHECK_ARG(N % 4 == 0 && Cin % 4 == 0 && lddy % 4 == 0 && ldx % 4 == 0, "conv_wgrad: N/Cin/ld must be multiples of 4");
    if (M == 0) return FS2_OK;
    WgradArgs a;
    a.dY = dY; a.lddy = lddy; a.X = X; a.ldx = ldx; a.dW = dW; a.dbias = dbias; a.lens = lens; a.M = M; a.N = N; a.Cin = Cin; a.S = S; a.taps = taps;
    a.dil = dil; a.pad = pad; a.g3 = 0; a.dbg = 0; a.slab = nullptr; a.slab_stride = 0;
    long tiles = (long)fs2_cdiv(N, 128) * fs2_cdiv(Cin, 128) * taps;
    // split rows so that ~1024 workgroups exist, each covering a multiple of 32 rows (>= 256 rows).
    long want = (1024 + tiles - 1) / tiles;
    long rps = (M + want - 1) / want;
    rps = ((rps + 63) / 64) * 64;
    if (rps < 256) rps = 256;
    a.rows_per_split = (int)rps;
    int splits = fs2_cdiv(M, rps);
    dim3 grid((unsigned)tiles, (unsigned)splits);
    bool bias_fused = false;
    if (dtype == FS2_F32) conv_wgrad_kernel<float><<<grid, 256, 0, stream>>>(a);
    else if (dtype == FS2_BF16) {
        bool fast = (N % 8 == 0) && (Cin % 8 == 0) && (lddy % 8 == 0) && (ldx % 8 == 0) && (((uintptr_t)dY | (uintptr_t)X) & 15) == 0;
        static const int wg_waves = fs2_dev_env("FS2_WGRAD_WAVES", 8);   // dev A/B: 4 | 8
        static const int wg_waves1 = fs2_dev_env("FS2_WGRAD_WAVES1", 4);            // dev A/B: 4 | 8
        const bool conv_ok = fast && taps > 1 && 2 * dil <= 8 && M % S == 0;
        const bool one_ok = fast && taps == 1;
        // slab split-K (round 3): needs the caller's workspace (fs2_conv_wgrad_ws_bytes) and 16-byte addressable gradients
        static const int slab_env = fs2_dev_env("FS2_WGRAD_SLAB", 1);   // dev A/B: 0 = atomics even with a workspace
        WgradPlan plan = wgrad_plan(M, N, Cin, S, taps, dil, lens != nullptr, true);
        const long need = (long)plan.splits * ((long)N * taps * Cin + N) * 4;
        // One-tap (Linear) gradients stay on the atomic path: they are short HBM-bound launches whose split depth was tuned for
        // it, and with slabs they measured SLOWER on the first round-3 run (r03b: fc 28 -> 42 us, mel 19 -> 44, encoder shapes
        // 17-21 -> 22-23; only the K = 1024 FFN shape gained, 70 -> 62): 64-128 splits of a 64 KB tile turn the finalize pass into
        // a latency chain.  one_slab (dev builds) re-enables it for A/B.
        static const int one_slab = fs2_dev_env("FS2_WGRAD_SLAB1", 0);
        const bool slab = slab_env && ws && ws_bytes >= need && ((one_ok && one_slab) || conv_ok) && (((uintptr_t)dW | (uintptr_t)dbias | (uintptr_t)ws) & 15) == 0 &&
                          ((long)N * taps * Cin) % 4 == 0;
        if (slab) { a.slab = ws; a.slab_stride = (long)N * taps * Cin + N; }
        if (one_ok) {
            const int S_eff = (lens && M % S == 0) ? S : M;                 // no taps, no lens: one "sequence" of M rows
            // round 3: the LDS-DMA kernel's one-tap form (four 32 KiB buffers: three K-tiles in flight per CU instead of the
            // register-staged kernel's two) with the ATOMIC epilogue - these launches are HBM-bound and short, slabs lose here
            static const int one_tg = fs2_dev_env("FS2_WGRAD_TG1", 1);
            // workgroups to aim for: 128 (same-box sweeps of the whole step, r03e / r03f: 64: 8.73, 96: 8.68, 128: 8.66, 160: 8.66,
            // 192: 9.53 vs 9.47 at 128 on the other box, 256: 8.73, 384+: worse; the register-staged kernel: 8.74): a workgroup holds
            // 128 KiB of LDS, i.e. a whole CU that the data-gradient chain's persistent kernels cannot use meanwhile
            static const int one_wgs = fs2_dev_env("FS2_WGRAD_TG1_WGS", 128);
            bool done = false;
            if (one_tg && !slab && (double)M * (lddy > ldx ? lddy : ldx) * 2 < 4.0e9) {
                WgradPlan p1 = {};
                p1.g_first = 1; p1.n_first = 1; p1.n_rest = 0; p1.share = 1;
                p1.tiles = fs2_cdiv(N, 128) * fs2_cdiv(Cin, 128);
                // 128 (n) x 256 (c) tiles (round 6: a dY tile fetched once per 256 columns of X, a wave's two X runs against the same two dY
                // fragments - 2 LDS instructions per MFMA instead of 3; three 48 KiB ring buffers).  Same box, profiles/r06x_bench_wgrad_wide.log:
                // decoder w_2 (Cin = 1024) 69.6 -> 55.5 us, decoder QKV (N = 768) 53.4 -> 48.3; with few tiles it LOSES (decoder fc, 2 wide
                // tiles: 24.5 -> 29.4; every encoder shape + 2.6 - 4.3 us): only launches with >= 6 wide tiles and a long row stream take it.
                // FS2_WGRAD_TG1_WIDE (dev builds): 0 = never, 2 = whenever Cin % 256 == 0.
                static const int one_wide = fs2_dev_env("FS2_WGRAD_TG1_WIDE", 1);
                const long wide_tiles = (long)fs2_cdiv(N, 128) * (Cin / 256);
                if (Cin % 256 == 0 && (one_wide == 2 || (one_wide == 1 && M >= 16384 && wide_tiles >= 6))) { p1.share = 2; p1.tiles = (int)wide_tiles; }
                p1.units = (M / S_eff) * ((S_eff + 63) / 64);
                long want = p1.tiles >= one_wgs ? 1 : (one_wgs + p1.tiles / 2) / p1.tiles;
                const long max_by_units = p1.units / 8 > 0 ? p1.units / 8 : 1;
                if (want > max_by_units) want = max_by_units;
                p1.ups = (int)((p1.units + want - 1) / want);
                p1.splits = fs2_cdiv(p1.units, p1.ups);
                WgradArgs a1 = a;
                a1.S = S_eff;
                done = fs2_wgrad_tg_launch(a1, p1, stream);
            }
            if (!done) {
                if (wg_waves1 == 8) launch_wgrad_bf16<1, 0, 8>(a, S_eff, 1, stream, slab ? &plan : nullptr);
                else launch_wgrad_bf16<1, 0, 4>(a, S_eff, 1, stream, slab ? &plan : nullptr);
            }
            bias_fused = true;
        }
        else if (conv_ok && slab && plan.share && (double)M * (lddy > ldx ? lddy : ldx) * 2 < 4.0e9 &&
                 fs2_wgrad_tg_launch(a, plan, stream)) {
            // the LDS-DMA tap-group kernel (fs2_wgrad.hip): groups of up to 5 taps share one set of X fragment reads
            bias_fused = true;
        }
        else if (conv_ok) {
            WgradPlan p3 = wgrad_plan(M, N, Cin, S, taps, dil, lens != nullptr, false);   // 3-tap groups, per-tap fragment reads
            const int g3 = taps / 3, rem = taps - 3 * g3;
            const WgradPlan* pp = nullptr;
            if (slab) {      // (dil > 1 with a workspace: the per-tap kernels with slab stores; same split depth rule)
                p3.ups = plan.ups; p3.splits = plan.splits; pp = &p3;
                const int tps = (S + 63) / 64; const long units = (long)(M / S) * tps;
                p3.splits = fs2_cdiv(units, p3.ups);
            }
            if (wg_waves == 8) {
                if (rem == 0) launch_wgrad_bf16<3, 0, 8>(a, S, g3, stream, pp);
                else if (rem == 2) launch_wgrad_bf16<3, 2, 8>(a, S, g3, stream, pp);
                else launch_wgrad_bf16<3, 1, 8>(a, S, g3, stream, pp);
            } else {
                if (rem == 0) launch_wgrad_bf16<3, 0, 4>(a, S, g3, stream, pp);
                else if (rem == 2) launch_wgrad_bf16<3, 2, 4>(a, S, g3, stream, pp);
                else launch_wgrad_bf16<3, 1, 4>(a, S, g3, stream, pp);
            }
            bias_fused = true;
        }
        else conv_wgrad_kernel<bf16_t><<<grid, 256, 0, stream>>>(a);
        if (a.slab && bias_fused) {
            const long nw = (long)N * taps * Cin;
            const long nthreads = (nw + N + 3) / 4;
            wgrad_finalize_kernel<<<(unsigned)((nthreads + 255) / 256), 256, 0, stream>>>(dW, dbias, ws, a.slab_stride, plan.splits, nw, N);
        }
    } else { fs2_set_error("conv_wgrad: dtype"); return FS2_EDTYPE; }
    FS2_CHECK_LAUNCH("conv_wgrad");
    if (dbias && !bias_fused) return fs2_colsum(dY, lddy, dbias, M, N, dtype, stream);   // slow paths: separate column-sum pass
    return FS2_OK;
}

extern "C" int fs2_conv_wgrad(const void* dY, long lddy, const void* X, long ldx, float* dW, float* dbias, const int32_t* lens,
                              int M, int N, int Cin, int S, int taps, int dil, int pad, int dtype, hipStream_t stream) {
    return conv_wgrad_impl(dY, lddy, X, ldx, dW, dbias, lens, M, N, Cin, S, taps, dil, pad, dtype, nullptr, 0, stream);
}

extern "C" int fs2_conv_wgrad_ws(const void* dY, long lddy, const void* X, long ldx, float* dW, float* dbias, const int32_t* lens,
                                 int M, int N, int Cin, int S, int taps, int dil, int pad, int dtype, float* ws, long ws_bytes,
                                 hipStream_t stream) {
    return conv_wgrad_impl(dY, lddy, X, ldx, dW, dbias, lens, M, N, Cin, S, taps, dil, pad, dtype, ws, ws_bytes, stream);
}

// ------------------------------------------------------------------ column sums (bias gradients)
// out[n] += sum_m x[m][n].  grid (ceil(N/64), row-splits); each wave owns 64 columns (coalesced 256-B rows for f32).
template <typename T>
__global__ void colsum_kernel(const T* __restrict__ x, long ldx, float* __restrict__ out, int M, int N, int rows_per_block) {
    __shared__ float s[4][64];
    int col = blockIdx.x * 64 + (threadIdx.x & 63);
    int w = threadIdx.x >> 6;
    int mbeg = blockIdx.y * rows_per_block, mend = min(M, mbeg + rows_per_block);
    float acc = 0.f;
    if (col < N)
        for (int m = mbeg + w; m < mend; m += 4) acc += Elem<T>::ld(x + (size_t)m * ldx + col);
    s[w][threadIdx.x & 63] = acc;
    __syncthreads();
    if (w == 0 && col < N) atomicAdd(out + col, s[0][threadIdx.x] + s[1][threadIdx.x] + s[2][threadIdx.x] + s[3][threadIdx.x]);
}
extern "C" int fs2_colsum(const void* x, long ldx, float* out, int M, int N, int dtype, hipStream_t stream) {
    FS2_CHECK_ARG(x && out, "colsum: null pointer");
    if (M == 0 || N == 0) return FS2_OK;
    int rpb = 256;
    dim3 grid(fs2_cdiv(N, 64), fs2_cdiv(M, rpb));
    if (dtype == FS2_F32) colsum_kernel<float><<<grid, 256, 0, stream>>>((const float*)x, ldx, out, M, N, rpb);
    else if (dtype == FS2_BF16) colsum_kernel<bf16_t><<<grid, 256, 0, stream>>>((const bf16_t*)x, ldx, out, M, N, rpb);
    else { fs2_set_error("colsum: dtype"); return FS2_EDTYPE; }
    FS2_CHECK_LAUNCH("colsum");
    return FS2_OK;
}
