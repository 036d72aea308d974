// fs2_wgrad.hip - bf16 weight gradient of Conv1d (dil == 1, k >= 2) as an LDS-DMA, tap-sharing, split-K kernel (gfx950, round 3).
//
//   dW[n][j][c] += sum_m dY[m][n] * X[m + j - pad][c]            (autograd of transformer/SubLayers.py:87-88 w_1 k = 9,
//                                                                  transformer/Layers.py:129-137 PostNet k = 5, model/modules.py:209-240 k = 3)
// A TN contraction: the reduction index m is the SLOW dimension of both operands, so MFMA fragments (8 consecutive k = 8
// consecutive rows of one column) come out of LDS through gfx950's transposing read ds_read_b64_tr_b16, as in the round-1/2
// kernel (fs2_gemm.hip: conv_wgrad_bf16_kernel, which stays for one-tap launches, dil > 1 and callers without a workspace).
// What round 2's profile said about that kernel (4.26 ms per step, 13-17 % MFMA busy, 1.8-2.5 LDS instructions
// per MFMA, ~80 us atomic tail per launch; r01 ablation: register staging ALONE 130 us of a 310 us launch) and what changes here:
//   * TAP SHARING.  One workgroup owns 128 (n) x 128 (c) outputs for a group of up to FIVE adjacent taps (k = 5: one group,
//     k = 9: 5 + 4, k = 3: one) instead of three: the dY / X tiles of a K-tile are staged once (twice) instead of 2x (3x).  A lane
//     of k-group h needs, for tap t, rows 8h + t .. 8h + t + 7 of its X column; it reads the 12-row run 8h .. 8h + 11 ONCE (three
//     transposing reads) and takes tap t's operand out of registers: even t = dwords t/2 .. t/2 + 3 as they are, odd t = four
//     v_alignbit_b32.  Per wave (64 n x 32 c x 5 taps) and 16-row sub-step: 7 LDS reads for 10 MFMAs (was 10 for 6).
//   * LDS-DMA STAGING.  global_load_lds_dwordx4 (1 KiB per wave-instruction, lane-linear destination) into a ring of THREE
//     34 KiB buffers, every wave issuing its 4 - 5 pieces of the tile two K-tiles ahead, counted vmcnt + ONE raw barrier per
//     K-tile.  No staging registers (the 5-tap accumulators need 160 of the 256 a wave may hold at two waves per SIMD: the
//     register-staged form spilled 57), no ds_write pass (16-byte LDS stores cost 13 cycles per wave-instruction on the VGPR ->
//     LDS path: 430 cycles per K-tile), no address / mask VALU in the MFMA stream.  The 64-byte column groups are XOR-swizzled
//     with (row & 3) so that the four rows of a transposing read fall into four bank quarters; the DMA destination being
//     lane-linear, the permutation is applied to each lane's SOURCE column (same involution on the read side).
//   * EDGES.  DMA cannot transform data, so rows outside the tile's sequence are fetched from clamped (always addressable) rows
//     and then ZEROED IN LDS by the wave that fetched them: in the first / last K-tile of a sequence (2 of 15 at T = 925) each
//     wave, once its own pieces have landed and before the barrier that publishes the tile, stores 16 zero bytes over every
//     chunk of a dY row >= lens[b] or an X row outside [0, S) (lane l of a piece owns exactly its own 16 bytes).  The MFMA loop
//     has ONE form (a masked instantiation of it beside the plain one made hipcc spill ~1500 registers).  Column tails (N or Cin
//     not a multiple of 128) need nothing: their outputs are never stored.
//   * ATOMIC-FREE SPLIT-K.  A split stores its partial tile with plain stores into its own f32 copy of dW (the caller's
//     workspace); wgrad_finalize_kernel (fs2_gemm.hip) sums the splits in index order: bit-reproducible, no atomic tail.
//     The bias gradient rides along: waves holding column block c = 0 sum their dY fragments with v_dot2c_f32_bf16.
//   * Fully padded K-tiles (t0 >= lens[b]) are skipped BEFORE they are fetched (round 2 fetched them and skipped the MFMAs).
#include "fs2_gemm.h"
#include "fs2_wgrad.h"

typedef short tg_s16x4 __attribute__((ext_vector_type(4)));
typedef short tg_s16x8 __attribute__((ext_vector_type(8)));
typedef unsigned tg_u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned tg_u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 tg_bf16x2 __attribute__((ext_vector_type(2)));

static constexpr int TG_A_BYTES = 64 * 256;              // dY tile: 64 rows x 128 n (bf16)
// X tile: 64 rows (+ halo of NT - 1 <= 4 rows, rounded up to whole 4-row DMA pieces: 72) x 128 c
// CB (one-tap launches only): 128-column X images per workgroup tile.  CB = 2 is the 128 (n) x 256 (c) tile of round 6: a wave owns
// 64 x (32 + 32) outputs - two X runs against the same two dY fragments (2 LDS instructions per MFMA instead of 3) - and a dY tile is
// fetched once per 256 columns of X instead of once per 128
template <int NT, int CB = 1> struct TgCfg {
    static constexpr int X_ROWS = NT == 1 ? 64 : 72;
    static constexpr int X_BYTES = X_ROWS * 256;
    static constexpr int BUF_BYTES = TG_A_BYTES + CB * X_ROWS * 256;
    // ring depth: tile i is fetched NBUF - 1 tiles ahead.  One-tap launches are HBM-bound (a 32 KiB tile feeds 16 MFMAs per
    // wave): FOUR 32 KiB buffers keep three tiles in flight per CU (the register-staged kernel held two and moved 1.6 TB/s)
    static constexpr int NBUF = (NT == 1 && CB == 1) ? 4 : 3;
};

template <int N> __device__ __forceinline__ void tg_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

// byte offset of element (row, col) in a swizzled [rows][128] bf16 tile image (col a multiple of 4: transposing reads take 8 bytes)
__device__ __forceinline__ unsigned tg_off(int row, int col) {
    return (unsigned)(row * 256 + ((((col >> 5) ^ (row & 3))) << 6) + ((col & 31) << 1));
}
struct TgUnit { int seq, j, u, tend; };                  // K-tile j of sequence seq (tend = its valid rows); u = seq * tps + j (u >= uend: none left)

// lens[seq] through the SCALAR cache.  Written as a plain load hipcc emits global_load_dword + s_waitcnt vmcnt(0) for it (the
// pointer sits in a by-value argument struct: no noalias / readonly facts) - and that wait drains the DMA ring several times per
// K-tile.  An s_load counts on lgkmcnt and leaves the vector-memory queue alone.
__device__ __forceinline__ int tg_sload(const int32_t* p) {
    int v;
    asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p) : "memory");
    return v;
}

template <int NT, int CB = 1>
__device__ __forceinline__ void wgrad_tg_body(const WgradArgs& a, unsigned char* smem, int tile_n, int tile_c, int tap0, int split) {
    static_assert(NT >= 1 && NT <= 5, "tap group of 1 .. 5 taps");
    static_assert(CB == 1 || (CB == 2 && NT == 1), "two X images only for one-tap launches");
    constexpr int XI = TgCfg<NT, CB>::X_BYTES;
    constexpr int NR = (8 + NT - 1 + 3) / 4;             // transposing reads per X run (4 rows each): 2 (NT = 1) or 3
    constexpr int TG_X_ROWS = TgCfg<NT, CB>::X_ROWS, TG_BUF_BYTES = TgCfg<NT, CB>::BUF_BYTES, NBUF = TgCfg<NT, CB>::NBUF, D = NBUF - 1;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);           // 8 waves: wm = n half, wn = 32-column block of c
    const int wm = wave >> 2, wn = wave & 3;
    const int n0 = tile_n * 128, c0 = tile_c * 128 * CB;
    const int shift0 = tap0 - a.pad;                     // tile row r of the X image = sequence row t0 + shift0 + r (dil == 1)
    const int tps = (a.S + 63) >> 6;                     // K-tiles per sequence
    const int nunits = (a.M / a.S) * tps;
    const int ubeg = split * a.rows_per_split;           // units per split
    const int uend = min(nunits, ubeg + a.rows_per_split);
    const unsigned smem_u = lds_addr(smem);

    // ---- K-tile list: live units of [ubeg, uend) in order (a unit is dead when its first row is at or beyond lens[seq])
    auto tend_of = [&](int seq) -> int { return a.lens ? min(tg_sload(a.lens + seq), a.S) : a.S; };
    auto settle = [&](TgUnit q) -> TgUnit {              // first live unit at or after q (q.tend is valid while q.u < uend)
        while (q.u < uend && q.j * 64 >= q.tend) {
            q.seq += 1; q.j = 0; q.u = q.seq * tps;
            if (q.u < uend) q.tend = tend_of(q.seq);
        }
        if (q.u >= uend) q.u = uend;
        return q;
    };
    auto next_of = [&](TgUnit q) -> TgUnit {
        if (q.u >= uend) return q;
        q.j += 1; q.u += 1;
        if (q.j >= tps) { q.seq += 1; q.j = 0; if (q.u < uend) q.tend = tend_of(q.seq); }
        return settle(q);
    };

    // ---- DMA geometry.  A 1 KiB piece = 4 tile rows; lane l writes LDS position (row l >> 4, 16-byte chunk l & 15), which must
    // hold global chunk cg = swizzle^-1: the 64-byte group index XOR (row & 3) (row & 3 == l >> 4 for every piece).
    // Wave w issues dY pieces w, w + 8 (rows 4w.., 4w + 32..), X pieces w, w + 8 and - waves 0, 1 - X piece 16 + w (rows 64 + 4w..).
    const int lrow = lane >> 4, lq = lane & 15;
    const int cg = (((lq >> 2) ^ lrow) << 2) | (lq & 3);
    const unsigned acol_b = (unsigned)min(n0 + cg * 8, a.N - 8) * 2u;    // clamped into the matrix (tails feed outputs nobody stores)
    const unsigned xcol_b = (unsigned)min(c0 + cg * 8, a.Cin - 8) * 2u;
    const unsigned xcol_b2 = (unsigned)min(c0 + 128 + cg * 8, a.Cin - 8) * 2u;        // (CB == 2: the second 128-column image)
    const unsigned dy_rs = (unsigned)a.lddy * 2u, x_rs = (unsigned)a.ldx * 2u;
    const bool w5 = TG_X_ROWS > 64 && wave < 2;         // (NT == 1: no halo pieces, 4 pieces per wave)
    auto issue = [&](const TgUnit& q, int buf) {
        const int t0 = q.j * 64;
        const unsigned base = (unsigned)(q.seq * a.S);
        const unsigned dst = __builtin_amdgcn_readfirstlane(smem_u + (unsigned)(buf * TG_BUF_BYTES + wave * 1024));   // (uniform: M0)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int t = min(t0 + 4 * wave + 32 * i + lrow, a.S - 1);
            glds16_sbase((base + (unsigned)t) * dy_rs + acol_b, a.dY, dst + (unsigned)(i * 8192));
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int t = min(max(t0 + shift0 + 4 * wave + 32 * i + lrow, 0), a.S - 1);
            glds16_sbase((base + (unsigned)t) * x_rs + xcol_b, a.X, dst + (unsigned)(TG_A_BYTES + i * 8192));
            if (CB == 2) glds16_sbase((base + (unsigned)t) * x_rs + xcol_b2, a.X, dst + (unsigned)(TG_A_BYTES + XI + i * 8192));
        }
        if (w5) {
            const int t = min(max(t0 + shift0 + 64 + 4 * wave + lrow, 0), a.S - 1);
            glds16_sbase((base + (unsigned)t) * x_rs + xcol_b, a.X, dst + (unsigned)(TG_A_BYTES + 16384));
        }
    };

    // ---- fragment addressing (per lane, computed once): lane = (li, g): 16-lane group g supplies the [4 rows][16 cols] block of a
    // transposing read; h = g >> 1 is the MFMA k-group (rows 8h .. 8h + 7 of a 16-row sub-step)
    const int li = lane & 15, g = lane >> 4, h = g >> 1;
    const int rrow = 8 * h + (li >> 2);
    const int acol = wm * 64 + 16 * (g & 1) + 4 * (li & 3);
    const int bcol = wn * 32 + 16 * (g & 1) + 4 * (li & 3);
    unsigned offA[2][2];                                 // [32-column block of n][rows 0-3 / 4-7 of the lane's 8]
#pragma unroll
    for (int blk = 0; blk < 2; ++blk)
#pragma unroll
        for (int hl = 0; hl < 2; ++hl) offA[blk][hl] = tg_off(rrow + 4 * hl, acol + blk * 32);
    const unsigned offB = TG_A_BYTES + tg_off(rrow, bcol);   // first row of the lane's run; reads q add 4 q rows = 1024 q bytes
    typedef __attribute__((address_space(3))) tg_s16x4* lds_s4;
    auto tr_read = [&](unsigned off) -> tg_s16x4 { return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(size_t)(smem_u + off)); };

    f32x16 acc[NT * CB][2];                              // (CB == 2, one tap: [X image][n block])
#pragma unroll
    for (int t = 0; t < NT * CB; ++t)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][i][r] = 0.f;
    const bool do_bias = a.dbias != nullptr && tile_c == 0 && tap0 == 0 && wn == 0;      // wave-uniform
    float bsum[2] = {0.f, 0.f};

    tg_u32x4 af[2][2];                                   // [set][n block]: fragment double buffer over the 16-row sub-steps
    tg_u32x2 run[2][CB][NR];                             // [set][X image][read]: rows 8h + 4q .. + 3 of the lane's X column as two row-pair dwords
    // one K-tile out of LDS buffer BUF (compile time: every ds_read offset is an immediate)
    auto ktile = [&](auto bufc) {
        constexpr int BUF = decltype(bufc)::value;
        constexpr unsigned bufoff = BUF * TG_BUF_BYTES;
        auto read_frags = [&](int set, int ks) {
#pragma unroll
            for (int blk = 0; blk < 2; ++blk) {
                const tg_s16x4 lo = tr_read(offA[blk][0] + bufoff + ks * 4096);
                const tg_s16x4 hi = tr_read(offA[blk][1] + bufoff + ks * 4096);
                af[set][blk] = __builtin_bit_cast(tg_u32x4, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
            }
#pragma unroll
            for (int cb = 0; cb < CB; ++cb)
#pragma unroll
                for (int r = 0; r < NR; ++r) run[set][cb][r] = __builtin_bit_cast(tg_u32x2, tr_read(offB + bufoff + cb * XI + ks * 4096 + r * 1024));
        };
        auto mma = [&](int set) {
            const tg_u32x4 av[2] = {af[set][0], af[set][1]};
            unsigned R[2 * NR];
#pragma unroll
            for (int r = 0; r < NR; ++r) { R[2 * r] = run[set][0][r][0]; R[2 * r + 1] = run[set][0][r][1]; }
            if (do_bias) {
                // (each dword goes through a named scalar: written as av[blk][d] inside the bit_cast, hipcc 7.2 fed element 0 to all
                // four v_dot2c - seen in the ISA and as an O(1)-wrong bias gradient on the first GPU run)
#pragma unroll
                for (int blk = 0; blk < 2; ++blk) {
                    const unsigned e0 = av[blk][0], e1 = av[blk][1], e2 = av[blk][2], e3 = av[blk][3];
                    float t = bsum[blk];
                    t = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(tg_bf16x2, e0), __builtin_bit_cast(tg_bf16x2, 0x3f803f80u), t, false);
                    t = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(tg_bf16x2, e1), __builtin_bit_cast(tg_bf16x2, 0x3f803f80u), t, false);
                    t = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(tg_bf16x2, e2), __builtin_bit_cast(tg_bf16x2, 0x3f803f80u), t, false);
                    t = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(tg_bf16x2, e3), __builtin_bit_cast(tg_bf16x2, 0x3f803f80u), t, false);
                    bsum[blk] = t;
                }
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                tg_u32x4 b;
                if (t & 1) {
#pragma unroll
                    for (int d = 0; d < 4; ++d) b[d] = __builtin_amdgcn_alignbit(R[(t >> 1) + d + 1], R[(t >> 1) + d], 16);
                } else {
#pragma unroll
                    for (int d = 0; d < 4; ++d) b[d] = R[(t >> 1) + d];
                }
#pragma unroll
                for (int mb = 0; mb < 2; ++mb)
                    acc[t][mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av[mb]), __builtin_bit_cast(bf16x8, b),
                                                                         acc[t][mb], 0, 0, 0);
            }
            if constexpr (CB == 2) {                     // the second X image against the same dY fragments
                tg_u32x4 b2;
                b2[0] = run[set][1][0][0]; b2[1] = run[set][1][0][1]; b2[2] = run[set][1][1][0]; b2[3] = run[set][1][1][1];
#pragma unroll
                for (int mb = 0; mb < 2; ++mb)
                    acc[1][mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av[mb]), __builtin_bit_cast(bf16x8, b2),
                                                                         acc[1][mb], 0, 0, 0);
            }
        };
        read_frags(0, 0);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            if (ks < 3) read_frags((ks + 1) & 1, ks + 1);
            mma(ks & 1);
        }
    };

    // ---- pipeline: tile i lives in buffer i % NBUF; its DMA is issued D = NBUF - 1 tiles ahead.  Per tile: wait for MY pieces of it
    // (the later tiles' stay in flight), ONE barrier (everybody's pieces landed; everybody has left the buffer the next issue overwrites)
    TgUnit q[D];                                          // q[0] = the tile being multiplied next, q[1 ..] = issued ahead of it
    q[0] = TgUnit{ubeg / tps, ubeg % tps, ubeg, 0};
    if (q[0].u < uend) q[0].tend = tend_of(q[0].seq);
    q[0] = settle(q[0]);
#pragma unroll
    for (int i = 1; i < D; ++i) q[i] = next_of(q[i - 1]);
#pragma unroll
    for (int i = 0; i < D; ++i)
        if (q[i].u < uend) issue(q[i], i);
    // rows of tile u (in buffer buf) that lie outside its sequence -> zeros, by the wave that fetched them (its pieces have landed)
    auto zero_rows = [&](const TgUnit& u, int buf) {
        const int t0 = u.j * 64, tend = u.tend;
        const int xdead = tend - t0 + NT - 1;             // first X tile row all of whose partner dY rows are >= tend
        unsigned char* bp = smem + buf * TG_BUF_BYTES + wave * 1024 + lane * 16;
        const uint4 z = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (t0 + 4 * wave + 32 * i + lrow >= tend) *reinterpret_cast<uint4*>(bp + i * 8192) = z;
            // X tile row r pairs with dY rows r - t, t < NT: outside [0, S) it is not this sequence's data; at r >= xdead every
            // partner row is beyond the valid part (a zero) - and 0 x whatever the producer left in a padded row (attention
            // output, skipped tiles: possibly NaN bit patterns) must stay 0, so those rows are zeroed too
            const int rx = 4 * wave + 32 * i + lrow, tx = t0 + shift0 + rx;
            if (tx < 0 || tx >= a.S || rx >= xdead) {
                *reinterpret_cast<uint4*>(bp + TG_A_BYTES + i * 8192) = z;
                if (CB == 2) *reinterpret_cast<uint4*>(bp + TG_A_BYTES + XI + i * 8192) = z;
            }
        }
        if (w5) {
            const int rx = 64 + 4 * wave + lrow, tx = t0 + shift0 + rx;
            if (tx < 0 || tx >= a.S || rx >= xdead) *reinterpret_cast<uint4*>(bp + TG_A_BYTES + 16384) = z;
        }
    };
    auto step = [&](auto bufc) {
        constexpr int BUF = decltype(bufc)::value;
        const TgUnit nxt = next_of(q[D - 1]);
        int ahead = 0;                                    // tiles issued after q[0] whose pieces may stay in flight
#pragma unroll
        for (int i = 1; i < D; ++i) ahead += (q[i].u < uend) ? 1 : 0;
        constexpr int P = 2 + 2 * CB;                     // DMA pieces per wave and tile (+ 1 halo piece on waves 0, 1 of a multi-tap tile)
        if (ahead == 0) tg_wait_vm<0>();
        else if (ahead == 1) { if (w5) tg_wait_vm<P + 1>(); else tg_wait_vm<P>(); }
        else if (ahead == 2) { if (w5) tg_wait_vm<2 * P + 2>(); else tg_wait_vm<2 * P>(); }
        else { if (w5) tg_wait_vm<3 * P + 3>(); else tg_wait_vm<3 * P>(); }
        const int t0 = q[0].j * 64;
        if ((t0 + 64 > q[0].tend) || (t0 + shift0 < 0) || (t0 + shift0 + TG_X_ROWS > a.S)) zero_rows(q[0], BUF);   // (block-uniform)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (nxt.u < uend) issue(nxt, (BUF + D) % NBUF);
        ktile(bufc);
#pragma unroll
        for (int i = 0; i + 1 < D; ++i) q[i] = q[i + 1];
        q[D - 1] = nxt;
    };
    static_assert(NBUF == 3 || NBUF == 4, "ring of 3 or 4 buffers");
    while (q[0].u < uend) {
        step(std::integral_constant<int, 0>{});
        if (q[0].u >= uend) break;
        step(std::integral_constant<int, 1>{});
        if (q[0].u >= uend) break;
        step(std::integral_constant<int, 2>{});
        if (NBUF == 4) {
            if (q[0].u >= uend) break;
            step(std::integral_constant<int, (NBUF == 4 ? 3 : 0)>{});
        }
    }

    // ---- epilogue: acc[t][mb][r]: row n = wm*64 + mb*32 + (r&3) + 8*(r>>2) + 4*fh, column c = wn*32 + fl
    const int fl = lane & 31, fh = lane >> 5;
    float* const outW = a.slab ? a.slab + (size_t)split * a.slab_stride : a.dW;
    if (do_bias) {
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
            const float tot = bsum[blk] + __shfl_xor(bsum[blk], 32, 64);          // the two k-groups of a column
            const int n = n0 + wm * 64 + blk * 32 + fl;
            if (fh == 0 && n < a.N) {
                if (a.slab) outW[(size_t)a.N * a.taps * a.Cin + n] = tot;
                else atomicAdd(a.dbias + n, tot);
            }
        }
    }
    const int c = c0 + wn * 32 + fl;
    if (c < a.Cin) {
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + wm * 64 + mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                if (n >= a.N) continue;
                float* dst = outW + ((size_t)n * a.taps + tap0) * a.Cin + c;
                if (a.slab) {
#pragma unroll
                    for (int t = 0; t < NT; ++t) dst[(size_t)t * a.Cin] = acc[t][mb][r];
                } else {
#pragma unroll
                    for (int t = 0; t < NT; ++t) atomicAdd(dst + (size_t)t * a.Cin, acc[t][mb][r]);
                }
                if constexpr (CB == 2) {                  // the second X image: columns c + 128 (one tap)
                    if (c + 128 < a.Cin) {
                        if (a.slab) dst[128] = acc[1][mb][r];
                        else atomicAdd(dst + 128, acc[1][mb][r]);
                    }
                }
            }
    }
}

// A (K-split, tile) pair per workgroup; tile = (n-tile, c-tile, tap group): the first a.g3 groups take NF taps each, one more group
// (if NR) the remaining NR taps.  One 512-thread workgroup per CU (102 KiB of LDS, <= 256 registers per wave: two waves per SIMD).
// XCD-AWARE PLACEMENT (round 6).  The tiles of ONE split read the same rows of dY and X - each dY column block ntc x groups times,
// each X column block ntn x groups times.  Rounds 3-5 launched a (tiles, splits) grid: the dispatcher deals consecutive
// workgroups round-robin to the 8 XCDs, so the tiles of a split landed on 8 different L2s and every one of them fetched its
// operands from the Infinity Cache / HBM again (PMC r05zzz: one-tap launches 104 MB fetched for 30-45 MB of operands, 5 + 4 taps
// 229 MB for 114 MB).  Now the grid is 1-D and padded to 8 x per_xcd: workgroup `lin` sits on XCD lin & 7 (the dispatcher's
// rule) and takes pair p = (lin & 7) * per_xcd + (lin >> 3) of the split-major pair list - every XCD gets the same number of
// pairs, a CONTIGUOUS range of them (the tiles of one split, or of two neighbouring splits), and they start together, so the
// second reader of a row block hits in its XCD's L2.
template <int NF, int NR>
__global__ void __launch_bounds__(512, 2) conv_wgrad_tg_kernel(WgradArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // NBUF x [dY 64 rows | X 64 / 72 rows][256 B]
    const int ntn = (a.N + 127) >> 7, ntc = (a.Cin + 127) >> 7;
    const int kx = blockIdx.x >> 3;
    const int pair = (int)(blockIdx.x & 7) * a.per_xcd + kx;
    if (pair >= a.n_tiles * a.n_splits) return;              // padding of the last XCD's range (no barrier has been executed)
    const int split = pair / a.n_tiles;
    int bx = pair - split * a.n_tiles;
    const int tile_n = bx % ntn; bx /= ntn;
    const int tile_c = bx % ntc; bx /= ntc;
    if (NR == 0 || bx < a.g3) wgrad_tg_body<NF>(a, smem, tile_n, tile_c, bx * NF, split);
    else wgrad_tg_body<(NR ? NR : 1)>(a, smem, tile_n, tile_c, a.g3 * NF, split);
}

// one tap, 128 (n) x 256 (c) workgroup tiles (TgCfg<1, 2>: three 48 KiB buffers)
__global__ void __launch_bounds__(512, 2) conv_wgrad_tg1w_kernel(WgradArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int ntn = (a.N + 127) >> 7, ntc = (a.Cin + 255) >> 8;
    const int kx = blockIdx.x >> 3;
    const int pair = (int)(blockIdx.x & 7) * a.per_xcd + kx;
    if (pair >= a.n_tiles * a.n_splits) return;
    const int split = pair / a.n_tiles;
    int bx = pair - split * a.n_tiles;
    const int tile_n = bx % ntn; bx /= ntn;
    const int tile_c = bx % ntc;
    wgrad_tg_body<1, 2>(a, smem, tile_n, tile_c, 0, split);
}
static void launch_tg1w(WgradArgs a, const WgradPlan& p, hipStream_t stream) {
    constexpr int dyn = TgCfg<1, 2>::NBUF * TgCfg<1, 2>::BUF_BYTES;
    static Fs2DevOnce once;
    once.run([&] { (void)hipFuncSetAttribute((const void*)conv_wgrad_tg1w_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, dyn); });
    a.g3 = p.g_first;
    a.rows_per_split = p.ups;
    a.n_tiles = p.tiles; a.n_splits = p.splits;
    a.per_xcd = (p.tiles * p.splits + 7) / 8;
    conv_wgrad_tg1w_kernel<<<dim3((unsigned)(8 * a.per_xcd)), 512, dyn, stream>>>(a);
}

template <int NF, int NR>
static void launch_tg(WgradArgs a, const WgradPlan& p, hipStream_t stream) {
    constexpr int dyn = (TgCfg<NF>::NBUF * TgCfg<NF>::BUF_BYTES > TgCfg<(NR ? NR : NF)>::NBUF * TgCfg<(NR ? NR : NF)>::BUF_BYTES)
                            ? TgCfg<NF>::NBUF * TgCfg<NF>::BUF_BYTES : TgCfg<(NR ? NR : NF)>::NBUF * TgCfg<(NR ? NR : NF)>::BUF_BYTES;
    static Fs2DevOnce once;
    once.run([&] { (void)hipFuncSetAttribute((const void*)conv_wgrad_tg_kernel<NF, NR>, hipFuncAttributeMaxDynamicSharedMemorySize, dyn); });
    a.g3 = p.g_first;
    a.rows_per_split = p.ups;
    a.n_tiles = p.tiles; a.n_splits = p.splits;
    a.per_xcd = (p.tiles * p.splits + 7) / 8;
    conv_wgrad_tg_kernel<NF, NR><<<dim3((unsigned)(8 * a.per_xcd)), 512, dyn, stream>>>(a);
}

bool fs2_wgrad_tg_launch(WgradArgs a, const WgradPlan& p, hipStream_t stream) {
    const int nf = p.n_first, nr = p.n_rest;
    if (p.share == 2 && nf == 1 && nr == 0) { launch_tg1w(a, p, stream); return true; }     // (the caller counted 128 x 256 tiles)
    if (nf == 5 && nr == 0) launch_tg<5, 0>(a, p, stream);
    else if (nf == 5 && nr == 4) launch_tg<5, 4>(a, p, stream);
    else if (nf == 4 && nr == 0) launch_tg<4, 0>(a, p, stream);
    else if (nf == 4 && nr == 3) launch_tg<4, 3>(a, p, stream);
    else if (nf == 3 && nr == 0) launch_tg<3, 0>(a, p, stream);
    else if (nf == 2 && nr == 0) launch_tg<2, 0>(a, p, stream);
    else if (nf == 1 && nr == 0) launch_tg<1, 0>(a, p, stream);
    else return false;
    return true;
}
