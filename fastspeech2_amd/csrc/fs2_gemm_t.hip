// fs2_gemm_t.hip - TALL-tile, all-consumer persistent implicit GEMM for the large bf16 convolutions (gfx950, round 4).
//
// Same contraction, operands and epilogue as conv_gemm_p_kernel<false> (fs2_gemm_p.hip; reference call sites
// transformer/SubLayers.py:87-88 k = 9 FFN conv, transformer/Layers.py:129-137 PostNet k = 5), re-cut after round 3's profile of
// that kernel (0.40 of the MFMA peak, MFMA busy 46 %, wave time 30 active / 26 issue-stalled / 44 parked, L2 ->
// LDS traffic 991 MB per k = 9 launch = 13 x the algorithmic bytes):
//   * the 256 x 128 kernel is co-bound by its OWN operand stream.  Per output tile it streams the whole weight slice of its
//     N-tile: L2 -> LDS bytes per MFLOP ~ 1 / (taps N_t) + 1 / M_t, and at M_t = 256 the weight term is 4/5 of it; its r02 ablation
//     says the same - loaders + barriers alone 97 us of a 215 us launch whose MFMA floor is 105 us.  Here M_t = 512: the weight
//     stream is shared by twice the rows (k = 9: -41 % operand bytes per FLOP, DMA floor 97 -> 57 us).
//   * one consumer wave per SIMD issues reads and MFMAs in order and nothing runs while it waits.  Here EIGHT waves, two per
//     SIMD, each owning 64 (M) x 128 (N) of the 512 x 128 tile - same fragment traffic per MFMA (6 ds_read_b128 per 8 MFMAs,
//     37 % of the LDS pipe at gfx950's 4 cycles per ds_read_b128) - and every wave issues its own share of the LDS-DMA (the
//     structure of fs2_wgrad.hip): no loader waves, 256 registers per wave.
//   * LDS: a 528-row halo tile at 64 channels would be 67.6 KB per buffer and could not be double-buffered next to a weight ring,
//     so the K loop walks HALF chunks of 32 channels: activation halo tile [528][32] (33 KB) x 2 buffers (buffer = half h), weight
//     ring of TAPS [128][32] slots (slot = tap; 8 KB each: exactly one 1 KiB DMA piece per wave and K-step).  Rows are 64 bytes;
//     the four 16-byte chunks of a row are XOR-swizzled with (row >> 2) & 3, which puts the 16 lanes of every ds_read_b128 lane
//     group on 16 different 16-byte bank groups (measured: SQ_LDS_BANK_CONFLICT = 0, profiles/r04h_tall_pmc_mfma.md).
//   * a K-step = (half chunk, tap): 2 k-slices of 16, 16 MFMAs per wave, ONE raw barrier.
//   * the first version of this file kept run-time positions, issue stamps and a computed vmcnt for every step: correct, and
//     slow (250 us on the k = 9 FFN forward, 197 for fs2_gemm_p.hip): ~300 instructions per step for 16 MFMAs, 67 spilled SGPRs
//     (r04h: MFMA busy 31.7 %, 3.45 VALU per MFMA).  This version makes the schedule EXACTLY periodic and unrolls it over one
//     64-channel chunk (2 halves x TAPS steps): LDS offsets are immediates, the refill of a step is "the same tap of the next half
//     chunk into the slot just read" (one DMA piece per wave), the halo tile of the half chunk after the next is issued at two
//     fixed taps, every wave issues the same number of operations per step (waves without a fifth halo piece re-issue their
//     fourth; past the end of the work the refills re-fetch the last chunk into free slots) - so every vmcnt is a compile-time
//     constant, the only run-time bookkeeping happens once per chunk, and a step is 58 instructions.
// WHAT IT MEASURES (round 4, same-box A/B against fs2_gemm_p.hip, profiles/r04f..r04m_*): parity green in all three forms below, and
// NOT faster at the bench shapes - the default dispatch never picks it (fs2_conv_gemm_t_ok); it stays reachable through
// fs2_conv_gemm_tall for the parity test and tools/bench_conv.py.
//   form A (this file): 8 waves, reads interleaved with MFMAs ............ k = 9 FFN forward 205 us, PostNet k = 5 122 us
//   form B (git history): the same 8 waves in compute / load ping-pong ... 213 / 132 us
//   form C (git history): 4 waves x 128 x 128, 512 registers each ........ 234 / 134 us
//   fs2_gemm_p.hip (256 x 128, 4 loader + 4 consumer waves) .............. 197 / 105 us
// The compile-time ablations of all three forms agree on where the time goes, whatever the structure (k = 9 forward, 3 rounds of
// tiles, MFMA floor 93 us at the nominal 2.4 GHz): MFMAs + barriers alone 128-133 us - the chip clocks to ~1.95 GHz under a
// back-to-back MFMA stream (DVFS), i.e. 0.65 of the nominal peak is the ceiling of ANY kernel here; + fragment reads +34..36 us
// (also with every lgkmcnt wait removed: not latency - the reads' issue / register write-back take MFMA time, 13-20 cycles per
// ds_read_b128) -> 0.51 of nominal with free operands and a free epilogue; + DMA issue +16..21 us; + epilogue +25..47 us (store-
// issue-bound, and 3 rounds of 512-row tiles for 2.72 rounds of work).  fs2_gemm_p.hip's 197 us sits at 0.83 of that 164 us
// read-fed ceiling; the 0.50-of-nominal target of VERDICT r03 is above what an LDS-fed bf16 MFMA loop reaches on this chip.
#include "fs2_gemm.h"
#include "fs2_sched.h"
#include "fs2_gemm_epi.h"

static constexpr int T_TM = 512;
static constexpr int T_AROWS = 528;
static constexpr int T_A_BYTES = T_AROWS * 64;            // one half-chunk halo tile
static constexpr int T_B_BYTES = 128 * 64;                // one weight slot
static constexpr int T_B_OFF = 2 * T_A_BYTES;
template <int TAPS> struct TCfg {
    static constexpr int AUX = T_B_OFF + TAPS * T_B_BYTES;    // 8 bias lines of 128 floats behind the rings
    static constexpr int LDS = AUX + 8 * 512;
};

#define FS2T_WAIT_LGKM(N) asm volatile("s_waitcnt lgkmcnt(" #N ")" ::: "memory")
#define FS2T_FENCE() __builtin_amdgcn_sched_barrier(0)
template <int N> __device__ __forceinline__ void t_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }
// LDS-DMA of one 1 KiB piece: per-lane byte offset voff from the (scalar) base, lane-linear destination at lds_dst (M0)
__device__ __forceinline__ void t_glds(unsigned voff, const unsigned char* sbase, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}

// the workgroup's unit list in two VGPRs (lane k = k-th unit), as fs2_gemm_p.hip's PUnits, without a tile map
struct TUnits { unsigned v0, v1; };
__device__ __forceinline__ void t_units_load(const PSched& s, int lane, TUnits& t) {
    int mi, nt, kc0, nkc, np;
    unsigned a0 = 0, a1 = 0;
    if (p_unit(s, lane, mi, nt, kc0, nkc, np)) {
        a0 = (unsigned)mi | ((unsigned)nt << 20) | ((unsigned)np << 28);
        a1 = (unsigned)kc0 | ((unsigned)nkc << 16);
    }
    t.v0 = a0; t.v1 = a1;
}
__device__ __forceinline__ void t_tile_of(int k, const TUnits& t, int& mt, int& nt, int& kc0, int& nparts) {
    const unsigned a0 = __builtin_amdgcn_readlane(t.v0, k), a1 = __builtin_amdgcn_readlane(t.v1, k);
    mt = (int)(a0 & 0xfffffu);
    nt = (int)((a0 >> 20) & 0xffu);
    nparts = (int)(a0 >> 28);
    kc0 = __builtin_amdgcn_readfirstlane((int)(a1 & 0xffffu));
}
__device__ __forceinline__ int t_ntiles(const PSched& s) {
    const PPlan p = p_plan(s, s.b);
    return p.R + (p.j < p.tail * p.tks ? 1 : 0);
}
__device__ __forceinline__ int t_last_nkc(const PSched& s) {
    const PPlan p = p_plan(s, s.b);
    return (p.j < p.tail * p.tks) ? s.nkc_u / p.tks : s.nkc_u;
}

// tail part: the partial 512 x 128 tile -> the workgroup's f32 slab, tile-local row-major
__device__ __forceinline__ void t_epilogue_part(float* slab, f32x16 (&acc)[2][4], int wm, int fl, int fh) {
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
        float* wrow = slab + (size_t)(wm * 64 + mb * 32 + fl) * 128;
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
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
                const int n = nb * 32 + ch * 16 + fh * 8;
                *reinterpret_cast<float4*>(wrow + n) = make_float4(c[ch][0], c[ch][1], c[ch][2], c[ch][3]);
                *reinterpret_cast<float4*>(wrow + n + 4) = make_float4(c[ch][4], c[ch][5], c[ch][6], c[ch][7]);
            }
        }
    }
}


// ABL (dev builds): 1 = no DMA, 2 = no MFMA, 4 = no fragment reads, 8 = no per-step barrier (only with 1), 16 = no epilogue
template <int TAPS, int ABL>
__global__ void __launch_bounds__(512) conv_gemm_t_kernel(ConvGemmArgs a, PSched sc0) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef TCfg<TAPS> C;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);           // = wm: rows 64 wave .. of the tile
    PSched sc = sc0;
    sc.b = blockIdx.x;
    const int ntiles = t_ntiles(sc);
    if (ntiles == 0) return;
    TUnits units;
    t_units_load(sc, lane, units);
    float* bias_s = reinterpret_cast<float*>(smem + C::AUX) + wave * 128;
    const int nkc = sc.nkc_u, nkc_last = t_last_nkc(sc);
    const unsigned smem_u = lds_addr(smem);
    const unsigned char* Xb = reinterpret_cast<const unsigned char*>(a.X);
    const unsigned char* Wb = reinterpret_cast<const unsigned char*>(a.W);

    // ------------------------------------------------------------------ DMA side (every wave: its own 1 KiB pieces)
    // a piece = 16 rows x 64 bytes; lane l writes LDS position (row l >> 2, 16-byte chunk l & 3), which must hold the SOURCE chunk
    // (l & 3) ^ key(row), key(row) = (row >> 2) & 3 = (l >> 4) & 3 for every piece (pieces start at multiples of 16 rows).
    const int prow = lane >> 2;
    const unsigned pchunk = (unsigned)(((lane & 3) ^ ((lane >> 4) & 3)) << 4);
    const unsigned ldx2 = (unsigned)(a.ldx * 2), ldw2 = (unsigned)(a.ldw * 2), cin2 = (unsigned)(a.Cin * 2);
    const unsigned adst = smem_u + (unsigned)(wave * 1024);              // + buffer * A_BYTES + i * 8192: piece i of this wave
    const unsigned bdst = smem_u + (unsigned)(T_B_OFF + wave * 1024);    // + slot * B_BYTES
    // the chunk being multiplied (cur) and the one after it (nxt: the next 64 channels of the unit, the first chunk of the
    // workgroup's next unit, or - past the end - the current one again)
    const unsigned char *xcur, *wcur, *xnxt, *wnxt;
    int m0cur, m0nxt, nk = 0, nc = 0, nkc_n;
    unsigned offB, offBn;
    auto unit_origin = [&](int k, int& m0, const unsigned char*& xp, const unsigned char*& wp, unsigned& ob) {
        int mt, nt, kc0, np;
        t_tile_of(k, units, mt, nt, kc0, np);
        m0 = mt * T_TM;
        xp = Xb + (size_t)kc0 * 128;
        wp = Wb + (size_t)kc0 * 128;
        ob = (unsigned)min(nt * 128 + wave * 16 + prow, a.N - 1) * ldw2 + pchunk;
    };
    unit_origin(0, m0cur, xcur, wcur, offB);
    nkc_n = (ntiles == 1) ? nkc_last : nkc;
    xnxt = xcur; wnxt = wcur; m0nxt = m0cur; offBn = offB;
#define T_ADVANCE_NEXT()                                                                                                    \
    do {                                                                                                                    \
        if (nc + 1 < nkc_n) { ++nc; xnxt += 128; wnxt += 128; }                                                             \
        else if (nk + 1 < ntiles) { ++nk; nc = 0; nkc_n = (nk == ntiles - 1) ? nkc_last : nkc; unit_origin(nk, m0nxt, xnxt, wnxt, offBn); } \
    } while (0)
    T_ADVANCE_NEXT();
    // piece I (0 .. 4) of the halo tile whose rows start at tile origin M0_, channels at XP: tile rows 16 (wave + 8 I) .. ; piece 4
    // exists for wave 0 only (rows 512 .. 527) - the other waves re-issue their piece 3 so that every wave issues the same count
#define T_ISSUE_A(M0_, XP, BUF, I)                                                                                          \
    do {                                                                                                                    \
        if (!(ABL & 1)) {                                                                                                   \
            const int i_ = ((I) == 4 && wave != 0) ? 3 : (I);                                                               \
            const int g_ = min(max((M0_) - a.pad + (wave + 8 * i_) * 16 + prow, 0), a.M - 1);                               \
            t_glds((unsigned)g_ * ldx2 + pchunk, (XP), adst + (unsigned)((BUF) * T_A_BYTES + i_ * 8192));                   \
        }                                                                                                                   \
    } while (0)
#define T_ISSUE_B(WP, OFFB, TAP)                                                                                            \
    do { if (!(ABL & 1)) t_glds((OFFB), (WP) + (size_t)(TAP) * cin2, bdst + (unsigned)((TAP) * T_B_BYTES)); } while (0)

    // prologue: both halo tiles of the first chunk, the weight slots of its first half
#pragma unroll
    for (int i = 0; i < 5; ++i) T_ISSUE_A(m0cur, xcur, 0, i);
#pragma unroll
    for (int i = 0; i < 5; ++i) T_ISSUE_A(m0cur, xcur + 64, 1, i);
#pragma unroll
    for (int t = 0; t < TAPS; ++t) T_ISSUE_B(wcur, offB, t);

    // ------------------------------------------------------------------ MFMA side
    const int fl = lane & 31, fh = lane >> 5;
    // per-lane LDS addresses.  A: row base + (((2 s + fh) ^ key(tap)) << 4), key(tap) = ((fl + tap) >> 2) & 3 (dil = 1): the keys of
    // all taps packed two bits each; buffer, tap shift and the second 32-row block are immediates.  B: the key never changes.
    const unsigned alane = smem_u + (unsigned)((wave * 64 + fl) * 64);
    unsigned kpack = 0;
#pragma unroll
    for (int t = 0; t < TAPS; ++t) kpack |= (unsigned)(((fl + t) >> 2) & 3) << (2 * t);
    const unsigned bkey = (unsigned)((fl >> 2) & 3);
    unsigned blane0 = smem_u + T_B_OFF + (unsigned)(fl * 64) + ((((unsigned)fh) ^ bkey) << 4);          // k-slice 0
    unsigned blane1 = smem_u + T_B_OFF + (unsigned)(fl * 64) + ((((unsigned)(2 + fh)) ^ bkey) << 4);    // k-slice 1
#define T_A_ADDR(T_, S_) (alane + (((((kpack >> (2 * (T_))) & 3u) ^ (unsigned)(2 * (S_) + fh))) << 4))
#define T_B_ADDR(T_, S_) (((S_) ? blane1 : blane0) + ((T_) >= 4 ? 32768u : 0u))
#define T_A_OFF(H_, T_) ((H_) * T_A_BYTES + (T_) * 64)
#define T_B_OFFS(T_) (((T_) >= 4 ? (T_) - 4 : (T_)) * T_B_BYTES)

    u32x4 Af[2][2], Bf[2][4];
    f32x16 acc[2][4];
#define FS2T_DS_READ(dst, addr, OFF)                                                                                        \
    do {                                                                                                                    \
        if (!(ABL & 4)) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));                  \
        else asm volatile("" : "=v"(dst));                                                                                  \
    } while (0)
#define FS2T_MFMA(SET, MB, NB, AV)                                                                                          \
    do { if (!(ABL & 2)) acc[MB][NB] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, Bf[SET][NB]),     \
                                                                           __builtin_bit_cast(bf16x8, AV), acc[MB][NB], 0, 0, 0); } while (0)
    // one k-slice: 8 MFMAs on fragment set SET, the six reads of the next slice into set SET^1, one behind each of the first six
    // MFMAs; read order A0 B0 B1 B2 B3 A1, counted lgkmcnt (see fs2_gemm_p.hip FS2P_SLICE)
#define T_LGKM4() do { if (!(ABL & 32)) FS2T_WAIT_LGKM(4); } while (0)
#define FS2T_SLICE(SET, MASKED, LIVE0, LIVE1, AADDR, AOFF, BADDR, BOFF)                                                     \
    do {                                                                                                                    \
        u32x4 av0, av1;                                                                                                     \
        const unsigned aa_ = (AADDR), ba_ = (BADDR);                                                                        \
        T_LGKM4(); FS2T_FENCE();                                                                                            \
        av0 = Af[SET][0];                                                                                                   \
        if (MASKED && !(LIVE0)) av0 = u32x4{0u, 0u, 0u, 0u};                                                                \
        FS2T_MFMA(SET, 0, 0, av0); FS2T_FENCE();                                                                            \
        FS2T_DS_READ(Af[SET ^ 1][0], aa_, (AOFF)); FS2T_FENCE();                                                            \
        T_LGKM4(); FS2T_FENCE();                                                                                            \
        FS2T_MFMA(SET, 0, 1, av0); FS2T_FENCE();                                                                            \
        FS2T_DS_READ(Bf[SET ^ 1][0], ba_, (BOFF)); FS2T_FENCE();                                                            \
        T_LGKM4(); FS2T_FENCE();                                                                                            \
        FS2T_MFMA(SET, 0, 2, av0); FS2T_FENCE();                                                                            \
        FS2T_DS_READ(Bf[SET ^ 1][1], ba_, (BOFF) + 2048); FS2T_FENCE();                                                     \
        T_LGKM4(); FS2T_FENCE();                                                                                            \
        FS2T_MFMA(SET, 0, 3, av0); FS2T_FENCE();                                                                            \
        FS2T_DS_READ(Bf[SET ^ 1][2], ba_, (BOFF) + 4096); FS2T_FENCE();                                                     \
        T_LGKM4(); FS2T_FENCE();                                                                                            \
        av1 = Af[SET][1];                                                                                                   \
        if (MASKED && !(LIVE1)) av1 = u32x4{0u, 0u, 0u, 0u};                                                                \
        FS2T_MFMA(SET, 1, 0, av1); FS2T_FENCE();                                                                            \
        FS2T_DS_READ(Bf[SET ^ 1][3], ba_, (BOFF) + 6144); FS2T_FENCE();                                                     \
        FS2T_MFMA(SET, 1, 1, av1); FS2T_FENCE();                                                                            \
        FS2T_DS_READ(Af[SET ^ 1][1], aa_, (AOFF) + 2048); FS2T_FENCE();                                                     \
        FS2T_MFMA(SET, 1, 2, av1); FS2T_FENCE();                                                                            \
        FS2T_MFMA(SET, 1, 3, av1); FS2T_FENCE();                                                                            \
    } while (0)

    // vmcnt before the barrier that publishes step (half, tap NT): at most this many of the wave's operations may still be in
    // flight.  The weight piece of (half, NT) was issued TAPS steps earlier, first thing after that step's barrier; after it came the
    // halo pieces of that refill, then TAPS - 2 refills of one weight piece each, of which the ones at tap TAPS-1 / tap 0 add 3 / 2
    // halo pieces: TAPS - 2 + 5 - (halo pieces issued at tap NT - 1).  At NT = 0 the step also needs its halo tile, whose last two
    // pieces went out TAPS - 1 steps earlier, behind that refill's weight piece: TAPS - 2.
#define T_VMCNT(NT) ((NT) == 0 ? TAPS - 2 : ((NT) == 1 ? TAPS + 1 : TAPS + 3))
    // K-step (H, T) of the current chunk
#define T_STEP(H, T, MASKED)                                                                                                \
    do {                                                                                                                    \
        constexpr int NH_ = ((T) == TAPS - 1) ? ((H) ^ 1) : (H), NT_ = ((T) == TAPS - 1) ? 0 : (T) + 1;                     \
        /* (opaque re-definitions: the addresses and mask bits below are loop-invariant per tap, and hoisting 4 TAPS of them */ \
        /* out of the chunk loop cost 239 spilled VGPRs; recomputing them is 3 VALU per slice) */                             \
        asm volatile("" : "+v"(kpack), "+v"(vm0), "+v"(vm1), "+v"(blane0), "+v"(blane1));                                   \
        const bool live0 = (vm0 >> (T)) & 1u, live1 = (vm1 >> (T)) & 1u;                                                    \
        /* k-slice 0 (fragments prefetched by the previous step); reads of k-slice 1 of this step */                        \
        FS2T_SLICE(0, MASKED, live0, live1, T_A_ADDR(T, 1), T_A_OFF(H, T), T_B_ADDR(T, 1), T_B_OFFS(T));                    \
        if (!(ABL & 64)) FS2T_WAIT_LGKM(0);                                                                                 \
        FS2T_FENCE();                          /* every read of this step's slot / halo rows has landed */                  \
        t_wait_vm<T_VMCNT(NT_)>();             /* my pieces of the next step have landed */                                 \
        if (!(ABL & 8)) __builtin_amdgcn_s_barrier();   /* next step published; this step's slot (and, at tap TAPS-1, buffer H) released */ \
        /* refill: the same tap of the NEXT half chunk into the slot just released */                                       \
        if ((H) == 0) T_ISSUE_B(wcur + 64, offB, T); else T_ISSUE_B(wnxt, offBn, T);                                        \
        /* halo tiles: half H of the next chunk goes into buffer H once this half is done (3 pieces now, 2 one step later) */ \
        if ((T) == TAPS - 1) { T_ISSUE_A(m0nxt, xnxt + (H) * 64, H, 0); T_ISSUE_A(m0nxt, xnxt + (H) * 64, H, 1); T_ISSUE_A(m0nxt, xnxt + (H) * 64, H, 2); } \
        if ((T) == 0) {                                                                                                     \
            if ((H) == 0) { T_ISSUE_A(m0cur, xcur + 64, 1, 3); T_ISSUE_A(m0cur, xcur + 64, 1, 4); }                         \
            else { T_ISSUE_A(m0nxt, xnxt, 0, 3); T_ISSUE_A(m0nxt, xnxt, 0, 4); }                                            \
        }                                                                                                                   \
        FS2T_FENCE();                                                                                                       \
        /* k-slice 1; reads of k-slice 0 of the next step */                                                                \
        FS2T_SLICE(1, MASKED, live0, live1, T_A_ADDR(NT_, 0), T_A_OFF(NH_, NT_), T_B_ADDR(NT_, 0), T_B_OFFS(NT_));          \
    } while (0)
#define T_HALF(H, MASKED)                                                                                                   \
    do {                                                                                                                    \
        T_STEP(H, 0, MASKED); T_STEP(H, 1, MASKED); T_STEP(H, 2, MASKED); T_STEP(H, 3, MASKED); T_STEP(H, 4, MASKED);       \
        if constexpr (TAPS == 9) { T_STEP(H, 5, MASKED); T_STEP(H, 6, MASKED); T_STEP(H, 7, MASKED); T_STEP(H, 8, MASKED); } \
    } while (0)

    FS2T_WAIT_LGKM(0);                                       // kernel arguments: lgkmcnt is ours from here
    t_wait_vm<0>();                                          // my pieces of the first chunk have landed
    __builtin_amdgcn_s_barrier();                            // ... and everybody's
    {
        const unsigned aa = T_A_ADDR(0, 0), ba = T_B_ADDR(0, 0);
        FS2T_DS_READ(Af[0][0], aa, 0); FS2T_DS_READ(Bf[0][0], ba, 0); FS2T_DS_READ(Bf[0][1], ba, 2048);
        FS2T_DS_READ(Bf[0][2], ba, 4096); FS2T_DS_READ(Bf[0][3], ba, 6144); FS2T_DS_READ(Af[0][1], aa, 2048);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(Af[0][0]), "+v"(Af[0][1]), "+v"(Bf[0][0]), "+v"(Bf[0][1]), "+v"(Bf[0][2]), "+v"(Bf[0][3]) :: "memory");
    }
    bool first_chunk = true;
    for (int k = 0; k < ntiles; ++k) {
        int mt, nt, kc0_, nparts;
        t_tile_of(k, units, mt, nt, kc0_, nparts);
        const int nkc_k = (k == ntiles - 1) ? nkc_last : nkc;
        const int m0 = mt * T_TM, n0 = nt * 128;
        // tap-validity bits of this lane's two rows (bit j: tap j stays inside the row's own sequence).  Two scalars, not an
        // array: an array captured by a lambda stayed in scratch in the first version, and a scratch load waits on vmcnt(0).
        const unsigned full_ = (1u << TAPS) - 1u;
        auto row_mask = [=](int m) -> unsigned {
            unsigned msk = 0;
            if (m < a.M) {
                const int t = m % a.S;
#pragma unroll
                for (int j = 0; j < TAPS; ++j) {
                    const int ts = t + j - a.pad;
                    if (ts >= 0 && ts < a.S) msk |= 1u << j;
                }
            }
            return msk;
        };
        unsigned vm0 = row_mask(m0 + wave * 64 + fl), vm1 = row_mask(m0 + wave * 64 + 32 + fl);
        const bool need_mask = __builtin_amdgcn_ballot_w64(vm0 != full_ || vm1 != full_) != 0ull;      // wave-uniform
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

        // (one chunk loop per mask variant: with the variant chosen inside ONE loop the register allocator parked the accumulators in
        // scratch around the loop header)
#define T_CHUNKS(MASKED)                                                                                                    \
        for (int c = 0; c < nkc_k; ++c) {                                                                                   \
            if (!first_chunk) {                                  /* the chunk being multiplied moves on; so does the one after it */ \
                xcur = xnxt; wcur = wnxt; m0cur = m0nxt; offB = offBn;                                                      \
                T_ADVANCE_NEXT();                                                                                           \
            }                                                                                                               \
            first_chunk = false;                                                                                            \
            T_HALF(0, MASKED); T_HALF(1, MASKED);                                                                           \
        }
        if (need_mask) { T_CHUNKS(true) } else { T_CHUNKS(false) }
#undef T_CHUNKS
        // the six fragments prefetched for the next unit's first k-slice: land them before the compiler may copy their registers
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(Af[0][0]), "+v"(Af[0][1]), "+v"(Bf[0][0]), "+v"(Bf[0][1]), "+v"(Bf[0][2]), "+v"(Bf[0][3]) :: "memory");

        if (ABL & 16) {
            float s_ = 0.f;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) s_ += acc[i][j][0];
            if (s_ == 12345.678f) reinterpret_cast<bf16_t*>(a.Y)[0] = 0;
        } else if (nparts > 1) {
            t_epilogue_part(sc.tws + (size_t)sc.b * (T_TM * 128), acc, wave, fl, fh);
        } else {
            FS2_ACT_DISPATCH(a.act, (p_epilogue<ACT>(a, acc, m0, n0, wave, fl, fh, nullptr, bias_s, lane)));
        }
        // stores and DMA loads share vmcnt and may retire out of order with each other: no counted wait is safe until the stores
        // are gone (the pieces already in flight for the next unit land meanwhile)
        t_wait_vm<0>();
    }
#undef T_HALF
#undef T_STEP
#undef T_VMCNT
#undef FS2T_SLICE
#undef FS2T_MFMA
#undef FS2T_DS_READ
#undef T_ISSUE_A
#undef T_ISSUE_B
#undef T_ADVANCE_NEXT
}

// Tail tiles -> Y (the 512-row form of p_tail_finalize_kernel).  Grid (8, G/2): workgroups (.., i) look at contraction workgroup
// b = 2i (order 1: XCD i & 7, place 2 (i >> 3)); when b ran part 0 of a split tail tile they sum the tile's tks slabs and apply
// bias / activation / residual / gate / scale; every other workgroup exits.
template <int ACT>
__global__ void __launch_bounds__(256) t_tail_finalize_kernel(ConvGemmArgs a, PSched sc) {
    const int i = blockIdx.y;
    sc.b = sc.order == 0 ? 2 * i : (i & 7) + 8 * (2 * (i >> 3));
    if (sc.b >= sc.G) return;
    const PPlan p = p_plan(sc, sc.b);
    if (p.tks <= 1 || p.j >= p.tail * p.tks || (p.j % p.tks) != 0) return;
    int mi, nt, sp;
    p_pos(sc, p, p.R * p.Gg + p.j / p.tks, mi, nt, sp);
    const bool gate = a.act == FS2_ACT_GATE;
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
        const int idx = (blockIdx.x * 4 + pass) * 256 + threadIdx.x;      // 0 .. 8191: (row, 8-column chunk) of the tile
        const int r = idx >> 4, cn = (idx & 15) * 8;
        const int m = mi * T_TM + r, n = nt * 128 + cn;
        if (m >= a.M || n >= a.N) continue;
        float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int q = 0; q < p.tks; ++q) {
            const int wg = sc.order == 0 ? sc.b + q : p.x + 8 * (p.j + q);
            const float* sl = sc.tws + (size_t)wg * (T_TM * 128) + r * 128 + cn;
            const float4 x0 = *reinterpret_cast<const float4*>(sl), x1 = *reinterpret_cast<const float4*>(sl + 4);
            v[0] += x0.x; v[1] += x0.y; v[2] += x0.z; v[3] += x0.w; v[4] += x1.x; v[5] += x1.y; v[6] += x1.z; v[7] += x1.w;
        }
        if (a.bias) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += a.bias[n + e];
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = act_ct<ACT>(v[e], a.slope);
        if (a.R) {
            const uint4 rv = *reinterpret_cast<const uint4*>(reinterpret_cast<const bf16_t*>(a.R) + (size_t)m * a.ldr + n);
            const uint32_t* u = reinterpret_cast<const uint32_t*>(&rv);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float r0 = __uint_as_float(u[e] << 16), r1 = __uint_as_float(u[e] & 0xffff0000u);
                v[2 * e] = gate ? (r0 > 0.f ? v[2 * e] : 0.f) : v[2 * e] + r0;
                v[2 * e + 1] = gate ? (r1 > 0.f ? v[2 * e + 1] : 0.f) : v[2 * e + 1] + r1;
            }
        }
        uint4 o;
        uint32_t* ou = reinterpret_cast<uint32_t*>(&o);
#pragma unroll
        for (int e = 0; e < 4; ++e) ou[e] = pack_bf16x2(v[2 * e] * a.out_scale, v[2 * e + 1] * a.out_scale);
        *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(a.Y) + (size_t)m * a.ldy + n) = o;
    }
}

// ------------------------------------------------------------------------------------------------ launcher
static int t_cu_count() {
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

// tail-split scratch: one 512 x 128 f32 slab per workgroup of a full-chip launch
int fs2_conv_gemm_t_tail_ws_bytes(void) { return t_cu_count() * T_TM * 128 * (int)sizeof(float); }

// rounds of tile-times a launch of T tiles takes on G workgroups when its last, partial round may be split tks ways
static double t_rounds(long T, int G, int tks_max, int* tks_out) {
    const long R = T / G, tail = T - R * G;
    int t = 1;
    if (tail > 0) while (2 * t <= tks_max && 2L * t * tail <= G) t *= 2;
    if (tks_out) *tks_out = t;
    return (double)R + (tail > 0 ? 1.0 / t : 0.0);
}
static int t_tks_max(const ConvGemmArgs& a, bool have_ws) {
    if (!have_ws || a.accumulate) return 1;
    const int nkc = a.Cin >> 6;
    int t = 1;
    while (t < 8 && nkc % (2 * t) == 0 && 2 * a.taps * (nkc / (2 * t)) >= 16) t *= 2;      // parts of >= 16 K-steps
    return t;
}

// Shapes the tall kernel can run (pure function of the launch description).
bool fs2_conv_gemm_t_can(const ConvGemmArgs& a, bool has_map, int dtype) {
    if (dtype != FS2_BF16 || a.in_act != FS2_ACT_NONE || a.lens || has_map) return false;
    if (!((a.taps == 5 || a.taps == 9) && a.dil == 1)) return false;
    if (a.Cin % 64 != 0 || !a.vec_ok || a.N % 8 != 0) return false;
    if ((double)a.M * a.ldx * 2 >= 2.0e9 || (double)a.N * a.taps * a.Cin * 2 >= 2.0e9) return false;
    const int cus = t_cu_count();
    const long ntm = fs2_cdiv(a.M, T_TM), ntn = fs2_cdiv(a.N, 128);
    if (ntn > 255 || ntm * ntn < cus) return false;
    if (p_max_units((int)ntm, (int)ntn, 1, cus, 0) > 64 || (cus % 8 == 0 && p_max_units((int)ntm, (int)ntn, 1, cus, 1) > 64)) return false;
    return true;
}
// "Does it pay": the tall kernel takes a launch of the DEFAULT dispatch when its estimated time - rounds of 512-row tiles at
// `speed` x the 256 x 128 kernel's in-tile rate - beats that kernel's by a margin.  Measured speed (r04j, same box): 0.96 on
// the k = 9 FFN forward, 0.86 on the PostNet k = 5 conv - so with the shipped constant it never does; FS2_T_SPEED / FS2_GEMM_T
// (dev builds) exist for the A/B.
bool fs2_conv_gemm_t_ok(const ConvGemmArgs& a, bool has_map, int dtype, bool have_tail_ws) {
    if (!fs2_conv_gemm_t_can(a, has_map, dtype)) return false;
    const int taps = a.taps;
    const int cus = t_cu_count();
    const long ntm = fs2_cdiv(a.M, T_TM), ntn = fs2_cdiv(a.N, 128);
    const int G = cus;
    // estimated time in units of one 256-row tile-time of the 256 x 128 kernel; the tall kernel's tile is 2 x the rows and runs
    // at `speed` x the in-tile rate (measured: profiles/r04*_bench_t.log)
    static const int speed_pct = fs2_dev_env("FS2_T_SPEED", 45);     // 2 x rows per tile at 0.9 x the in-tile rate -> 0.45 per row: never below 0.95 of the old time
    const long T256 = (long)fs2_cdiv(a.M, 256) * ntn;
    const int tks256 = (T256 <= 2L * cus && (long)taps * (a.Cin >> 6) >= 64) ? 8 : 1;
    const double t_old = t_rounds(T256, cus, tks256, nullptr);
    const double t_new = t_rounds(ntm * ntn, G, t_tks_max(a, have_tail_ws), nullptr) * 100.0 / speed_pct;
    static const int force = fs2_dev_env("FS2_GEMM_T", -1);             // dev A/B: 0 = never, 1 = whenever eligible
    if (force == 0) return false;
    if (force == 1) return true;
    return t_new < 0.95 * t_old;
}

template <int TAPS, int ABL>
static void t_launch(const ConvGemmArgs& a, const PSched& sc, hipStream_t stream) {
    static Fs2DevOnce once;
    once.run([&] { (void)hipFuncSetAttribute((const void*)conv_gemm_t_kernel<TAPS, ABL>, hipFuncAttributeMaxDynamicSharedMemorySize, TCfg<TAPS>::LDS); });
    conv_gemm_t_kernel<TAPS, ABL><<<(unsigned)sc.G, 512, TCfg<TAPS>::LDS, stream>>>(a, sc);
}
static void t_launch_any(const ConvGemmArgs& a, const PSched& sc, hipStream_t stream, int abl) {
#ifdef FS2_DEV
#define T_ABL_CASE(K) case K: if (a.taps == 9) t_launch<9, K>(a, sc, stream); else t_launch<5, K>(a, sc, stream); return;
    switch (abl) { T_ABL_CASE(1) T_ABL_CASE(2) T_ABL_CASE(3) T_ABL_CASE(16) T_ABL_CASE(17) T_ABL_CASE(32) T_ABL_CASE(96) T_ABL_CASE(33) T_ABL_CASE(97) T_ABL_CASE(113) default: break; }
#undef T_ABL_CASE
#endif
    (void)abl;
    if (a.taps == 9) t_launch<9, 0>(a, sc, stream); else t_launch<5, 0>(a, sc, stream);
}

void fs2_conv_gemm_t_launch(const ConvGemmArgs& a, hipStream_t stream, float* tail_ws) {
    const int ntm = fs2_cdiv(a.M, T_TM), ntn = fs2_cdiv(a.N, 128);
    const int cus = t_cu_count();
    PSched sc;
    sc.ks = 1; sc.nkc_u = a.Cin >> 6; sc.ws = nullptr;
    sc.tmap = nullptr; sc.ntm = ntm; sc.ntn = ntn; sc.b = 0; sc.n_real = ntm; sc.n_pad = 0;
    sc.G = cus;
    sc.tws = nullptr; sc.tks_max = 1;
    const int tmax = t_tks_max(a, tail_ws != nullptr);
    if (tmax >= 2 && ((long)ntm * ntn) % cus != 0) { sc.tws = tail_ws; sc.tks_max = tmax; }
    sc.order = (ntn >= 2 && sc.G % 8 == 0 && (long)ntm * ntn >= 2L * cus) ? 1 : 0;
    static const int order_env = fs2_dev_env("FS2_P_ORDER", -1);
    if (order_env >= 0) sc.order = (order_env == 1 && sc.G % 8 == 0) ? 1 : 0;
    int abl = 0;
#ifdef FS2_DEV
    static const int abl_env = fs2_dev_env("FS2_T_ABL", 0);
    abl = abl_env;
#endif
    t_launch_any(a, sc, stream, abl);
    if (sc.tws) {
        // (launched even when no workgroup holds a split tail - p_plan decides per group; the empty launch is ~4 us)
        FS2_ACT_DISPATCH(a.act, (t_tail_finalize_kernel<ACT><<<dim3(8, (unsigned)((sc.G + 1) / 2)), 256, 0, stream>>>(a, sc)));
    }
}
