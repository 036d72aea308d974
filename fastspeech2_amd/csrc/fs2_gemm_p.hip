// fs2_gemm_p.hip — PERSISTENT wave-specialised implicit-GEMM for the large bf16 contractions (gfx950).
//
// Same contraction family and LDS operand image as conv_gemm_ring_kernel (fs2_gemm.hip; reference call sites
// transformer/SubLayers.py:39-41,54,87-88, transformer/Layers.py:129-137, model/modules.py:209-240), re-cut after the
// round-1 measurements (profiles/r01k_pmc_mfma.md, VERDICT r01 "what's weak" 5):
//   * 64x64 consumer wave tiles issue one ds_read_b128 per MFMA: 8 consumer waves ask the CU's LDS pipe for as many
//     cycles as the four MFMA pipes have (co-bound by construction).  Here a consumer wave owns 64(M) x 128(N):
//     6 fragment reads per 8 MFMAs, and there are only FOUR consumer waves (one per SIMD, 256 registers each) next to
//     the four loader waves.  A lone consumer hides its reads by issuing exactly one ds_read_b128 behind each of the
//     first six MFMAs of an 8-MFMA k-slice (<= 5 single-issue instructions fit under one 32-cycle MFMA) and waiting
//     with a counted lgkmcnt - no read bursts.
//   * the accumulators are kept TRANSPOSED (D[n][m] = W-fragment x X-fragment): a lane then owns one output row m and
//     four runs of four consecutive columns; one v_permlane32_swap per register pair turns them into 16-byte runs, so
//     the epilogue stores bf16 rows straight from registers - no f32 staging tile in LDS, no barriers, and the operand
//     rings stay live, which is what makes the kernel PERSISTENT: one workgroup per CU walks a static list of tiles and
//     its loader waves prefetch the next tile's operands while the consumers store the current one (round 1 paid ring
//     fill + store burst per 256x128 tile: 86 us of a 304 us launch).
//   * fully padded M-tiles (rows t >= lens[b]) are known before the launch (fs2_tile_map): the real tiles are dealt
//     round-robin to the workgroups (M-fastest: neighbours share the weight slice), the padded ones are zero-filled by
//     the consumer waves while the first operands are in flight.
// Two redesigns were measured against this version in round 2 and NOT kept (git history: "persistent kernel v2" / "v3"):
//   * progress counters in LDS instead of the per-step barrier (loaders publish "step s landed", consumers "step s read", a
//     wave only waits when its data is missing) + an LDS-staged full-line epilogue: correct, but SLOWER at every shape (k=9 FFN
//     conv 247 vs 215 us, QKV 62 vs 52 us): with ONE consumer wave per SIMD every counter read/publish sits in the MFMA issue
//     stream, and the consumer alone (no DMA at all) already needs 203 us for ~105 us of MFMA work - the lone wave's in-order
//     issue of reads + MFMAs is the limiter, not the rendezvous;
//   * the same without any lgkmcnt(0) at the step boundary (counter read one slice early, release published from inside the
//     next slice): no faster (253 us), and it exposed how fragile asm-tracked LDS reads are across code the compiler may
//     re-register (stale copies of in-flight fragments at tile boundaries).
//   * (r02t) an EARLY fragment schedule - every fragment register re-read for the k-slice TWO ahead right behind its last MFMA
//     (same twelve registers, 11-13 MFMA slots between a read and its use instead of 7-8) with the loaders publishing a slot one
//     K-step earlier to make the look-ahead legal: correct (all production-shape tests), but SLOWER - k=9 FFN conv 249 vs 205 us,
//     PostNet k=5 129 vs 109 us (profiles/r02t_early_schedule_bench_p.txt).  Publishing earlier costs one step of DMA prefetch
//     depth (2 K-steps = 40 KB per CU in flight instead of 3), and the loader side is latency-bound (Little's law: ~10 TB/s of
//     L2 -> LDS traffic at ~2 us needs ~78 KB per CU in flight; the rings hold 60): the fragment-read latency it hides is worth
//     less than the DMA depth it gives up.  LDS is full (158 of 160 KB), so the depth cannot be bought back.
// Ablations of THIS version (profiles/r02a_bench_p.md, k=9 FFN conv, 215 us): loaders + barriers alone 97 us, + MFMAs 150 us,
// + fragment reads 215 us; epilogue 25 us of it.
// Synchronisation: one raw s_barrier per K-step (64 deep) publishes the slot the loaders filled D-1 steps earlier and
// releases the slot the consumers just left; LDS-DMA is issued from inline asm and tracked with counted vmcnt
// (loaders), fragment reads with counted lgkmcnt (consumers) - see fs2_gemm.hip for why the compiler cannot do either.
#include "fs2_gemm.h"
#include "fs2_sched.h"
#include "fs2_gemm_epi.h"


template <bool ONE_TAP, bool WIDE> struct PCfg;
// conv (taps >= 3), halo (taps-1)*dil <= 16 rows: activation halo tile (256 + 16 rows) double-buffered per Cin chunk, weight
// ring of D = 5 slots
template <> struct PCfg<false, false> {
    static constexpr int D = 5, HALO = 16, A_BYTES = 272 * 128, SCRATCH = 2 * 272 * 128, B_OFF = 2 * 272 * 128 + 1024, NJA = 9;
    static constexpr int AUX = B_OFF + D * 16384;
};
// conv with a WIDE halo, <= 64 rows (HiFi-GAN's dilated k = 7 / 11 convolutions: (k-1)*dil = 18 ... 50): 320-row halo tile,
// weight ring of D = 4 slots
template <> struct PCfg<false, true> {
    static constexpr int D = 4, HALO = 64, A_BYTES = 320 * 128, SCRATCH = 2 * 320 * 128, B_OFF = 2 * 320 * 128 + 1024, NJA = 10;
    static constexpr int AUX = B_OFF + D * 16384;
};
// taps == 1: activation and weight tiles both in rings of D slots
template <> struct PCfg<true, false> {
    static constexpr int D = 3, HALO = 0, A_BYTES = 256 * 128, SCRATCH = 0, B_OFF = 3 * 256 * 128, NJA = 8;
    static constexpr int AUX = B_OFF + D * 16384;
};
// a pointer the code knows to be wave-uniform, pinned into SGPRs (the LDS-DMA's base operand must be scalar)
__device__ __forceinline__ const unsigned char* p_uniform(const unsigned char* q) {
    const unsigned long long u = reinterpret_cast<unsigned long long>(q);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
    return reinterpret_cast<const unsigned char*>(((unsigned long long)hi << 32) | lo);
}
template <int N> __device__ __forceinline__ void p_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }
static constexpr int P_MAXB = 1024;                       // lens[] staged in LDS
static constexpr int P_AUX_BYTES = P_MAXB * 4 + 8 * 512;  // lens + one 128-float bias line per consumer wave (4 or 8 of them)
static constexpr int P_B_BYTES = 128 * 128;

__device__ __forceinline__ int p_ntiles(const PSched& s) {
    const PPlan p = p_plan(s, s.b);
    return p.R + (p.j < p.tail * p.tks ? 1 : 0);
}
// chunk count of the workgroup's LAST unit (the only one that can be a tail part)
__device__ __forceinline__ int p_last_nkc(const PSched& s) {
    const PPlan p = p_plan(s, s.b);
    return (p.j < p.tail * p.tks) ? s.nkc_u / p.tks : s.nkc_u;
}

// A wave's view of its workgroup's unit list: lane i holds unit 64 * blk + i, packed - v0 = M-tile | N-tile << 20 | parts << 28,
// v1 = first Cin chunk | chunk count << 16 - so that the schedule arithmetic (divisions, the tile-map lookup) runs once per 64
// units, outside the K loops, and a unit costs two v_readlane.  (The first tail-split build recomputed p_unit per tile in three
// places: 35 more spilled SGPRs in the main loop; so did a reload path for workgroups with more than 64 units - the launcher
// sends such launches, > 16 384 output tiles, to the ring kernel instead.)
struct PUnits { unsigned v0, v1; };
__device__ __forceinline__ void p_units_load(const PSched& s, int lane, PUnits& t) {
    int mi, nt, kc0, nkc, np;
    unsigned a0 = 0, a1 = 0;
    if (p_unit(s, lane, mi, nt, kc0, nkc, np)) {
        const int mt = s.tmap ? s.tmap[1 + mi] : mi;
        a0 = (unsigned)mt | ((unsigned)nt << 20) | ((unsigned)np << 28);
        a1 = (unsigned)kc0 | ((unsigned)nkc << 16);
    }
    t.v0 = a0; t.v1 = a1;
}
// k-th unit of this workgroup -> (M-tile, N-tile, first Cin chunk, tail part count, uniform K-split index)
__device__ __forceinline__ void p_tile_of(const PSched& s, int k, const PUnits& t, int& mt, int& nt, int& kc0, int* split = nullptr,
                                          int* nparts = nullptr) {
    const unsigned a0 = __builtin_amdgcn_readlane(t.v0, k), a1 = __builtin_amdgcn_readlane(t.v1, k);
    mt = (int)(a0 & 0xfffffu);
    nt = (int)((a0 >> 20) & 0xffu);
    kc0 = __builtin_amdgcn_readfirstlane((int)(a1 & 0xffffu));          // (uniform already: keeps the DMA base an SGPR operand)
    if (nparts) *nparts = (int)(a0 >> 28);
    if (split) *split = kc0 / s.nkc_u;
}

// ------------------------------------------------------------------------------------------------ loader waves
template <bool ONE_TAP, bool WIDE>
__device__ __forceinline__ void p_loader(const ConvGemmArgs& a, const PSched& sc, unsigned char* smem, int lane, int lw,
                                         const PUnits& units, int ntiles) {
    typedef PCfg<ONE_TAP, WIDE> C;
    constexpr int D = C::D, NJA = C::NJA;
    const unsigned char* Xb = reinterpret_cast<const unsigned char*>(a.X);
    const unsigned char* Wb = reinterpret_cast<const unsigned char*>(a.W);
    const int nkc = sc.nkc_u;                                // Cin chunks per unit (a K-split owns a contiguous range of chunks)
    const int nkc_last = p_last_nkc(sc);                     // ... of the workgroup's last unit (a tail part has fewer)
    const int taps = ONE_TAP ? 1 : a.taps;
    const int nsteps = taps * nkc;
    const int total = (ntiles - 1) * nsteps + taps * nkc_last;   // K-steps of this workgroup, all units
    const int nchunks = (ntiles - 1) * nkc + nkc_last;
    int kc0A = 0, kc0B = 0;                                  // first chunk of the unit the A / B offsets are set for
    const int lr = lane >> 3, lc = lane & 7;
    const int arows = 256 + (taps - 1) * a.dil;
    const unsigned smem_base = lds_addr(smem);

    unsigned offA[NJA], ldsA[NJA], offB[4];
    auto set_A = [&](int k) {
        int mt, nt;
        p_tile_of(sc, k, units, mt, nt, kc0A);
        const int m0 = mt * 256;
#pragma unroll
        for (int j = 0; j < NJA; ++j) {
            const int wl = lw + 4 * j;
            const bool live = wl * 8 < arows;               // wave-uniform
            const int r = wl * 8 + lr;
            const int g = min(max(m0 - a.pad + r, 0), a.M - 1);
            offA[j] = live ? (unsigned)g * (unsigned)(a.ldx * 2) + (unsigned)((lc ^ ((r >> 1) & 7)) << 4) : 0u;
            ldsA[j] = live ? (unsigned)(wl * 1024) : 0xffffffffu;
        }
    };
    auto set_B = [&](int k) {
        int mt, nt;
        p_tile_of(sc, k, units, mt, nt, kc0B);
        const int n0 = nt * 128;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = (lw * 4 + j) * 8 + lr;
            const int n = min(n0 + r, a.N - 1);
            offB[j] = (unsigned)n * (unsigned)(a.ldw * 2) + (unsigned)((lc ^ ((r >> 1) & 7)) << 4);
        }
    };
    // ---- issue side: step being filled (D-1 ahead of the step being consumed); conv: chunk whose halo tile is fetched
    int is = 0, ik = 0, ikc = 0, itap = 0, islot = 0;        // global step, its tile / chunk / tap / ring slot
    int kA = -1, kB = -1;                                    // tiles the per-lane offsets are set for
    auto issue_A = [&](int k, int kc, int buf_or_slot) {
        if (k != kA) { set_A(k); kA = k; }
        const unsigned char* base = p_uniform(Xb + (size_t)(kc0A + kc) * 128);
#pragma unroll
        for (int j = 0; j < NJA; ++j) {
            const unsigned d = (ldsA[j] == 0xffffffffu) ? smem_base + C::SCRATCH : smem_base + buf_or_slot * C::A_BYTES + ldsA[j];
            glds16_sbase(offA[j], base, __builtin_amdgcn_readfirstlane(d));
        }
    };
    auto issue_step = [&]() {                                // operands of global step `is` -> slot `islot`
        if (ik != kB) { set_B(ik); kB = ik; }
        if (ONE_TAP) issue_A(ik, ikc, islot);
        const unsigned char* base = p_uniform(Wb + ((size_t)itap * a.Cin + (size_t)(kc0B + ikc) * 64) * 2);
        const unsigned d0 = smem_base + C::B_OFF + islot * P_B_BYTES + lw * 4096;
#pragma unroll
        for (int j = 0; j < 4; ++j) glds16_sbase(offB[j], base, d0 + j * 1024);
    };
    auto advance_issue = [&]() {
        ++is;
        if (++itap == taps) { itap = 0; if (++ikc == (ik == ntiles - 1 ? nkc_last : nkc)) { ikc = 0; ++ik; } }
        if (++islot == D) islot = 0;
    };
    if (!ONE_TAP) issue_A(0, 0, 0);
#pragma unroll
    for (int p = 0; p < D - 1; ++p) {
        if (is < total) issue_step();
        advance_issue();
    }
    // ---- consume side
    int ck = 0, ckc = 0, ctap = 0, gc = 0;                   // tile / chunk / tap of the step being published; global chunk
    for (int cs = 0; cs < total; ++cs) {
        // my ops of step cs have landed = everything but the ops issued after them (compile-time counts per case)
        if (total - 1 - cs < D - 2) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else if (ONE_TAP) {
            p_wait_vm<(D - 2) * (NJA + 4)>();                                     // taps == 1: (D-2) x (8 A + 4 B)
        } else {
            const bool a_young = (ctap >= 1 && ctap <= D - 2) && (gc + 1 < nchunks);   // halo tile of chunk gc+1 issued after B(cs)
            if (a_young) p_wait_vm<(D - 2) * 4 + NJA>();                           // (D-2) x 4 B + the next chunk's halo tile
            else p_wait_vm<(D - 2) * 4>();
        }
        __builtin_amdgcn_s_barrier();                        // publishes slot(cs); the consumers have left slot(cs-1)
        if (!ONE_TAP && ctap == 0 && gc + 1 < nchunks) {
            int nk = ck, nkc_ = ckc + 1;
            if (nkc_ == (ck == ntiles - 1 ? nkc_last : nkc)) { nkc_ = 0; ++nk; }
            issue_A(nk, nkc_, (gc + 1) & 1);
        }
        if (is < total) issue_step();
        advance_issue();
        if (++ctap == taps) { ctap = 0; ++gc; if (++ckc == (ck == ntiles - 1 ? nkc_last : nkc)) { ckc = 0; ++ck; } }
    }
}

// ------------------------------------------------------------------------------------------------ consumer waves
#define FS2P_DS_READ(dst, addr, OFF) asm volatile("ds_read_b128 %0, %1 offset:" #OFF : "=v"(dst) : "v"(addr))
#define FS2P_WAIT_LGKM(N) asm volatile("s_waitcnt lgkmcnt(" #N ")" ::: "memory")
#define FS2P_FENCE() __builtin_amdgcn_sched_barrier(0)

// K-split epilogue: the partial tile goes into this split's own f32 slab of the workspace with plain 16-byte stores (same lane
// layout as above: after the half-wave swap a lane holds 2 x 8 consecutive columns of one row).  The first version ADDED it
// into one shared M x N slab with float atomics: a lane owns a ROW there, so every atomic instruction touched 64 different
// cache lines (r02f: 48 tiles x 144 K-steps ran 100 us unsplit, 131 us at 2 splits, 185 us at 4).  Bias / activation /
// residual happen once, in splitk_finalize_kernel, which also sums the slabs.
template <int MB>
__device__ __forceinline__ void p_epilogue_splitk(const ConvGemmArgs& a, float* ws, f32x16 (&acc)[MB][4], int m0, int n0, int wm,
                                                  int fl, int fh) {
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
        const int m = m0 + wm * (32 * MB) + mb * 32 + fl;
        if (m >= a.M) continue;
        float* wrow = ws + (size_t)m * a.N;
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
                const int n = n0 + nb * 32 + ch * 16 + fh * 8;
                if (n >= a.N) continue;
                *reinterpret_cast<float4*>(wrow + n) = make_float4(c[ch][0], c[ch][1], c[ch][2], c[ch][3]);
                *reinterpret_cast<float4*>(wrow + n + 4) = make_float4(c[ch][4], c[ch][5], c[ch][6], c[ch][7]);
            }
        }
    }
}

// ws (f32, ks slabs of M x N) -> Y: slab sum, bias, activation, residual / gate, scale, padded-row zero, bf16.  One thread = 8
// consecutive columns of one row.  Rows of M-tiles the tile map removed were written by nobody: they are padded rows and come
// out as zeros whatever the slabs hold.
template <int ACT>
__global__ void __launch_bounds__(256) splitk_finalize_kernel(ConvGemmArgs a, const float* __restrict__ ws, int ks) {
    const int cpr = a.N >> 3;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)a.M * cpr) return;
    const int m = (int)(idx / cpr), n = (int)(idx - (long)m * cpr) * 8;
    const float* wp = ws + (size_t)m * a.N + n;
    const size_t slab = (size_t)a.M * a.N;
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int sp = 0; sp < ks; ++sp) {
        const float4 x0 = *reinterpret_cast<const float4*>(wp + sp * slab), x1 = *reinterpret_cast<const float4*>(wp + sp * slab + 4);
        v[0] += x0.x; v[1] += x0.y; v[2] += x0.z; v[3] += x0.w; v[4] += x1.x; v[5] += x1.y; v[6] += x1.z; v[7] += x1.w;
    }
    if (a.bias) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += a.bias[n + e];
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = act_ct<ACT>(v[e], a.slope);
    const bool gate = a.act == FS2_ACT_GATE;
    if (a.R) {
        const uint4 rv = *reinterpret_cast<const uint4*>(reinterpret_cast<const bf16_t*>(a.R) + (size_t)m * a.ldr + n);
        const uint32_t* u = reinterpret_cast<const uint32_t*>(&rv);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float r0 = __uint_as_float(u[e] << 16), r1 = __uint_as_float(u[e] & 0xffff0000u);
            if (a.res_unlrelu > 0.f) { r0 = r0 > 0.f ? r0 : r0 * a.res_unlrelu; r1 = r1 > 0.f ? r1 : r1 * a.res_unlrelu; }
            v[2 * e] = gate ? (r0 > 0.f ? v[2 * e] : 0.f) : v[2 * e] + r0;
            v[2 * e + 1] = gate ? (r1 > 0.f ? v[2 * e + 1] : 0.f) : v[2 * e + 1] + r1;
        }
    }
    bool padrow = false;
    if (a.lens) { const int b = m / a.S; padrow = (m - b * a.S) >= a.lens[b]; }
    uint4 o;
    uint32_t* ou = reinterpret_cast<uint32_t*>(&o);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float x0 = v[2 * e] * a.out_scale, x1 = v[2 * e + 1] * a.out_scale;
        if (a.post_slope > 0.f) { x0 = x0 > 0.f ? x0 : x0 * a.post_slope; x1 = x1 > 0.f ? x1 : x1 * a.post_slope; }
        ou[e] = padrow ? 0u : pack_bf16x2(x0, x1);
    }
    *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(a.Y) + (size_t)m * a.ldy + n) = o;
}

// tail part: the partial 256x128 tile goes to the workgroup's slab, tile-local row-major (row = wm*64 + mb*32 + fl)
template <int MB>
__device__ __forceinline__ void p_epilogue_part(float* slab, f32x16 (&acc)[MB][4], int wm, int fl, int fh) {
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
        float* wrow = slab + (size_t)(wm * (32 * MB) + mb * 32 + fl) * 128;
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

// Tail tiles -> Y.  Grid (4, G/2): workgroups (.., i) look at contraction workgroup b = 2i (order 1: XCD i & 7, place 2 (i >> 3))
// - a split tile's part 0 always sits on an even place; when b ran part 0 of a split tail tile they sum the tile's tks slabs
// (the workgroups b .. of the same group) and apply bias / activation / residual / gate / scale / padded-row zero, one thread
// per 8 consecutive columns of one row, 4 passes each over a quarter of the 256 x 128 tile.  Every other workgroup exits.
// (16 workgroups per candidate - 4096 in all - cost 11 us per launch, most of it dispatching empty ones; one per candidate
// walking 16 dependent passes was slower still.)
template <int ACT>
__global__ void __launch_bounds__(256) p_tail_finalize_kernel(ConvGemmArgs a, PSched sc) {
    const int i = blockIdx.y;
    sc.b = sc.order == 0 ? 2 * i : (i & 7) + 8 * (2 * (i >> 3));
    if (sc.b >= sc.G) return;
    if (sc.tmap) { sc.n_real = sc.tmap[0]; sc.n_pad = sc.ntm - sc.n_real; }
    const PPlan p = p_plan(sc, sc.b);
    if (p.tks <= 1 || p.j >= p.tail * p.tks || (p.j % p.tks) != 0) return;
    int mi, nt, sp;
    p_pos(sc, p, p.R * p.Gg + p.j / p.tks, mi, nt, sp);
    const int mt = sc.tmap ? sc.tmap[1 + mi] : mi;
    const bool gate = a.act == FS2_ACT_GATE;
#pragma unroll
  for (int pass = 0; pass < 4; ++pass) {
    const int idx = (blockIdx.x * 4 + pass) * 256 + threadIdx.x;      // 0 .. 4095: (row, 8-column chunk) of the tile
    const int r = idx >> 4, cn = (idx & 15) * 8;
    const int m = mt * 256 + r, n = nt * 128 + cn;
    if (m >= a.M || n >= a.N) continue;
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int q = 0; q < p.tks; ++q) {
        const int wg = sc.order == 0 ? sc.b + q : p.x + 8 * (p.j + q);
        const float* sl = sc.tws + (size_t)wg * (256 * 128) + r * 128 + cn;
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
            float r0 = __uint_as_float(u[e] << 16), r1 = __uint_as_float(u[e] & 0xffff0000u);
            if (a.res_unlrelu > 0.f) { r0 = r0 > 0.f ? r0 : r0 * a.res_unlrelu; r1 = r1 > 0.f ? r1 : r1 * a.res_unlrelu; }
            v[2 * e] = gate ? (r0 > 0.f ? v[2 * e] : 0.f) : v[2 * e] + r0;
            v[2 * e + 1] = gate ? (r1 > 0.f ? v[2 * e + 1] : 0.f) : v[2 * e + 1] + r1;
        }
    }
    bool padrow = false;
    if (a.lens) { const int b = m / a.S; padrow = (m - b * a.S) >= a.lens[b]; }
    uint4 o;
    uint32_t* ou = reinterpret_cast<uint32_t*>(&o);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float x0 = v[2 * e] * a.out_scale, x1 = v[2 * e + 1] * a.out_scale;
        if (a.post_slope > 0.f) { x0 = x0 > 0.f ? x0 : x0 * a.post_slope; x1 = x1 > 0.f ? x1 : x1 * a.post_slope; }
        ou[e] = padrow ? 0u : pack_bf16x2(x0, x1);
    }
    *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(a.Y) + (size_t)m * a.ldy + n) = o;
  }
}

// leaky-ReLU on a landed activation fragment (HiFi-GAN's pre-activation convolutions; the LDS-DMA path cannot transform data on
// its way in): widen both halves of each dword, max(x, slope x) (0 < slope < 1), hardware bf16 pack - 7 VALU per dword, two
// fragments per k-slice of eight MFMAs
__device__ __forceinline__ u32x4 p_lrelu(u32x4 v, float slope) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float lo = __uint_as_float(v[e] << 16), hi = __uint_as_float(v[e] & 0xffff0000u);
        v[e] = pack_bf16x2(fmaxf(lo, lo * slope), fmaxf(hi, hi * slope));
    }
    return v;
}

// ABL bit 16 (shipped): leaky-ReLU prologue on the activation fragments.  Other ABL bits (dev builds only): 1 = no MFMA, 2 = no fragment reads, 4 = no epilogue
// MB = 2: four consumer waves of 64 x 128 (one per SIMD beside its loader wave); MB = 1: EIGHT consumer waves of 32 x 128 (two per
// SIMD: while one waits for a fragment or at its counted lgkmcnt the other issues MFMAs - the lone wave's in-order issue of reads
// and MFMAs was this kernel's measured limiter, see the header) at 5 fragment reads per 4 MFMAs instead of 6 per 8
template <bool ONE_TAP, bool WIDE, int ABL, int MB>
__device__ __forceinline__ void p_consumer(const ConvGemmArgs& a, const PSched& sc, unsigned char* smem, int lane, int wm,
                                           const PUnits& units, int ntiles, const int32_t* lens_s, float* bias_s) {
    typedef PCfg<ONE_TAP, WIDE> C;
    constexpr int D = C::D;
    const int nkc = sc.nkc_u;
    const int nkc_last = p_last_nkc(sc);
    const int taps = ONE_TAP ? 1 : a.taps;
    const int nsteps = taps * nkc;
    const int total = (ntiles - 1) * nsteps + taps * nkc_last;
    const int fl = lane & 31, fh = lane >> 5;
    const unsigned smem_u = lds_addr(smem);
    // fragment addresses.  Lane (fl, fh) reads 16-byte chunk c = fh*4 + j of its row in k-slice j; chunks are XOR-swizzled
    // with the key ((physical row >> 1) & 7), and (c0 | j) ^ key == (c0 ^ key) ^ j because c0 = fh*4 has its low bits clear:
    // one per-lane constant per operand and step, the k-slice enters as an immediate XOR.
    const unsigned c0 = (unsigned)(fh * 4);
    const unsigned ckb = c0 ^ (unsigned)((fl >> 1) & 7);                     // weight rows are never shifted
    const unsigned blane = smem_u + C::B_OFF + (unsigned)(fl * 128);
    const unsigned alane = smem_u + (unsigned)((wm * (32 * MB) + fl) * 128);
    auto a_base = [&](int abuf, int tap) -> unsigned { return alane + (unsigned)(abuf * C::A_BYTES + tap * a.dil * 128); };
    auto a_key = [&](int tap) -> unsigned { return c0 ^ (unsigned)(((fl + tap * a.dil) >> 1) & 7); };   // key of the PHYSICAL halo row
    auto b_base = [&](int slot) -> unsigned { return blane + (unsigned)(slot * P_B_BYTES); };

    u32x4 Af[2][MB], Bf[2][4];                               // [set][mb], [set][nb]: fragment double buffer
    f32x16 acc[MB][4];

    // one k-slice: 8 MFMAs of fragment set SET, the six reads of the NEXT slice (k-slice JN of the step whose operands sit at
    // ABASE / AKEY / BBASE, into set SET^1) issued one behind each of the first six MFMAs.
    // Read order A0 B0 B1 B2 B3 A1 / MFMA order (B0,A0) (B1,A0) (B2,A0) (B3,A0) (B0,A1) ... : the operand an MFMA needs was
    // issued >= 4 MFMA slots earlier and "at most 4 younger LDS reads outstanding" is the same count at every position
    // (lgkmcnt counts LDS operations in order).
#define FS2P_MFMA(SET, MB, NB, AV)                                                                                          \
    if (!(ABL & 1)) acc[MB][NB] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, Bf[SET][NB]),        \
                                                                       __builtin_bit_cast(bf16x8, AV), acc[MB][NB], 0, 0, 0)
#define FS2P_SLICE(SET, MASKED, LIVE0, LIVE1, DO_READS, JN, ABASE, AKEY, BBASE)                                             \
    do {                                                                                                                    \
        u32x4 av0, av1;                                                                                                     \
        FS2P_WAIT_LGKM(4); FS2P_FENCE();                                                                                    \
        av0 = Af[SET][0]; if (ABL & 16) av0 = p_lrelu(av0, a.in_slope);                                                     \
        if (MASKED && !(LIVE0)) av0 = u32x4{0u, 0u, 0u, 0u};                                                                \
        FS2P_MFMA(SET, 0, 0, av0); FS2P_FENCE();                                                                            \
        const unsigned aa_ = (ABASE) + ((((AKEY)) ^ (unsigned)(JN)) << 4);                                                  \
        const unsigned ba_ = (BBASE) + ((ckb ^ (unsigned)(JN)) << 4);                                                       \
        if (DO_READS) { FS2P_DS_READ(Af[SET ^ 1][0], aa_, 0); } FS2P_FENCE();                                               \
        FS2P_WAIT_LGKM(4); FS2P_FENCE();                                                                                    \
        FS2P_MFMA(SET, 0, 1, av0); FS2P_FENCE();                                                                            \
        if (DO_READS) { FS2P_DS_READ(Bf[SET ^ 1][0], ba_, 0); } FS2P_FENCE();                                               \
        FS2P_WAIT_LGKM(4); FS2P_FENCE();                                                                                    \
        FS2P_MFMA(SET, 0, 2, av0); FS2P_FENCE();                                                                            \
        if (DO_READS) { FS2P_DS_READ(Bf[SET ^ 1][1], ba_, 4096); } FS2P_FENCE();                                            \
        FS2P_WAIT_LGKM(4); FS2P_FENCE();                                                                                    \
        FS2P_MFMA(SET, 0, 3, av0); FS2P_FENCE();                                                                            \
        if (DO_READS) { FS2P_DS_READ(Bf[SET ^ 1][2], ba_, 8192); } FS2P_FENCE();                                            \
        FS2P_WAIT_LGKM(4); FS2P_FENCE();                                                                                    \
        av1 = Af[SET][1]; if (ABL & 16) av1 = p_lrelu(av1, a.in_slope);                                                     \
        if (MASKED && !(LIVE1)) av1 = u32x4{0u, 0u, 0u, 0u};                                                                \
        FS2P_MFMA(SET, 1, 0, av1); FS2P_FENCE();                                                                            \
        if (DO_READS) { FS2P_DS_READ(Bf[SET ^ 1][3], ba_, 12288); } FS2P_FENCE();                                           \
        FS2P_MFMA(SET, 1, 1, av1); FS2P_FENCE();                                                                            \
        if (DO_READS) { FS2P_DS_READ(Af[SET ^ 1][1], aa_, 4096); } FS2P_FENCE();                                            \
        FS2P_MFMA(SET, 1, 2, av1); FS2P_FENCE();                                                                            \
        FS2P_MFMA(SET, 1, 3, av1); FS2P_FENCE();                                                                            \
    } while (0)

    // MB == 1: 4 MFMAs per k-slice, the FIVE reads of the next slice in the order A0 B0 B1 B2 B3; MFMA i needs read i + 1 of the
    // five issued a slice ago: "at most 3 younger LDS reads outstanding" at every position
#define FS2P_SLICE1(SET, MASKED, LIVE0, DO_READS, JN, ABASE, AKEY, BBASE)                                                   \
    do {                                                                                                                    \
        u32x4 av0;                                                                                                          \
        FS2P_WAIT_LGKM(3); FS2P_FENCE();                                                                                    \
        av0 = Af[SET][0]; if (ABL & 16) av0 = p_lrelu(av0, a.in_slope);                                                     \
        if (MASKED && !(LIVE0)) av0 = u32x4{0u, 0u, 0u, 0u};                                                                \
        FS2P_MFMA(SET, 0, 0, av0); FS2P_FENCE();                                                                            \
        const unsigned aa_ = (ABASE) + ((((AKEY)) ^ (unsigned)(JN)) << 4);                                                  \
        const unsigned ba_ = (BBASE) + ((ckb ^ (unsigned)(JN)) << 4);                                                       \
        if (DO_READS) { FS2P_DS_READ(Af[SET ^ 1][0], aa_, 0); } FS2P_FENCE();                                               \
        FS2P_WAIT_LGKM(3); FS2P_FENCE();                                                                                    \
        FS2P_MFMA(SET, 0, 1, av0); FS2P_FENCE();                                                                            \
        if (DO_READS) { FS2P_DS_READ(Bf[SET ^ 1][0], ba_, 0); } FS2P_FENCE();                                               \
        FS2P_WAIT_LGKM(3); FS2P_FENCE();                                                                                    \
        FS2P_MFMA(SET, 0, 2, av0); FS2P_FENCE();                                                                            \
        if (DO_READS) { FS2P_DS_READ(Bf[SET ^ 1][1], ba_, 4096); } FS2P_FENCE();                                            \
        FS2P_WAIT_LGKM(3); FS2P_FENCE();                                                                                    \
        FS2P_MFMA(SET, 0, 3, av0); FS2P_FENCE();                                                                            \
        if (DO_READS) { FS2P_DS_READ(Bf[SET ^ 1][2], ba_, 8192); } FS2P_FENCE();                                            \
        if (DO_READS) { FS2P_DS_READ(Bf[SET ^ 1][3], ba_, 12288); } FS2P_FENCE();                                           \
    } while (0)
    // the slice of this instantiation (the other macro's body is discarded by the if constexpr)
#define FS2P_SL(SET, MASKED, LIVE0, LIVE1, DO_READS, JN, ABASE, AKEY, BBASE)                                                \
    do {                                                                                                                    \
        if constexpr (MB == 2) { FS2P_SLICE(SET, MASKED, LIVE0, LIVE1, DO_READS, JN, ABASE, AKEY, BBASE); }                 \
        else { FS2P_SLICE1(SET, MASKED, LIVE0, DO_READS, JN, ABASE, AKEY, BBASE); }                                         \
    } while (0)

    auto land_set0 = [&]() {                                 // lgkmcnt(0) with fragment set 0 as its OUTPUTS (see the two call sites)
        if constexpr (MB == 2)
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(Af[0][0]), "+v"(Af[0][1]), "+v"(Bf[0][0]), "+v"(Bf[0][1]), "+v"(Bf[0][2]), "+v"(Bf[0][3]) :: "memory");
        else
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(Af[0][0]), "+v"(Bf[0][0]), "+v"(Bf[0][1]), "+v"(Bf[0][2]), "+v"(Bf[0][3]) :: "memory");
    };
    int gs = 0, gc = 0, slot = 0;                            // global step / chunk, ring slot of step gs
    FS2P_WAIT_LGKM(0);                                       // kernel arguments, lens staging: lgkmcnt is ours from here
    __builtin_amdgcn_s_barrier();                            // slot(0) published
    if (!(ABL & 2)) {
        const unsigned aa = a_base(0, 0) + (a_key(0) << 4), ba = b_base(0) + (ckb << 4);
        FS2P_DS_READ(Af[0][0], aa, 0); FS2P_DS_READ(Bf[0][0], ba, 0); FS2P_DS_READ(Bf[0][1], ba, 4096);
        FS2P_DS_READ(Bf[0][2], ba, 8192); FS2P_DS_READ(Bf[0][3], ba, 12288);
        if constexpr (MB == 2) FS2P_DS_READ(Af[0][1], aa, 4096);
        // these six sit outside the tile loop: have them LANDED before the compiler may copy their registers into the
        // loop-carried ones (a copy of a register with a read in flight would carry stale data)
        land_set0();
    }
    for (int k = 0; k < ntiles; ++k) {
        int mt, nt, kc0_unused;
        int split, nparts;
        p_tile_of(sc, k, units, mt, nt, kc0_unused, &split, &nparts);
        const int nkc_k = (k == ntiles - 1) ? nkc_last : nkc;
        const int m0 = mt * 256, n0 = nt * 128;
        // tap-validity bits of this lane's two rows (bit j: tap j stays inside the row's own sequence)
        unsigned vmask[2] = {0xffffffffu, 0xffffffffu};
        bool need_mask = false;
        if (!ONE_TAP) {
            const unsigned full = (taps >= 32) ? 0xffffffffu : ((1u << taps) - 1u);
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                const int m = m0 + wm * (32 * MB) + mb * 32 + fl;
                unsigned msk = 0;
                if (m < a.M) {
                    const int t = m % a.S;
                    for (int j = 0; j < taps; ++j) {
                        const int ts = t + j * a.dil - a.pad;
                        if (ts >= 0 && ts < a.S) msk |= 1u << j;
                    }
                }
                vmask[mb] = msk;
                need_mask = need_mask || (msk != full);
            }
            need_mask = __builtin_amdgcn_ballot_w64(need_mask) != 0ull;      // wave-uniform
        }
#pragma unroll
        for (int i = 0; i < MB; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

        auto run_tile = [&](auto maskc) {
            constexpr bool MASKED = decltype(maskc)::value;
            // ONE loop over the unit's K-steps (chunk-major, taps inside a chunk).  As two nested loops the compiler gave the inner loop its
            // own registers for the loop-carried fragment set and COPIED the outer loop's into them at the inner loop's entry - with the
            // reads of that set still in flight (stale copies: the eight-wave form's last-issued fragment came out wrong in a timing-
            // dependent 1 % of the outputs).  tools/check_frag_copies.py scans the ISA for such copies.
            int tap = 0;
            for (int st = 0, nst = nkc_k * taps; st < nst; ++st) {
                {
                    const int abuf = ONE_TAP ? slot : (gc & 1);
                    const bool live0 = (vmask[0] >> tap) & 1u, live1 = (vmask[1] >> tap) & 1u;
                    const bool more = gs + 1 < total;                       // another step follows (this tile or the next)
                    const unsigned ab = a_base(abuf, tap), ak = a_key(tap), bb = b_base(slot);
                    FS2P_SL(0, MASKED, live0, live1, !(ABL & 2), 1, ab, ak, bb);
                    FS2P_SL(1, MASKED, live0, live1, !(ABL & 2), 2, ab, ak, bb);
                    FS2P_SL(0, MASKED, live0, live1, !(ABL & 2), 3, ab, ak, bb);
                    FS2P_WAIT_LGKM(0); FS2P_FENCE();                        // every read of slot(gs) / its halo tile has landed
                    if (more) __builtin_amdgcn_s_barrier();                 // slot(gs+1) published, slot(gs) released
                    FS2P_FENCE();
                    int nslot = slot + 1; if (nslot == D) nslot = 0;
                    int ntap = tap + 1, ngc = gc; if (ntap == taps) { ntap = 0; ++ngc; }
                    const int nabuf = ONE_TAP ? nslot : (ngc & 1);
                    const unsigned nab = a_base(nabuf, ntap), nak = a_key(ntap), nbb = b_base(nslot);
                    // (after the workgroup's very last step these six reads fetch operands nobody uses: issuing them unconditionally keeps the
                    // fragment registers single-definition - a conditional read would make the compiler merge two register sets with
                    // copies, and a copy of a register with a read in flight carries stale data)
                    FS2P_SL(1, MASKED, live0, live1, !(ABL & 2), 0, nab, nak, nbb);
                    ++gs; slot = nslot; gc = ngc; tap = ntap;
                }
            }
        };
        if (need_mask) run_tile(std::true_type{}); else run_tile(std::false_type{});
        // The six fragments prefetched for the NEXT unit's first k-slice are in flight here and stay in registers across the
        // epilogue.  The masked and unmasked loops are separate code with their own register assignment, so the compiler may
        // COPY those registers at a unit boundary - and a copy of a register whose read is still in flight carries stale data
        // (seen in the round-2 "v3" experiment: whole 32-row blocks of the last-issued fragment wrong in a few tiles).  Land
        // them first, AS OUTPUTS of the wait, so every copy is ordered behind it (one LDS drain per tile, not per step).
        land_set0();

        if (nparts > 1) {
            p_epilogue_part(sc.tws + (size_t)sc.b * (256 * 128), acc, wm, fl, fh);
        } else if (sc.ks > 1) {
            p_epilogue_splitk(a, sc.ws + (size_t)split * a.M * a.N, acc, m0, n0, wm, fl, fh);
        } else if (!(ABL & 4)) {
            FS2_ACT_DISPATCH(a.act, (p_epilogue<ACT, MB>(a, acc, m0, n0, wm, fl, fh, lens_s, bias_s, lane)));
        } else {
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < MB; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) s += acc[i][j][0];
            if (s == 12345.678f) reinterpret_cast<bf16_t*>(a.Y)[0] = 0;
        }
        if (ABL & 2) asm volatile("" :: "v"(Af[0][0]), "v"(Bf[0][0]));
    }
#undef FS2P_SL
#undef FS2P_SLICE1
#undef FS2P_SLICE
#undef FS2P_MFMA
}

// CW consumer waves (4: 64 x 128 each, 512 threads, 256 registers per wave; 8: 32 x 128 each, 768 threads, 168 registers) + 4 loader waves
template <bool ONE_TAP, bool WIDE, int ABL, int CW>
__global__ void __launch_bounds__(CW * 64 + 256, (CW + 4) / 4) conv_gemm_p_kernel(ConvGemmArgs a, PSched sc0) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef PCfg<ONE_TAP, WIDE> C;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    PSched sc = sc0;
    sc.b = blockIdx.x;
    if (sc.tmap) {                                           // the real-tile count lives on the device (no host sync anywhere)
        sc.n_real = __builtin_amdgcn_readfirstlane(sc.tmap[0]);
        sc.n_pad = sc.ntm - sc.n_real;
    }
    int32_t* lens_s = reinterpret_cast<int32_t*>(smem + C::AUX);
    float* bias_s = reinterpret_cast<float*>(smem + C::AUX + P_MAXB * 4) + (wave & 7) * 128;     // (consumer waves only)
    const int ntiles = p_ntiles(sc);                         // 0 when every real tile went to other workgroups
    // this workgroup's units (loaded separately in the two roles: a wave keeps its own packed copy in two VGPRs)
    PUnits units;
    p_units_load(sc, lane, units);
    if (wave < CW) {
        // per-sequence lengths -> LDS (the epilogue's padded-row test must not wait on global memory behind its own stores)
        if (a.lens) {
            const int B = a.M / a.S;
            for (int i = tid; i < B; i += CW * 64) lens_s[i] = a.lens[i];
        }
        // fully padded tiles of this workgroup: zeros (while the loaders fill the rings)
        if (!a.accumulate && sc.n_pad > 0 && sc.ks == 1) {      // (K-split launches: splitk_finalize_kernel writes every row)
            bf16_t* Y = reinterpret_cast<bf16_t*>(a.Y);
            const int n_pu = sc.n_pad * sc.ntn;
            for (int p = sc.b; p < n_pu; p += sc.G) {
                const int nt = p / sc.n_pad, pi = p - nt * sc.n_pad;
                const int m0 = sc.tmap[1 + sc.n_real + pi] * 256, n0 = nt * 128;
                for (int i = tid; i < 256 * 16; i += CW * 64) {
                    const int m = m0 + (i >> 4), n = n0 + (i & 15) * 8;
                    if (m < a.M && n < a.N) *reinterpret_cast<uint4*>(Y + (size_t)m * a.ldy + n) = make_uint4(0, 0, 0, 0);
                }
            }
        }
        // the four consumer waves make the staged lengths visible to each other before anyone's epilogue: they all pass
        // the per-step barriers (>= 1) before the first epilogue, and LDS writes are ordered ahead of the wave's barrier
        // arrival by the s_waitcnt lgkmcnt(0) in front of the first barrier.
        if (ntiles > 0) p_consumer<ONE_TAP, WIDE, ABL, 8 / CW>(a, sc, smem, lane, wave, units, ntiles, lens_s, bias_s);
    } else if (ntiles > 0) {                                 // (a workgroup without real tiles runs no barrier on either side)
        p_loader<ONE_TAP, WIDE>(a, sc, smem, lane, wave - CW, units, ntiles);
    }
}

// ------------------------------------------------------------------------------------------------ tile map
// out[0] = number of REAL 256-row M-tiles, out[1 ..] = their indices (ascending), then the fully padded ones.  A tile is
// padded when all of its rows belong to ONE sequence's tail t >= lens[b] (the same rule the other GEMM kernels apply).
__global__ void __launch_bounds__(64) tile_map_kernel(const int32_t* __restrict__ lens, int M, int S, int rows, int32_t* __restrict__ out) {
    const int lane = threadIdx.x;
    const int ntm = (M + rows - 1) / rows;
    int n_real = 0;
    for (int pass = 0; pass < 2; ++pass) {
        int n = 0;
        for (int base = 0; base < ntm; base += 64) {
            const int tm = base + lane;
            bool real = false, valid = tm < ntm;
            if (valid) {
                const int m0 = tm * rows, mlast = min(m0 + rows - 1, M - 1);
                const int b0 = m0 / S, b1 = mlast / S;
                real = !(b0 == b1 && (m0 - b0 * S) >= lens[b0]);
            }
            const bool pick = valid && (pass == 0 ? real : !real);
            const unsigned long long msk = __builtin_amdgcn_ballot_w64(pick);
            const int pos = n + __builtin_popcountll(msk & ((1ull << lane) - 1ull));
            if (pick) out[1 + (pass == 0 ? 0 : n_real) + pos] = tm;
            n += __builtin_popcountll(msk);
        }
        if (pass == 0) { n_real = n; if (lane == 0) out[0] = n; }
    }
}

// Everything the step derives from a lengths vector, in ONE launch (was ~7 tiny ones per vector: int32 cast, clamp, arange +
// compare for the mask the model returns, two reductions for the loss's valid count, the tile map):
//   lens32[b] = min(len[b], S)      mask[b][t] = t >= len[b] (bool, True = padding; reference utils/tools.py:91-99)
//   count[0]  = sum_b lens32[b]      tile_map   = as fs2_tile_map (rows-row M-tiles; optional)
__global__ void __launch_bounds__(256) lens_prep_kernel(const int64_t* __restrict__ lens, int B, int S, int rows,
                                                        int32_t* __restrict__ lens32, unsigned char* __restrict__ mask,
                                                        float* __restrict__ count, int32_t* __restrict__ tmap) {
    if (blockIdx.x > 0) {                                    // mask: one byte per position
        const long i0 = ((long)(blockIdx.x - 1) * 256 + threadIdx.x) * 4, n = (long)B * S;
        for (int e = 0; e < 4; ++e) {
            const long i = i0 + e;
            if (i < n) { const int b = (int)(i / S); mask[i] = (i - (long)b * S) >= lens[b]; }
        }
        return;
    }
    __shared__ int s_sum[4];
    int part = 0;
    for (int b = threadIdx.x; b < B; b += 256) {
        const long l = lens[b];
        const int v = (int)(l < 0 ? 0 : (l > S ? S : l));
        lens32[b] = v;
        part += v;
    }
    for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
    if ((threadIdx.x & 63) == 0) s_sum[threadIdx.x >> 6] = part;
    __syncthreads();
    if (threadIdx.x == 0) count[0] = (float)(s_sum[0] + s_sum[1] + s_sum[2] + s_sum[3]);
    if (tmap && threadIdx.x < 64) {                          // the tile map reads the lengths straight from the int64 vector
        const int lane = threadIdx.x, M = B * S, ntm = (M + rows - 1) / rows;
        int n_real = 0;
        for (int pass = 0; pass < 2; ++pass) {
            int n = 0;
            for (int base = 0; base < ntm; base += 64) {
                const int tm = base + lane;
                bool real = false, valid = tm < ntm;
                if (valid) {
                    const int m0 = tm * rows, mlast = min(m0 + rows - 1, M - 1);
                    const int b0 = m0 / S, b1 = mlast / S;
                    real = !(b0 == b1 && (long)(m0 - b0 * S) >= lens[b0]);
                }
                const bool pick = valid && (pass == 0 ? real : !real);
                const unsigned long long msk = __builtin_amdgcn_ballot_w64(pick);
                const int pos = n + __builtin_popcountll(msk & ((1ull << lane) - 1ull));
                if (pick) tmap[1 + (pass == 0 ? 0 : n_real) + pos] = tm;
                n += __builtin_popcountll(msk);
            }
            if (pass == 0) { n_real = n; if (lane == 0) tmap[0] = n; }
        }
    }
}

extern "C" int fs2_lens_prep(const int64_t* lens, int B, int S, int rows, int32_t* lens32, void* mask, float* count,
                             int32_t* tile_map, hipStream_t stream) {
    FS2_CHECK_ARG(lens && lens32 && mask && count && B > 0 && S > 0 && rows > 0, "lens_prep: bad arguments");
    const long n = (long)B * S;
    lens_prep_kernel<<<1 + (unsigned)((n + 1023) / 1024), 256, 0, stream>>>(lens, B, S, rows, lens32, (unsigned char*)mask, count, tile_map);
    FS2_CHECK_LAUNCH("lens_prep");
    return FS2_OK;
}

extern "C" int fs2_tile_map(const int32_t* lens, int B, int S, int rows, int32_t* out, hipStream_t stream) {
    FS2_CHECK_ARG(lens && out && B > 0 && S > 0 && rows > 0, "tile_map: bad arguments");
    tile_map_kernel<<<1, 64, 0, stream>>>(lens, B * S, S, rows, out);
    FS2_CHECK_LAUNCH("tile_map");
    return FS2_OK;
}

// ------------------------------------------------------------------------------------------------ launcher
static int fs2_cu_count() {
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

template <bool ONE_TAP, bool WIDE, int ABL, int CW>
static void launch_p_cw(const ConvGemmArgs& a, const PSched& sc, hipStream_t stream) {
    constexpr int dyn = PCfg<ONE_TAP, WIDE>::AUX + P_AUX_BYTES;
    static Fs2DevOnce once;
    once.run([&] { (void)hipFuncSetAttribute((const void*)conv_gemm_p_kernel<ONE_TAP, WIDE, ABL, CW>, hipFuncAttributeMaxDynamicSharedMemorySize, dyn); });
    conv_gemm_p_kernel<ONE_TAP, WIDE, ABL, CW><<<(unsigned)sc.G, CW * 64 + 256, dyn, stream>>>(a, sc);
}
// Consumer-wave count per launch (r06j / r06k same-box A/B, profiles/r06j_bench_p_cw.log): eight 32 x 128 waves win where the epilogue
// is a large share of a unit - N >= 1024 forward convolutions, 36 K-steps per 256 x 128 unit: k = 9 FFN forward 227 -> 211 us, the
// encoder's 38.1 -> 35.8 - and lose a little on long reductions (k = 9 data gradient, 144 steps per unit: 195.6 -> 200.9 us);
// PostNet / one-tap shapes are within noise.  The ablations say why it is not more: loaders alone 86 us, MFMAs + loaders 154,
// reads + loaders 121, everything 218 - DMA and MFMA time largely ADD at the per-step barrier, whoever issues the MFMAs.
template <bool ONE_TAP, bool WIDE, int ABL>
static void launch_p(const ConvGemmArgs& a, const PSched& sc, hipStream_t stream) {
    if constexpr (!ONE_TAP && !WIDE) {
        static const int cw_env = fs2_dev_env("FS2_P_CW", 0);
        const int cw = cw_env ? cw_env : (a.N >= 1024 ? 8 : 4);
        if (cw == 8) { launch_p_cw<ONE_TAP, WIDE, ABL, 8>(a, sc, stream); return; }
    }
    launch_p_cw<ONE_TAP, WIDE, ABL, 4>(a, sc, stream);
}

// bytes of tail-split scratch fs2_conv_gemm_tail wants: one f32 256 x 128 tile slab per workgroup of a full-chip launch
extern "C" int fs2_conv_gemm_tail_ws_bytes(void) { return fs2_cu_count() * 256 * 128 * (int)sizeof(float); }

// Eligibility of the persistent kernel (pure function of the launch description; shared with fs2_conv_gemm_variant).
bool fs2_conv_gemm_p_ok(const ConvGemmArgs& a, bool has_map, int dtype, int ks) {
    const bool inact = a.in_act == FS2_ACT_LRELU && a.in_slope > 0.f && a.in_slope < 1.f;
    if (dtype != FS2_BF16 || !(a.in_act == FS2_ACT_NONE || inact)) return false;
    const int taps = a.taps;
    if (!(taps == 1 || (taps >= 3 && (taps - 1) * a.dil <= 64 && taps <= 32))) return false;
    if (a.Cin % 64 != 0 || !a.vec_ok || a.N % 8 != 0) return false;
    if ((double)a.M * a.ldx * 2 >= 2.0e9 || (double)a.N * taps * a.Cin * 2 >= 2.0e9) return false;
    const long tiles = (long)fs2_cdiv(a.M, 256) * fs2_cdiv(a.N, 128) * ks;
    if (ks > 1 && ((a.Cin >> 6) % ks != 0 || a.accumulate)) return false;
    if (a.lens && (!has_map || a.M / a.S > P_MAXB)) return false;
    const int cus = fs2_cu_count();
    // too few tiles to fill the chip: the 128^2 kernels do better - except long convolutions (r02f: the encoder's k=9 data
    // gradient, 48 tiles x 144 K-steps: 100 us here, 115 us on the 128^2 kernel with in-workgroup split-K)
    const bool long_conv = taps >= 3 && (long)taps * (a.Cin >> 6) >= 96 && tiles >= cus / 8;
    if (tiles < (ks > 1 ? cus / 4 : cus / 2) && !long_conv) return false;
    const long G = tiles < cus ? tiles : cus;
    // a workgroup's units travel in two VGPRs (lane k = k-th unit, N-tile in 8 bits): at most 64 units per workgroup under
    // EITHER tile order (per-XCD dealing gives group 0 up to ceil(ntm / 8) * ntn * ks tiles over G / 8 workgroups - one more
    // round than the flat count near the boundary), at most 255 N-tiles; larger launches go to the ring kernel
    const int ntm = fs2_cdiv(a.M, 256), ntn = fs2_cdiv(a.N, 128);
    if (ntn > 255) return false;
    if (p_max_units(ntm, ntn, ks, (int)G, 0) > 64 || (G % 8 == 0 && p_max_units(ntm, ntn, ks, (int)G, 1) > 64)) return false;
    return true;
}

// The real-tile count lives in tile_map[0] on the device; the launch geometry must not depend on it (no host sync), so
// G = min(CUs, all tiles) and workgroups that find no real tile only zero-fill their share of the padded ones.
void fs2_conv_gemm_p_launch(const ConvGemmArgs& a, const int32_t* tile_map, hipStream_t stream, int abl, int ks, float* ws,
                            float* tail_ws) {
    const int ntm = fs2_cdiv(a.M, 256), ntn = fs2_cdiv(a.N, 128);
    const int cus = fs2_cu_count();
    PSched sc;
    sc.ks = ks; sc.nkc_u = (a.Cin >> 6) / ks; sc.ws = ws;
    // tail split: parts of >= 8 K-steps, at most 8 per tile, a power of two that divides the Cin chunks.  Only launches of at
    // most two rounds of tiles (by the static count) with a long reduction (>= 64 K-steps): r02n same-box A/B - the k=9 data
    // gradient (1.2 rounds x 144 steps) 221 -> 193 us, the encoder's (48 tiles) 93.5 -> 44.9 us; launches of 3+ rounds LOSE
    // (k=9 forward, 5.4 rounds: 210 -> 220 us; PostNet k=5 109 -> 113.5): the few workgroups of their last round already run
    // well above the loaded per-tile rate, and the finalize launch costs more than the split saves; short reductions lose too
    // (k=1 FFN forward, 16 steps: 42.7 -> 50.1 us).
    sc.tws = nullptr; sc.tks_max = 1;
    if (tail_ws && ks == 1 && !a.accumulate && (long)ntm * ntn <= 2L * cus && (long)a.taps * sc.nkc_u >= 64) {
        int t = 1;
        while (t < 8 && sc.nkc_u % (2 * t) == 0 && a.taps * (sc.nkc_u / (2 * t)) >= 8) t *= 2;
        static const int tks_env = fs2_dev_env("FS2_P_TKS", 8);          // dev A/B: 1 = tail split off
        if (t > tks_env) t = tks_env;
        if (t >= 2) { sc.tws = tail_ws; sc.tks_max = t; }
    }
    sc.tmap = a.lens ? tile_map : nullptr;
    sc.ntm = ntm; sc.ntn = ntn; sc.b = 0;
    sc.n_real = ntm; sc.n_pad = 0;
    sc.G = (int)((long)ntm * ntn * ks < cus ? (long)ntm * ntn * ks : cus);
    if (sc.tws) sc.G = cus;                                  // a launch with fewer tiles than CUs is all tail
    static const int g_env = fs2_dev_env("FS2_P_G", 0);
    if (g_env > 0 && g_env < sc.G) sc.G = g_env;
    const int taps = a.taps;
    static const int order_env = fs2_dev_env("FS2_P_ORDER", -1);
    // r02e same-box A/B: per-XCD N-fastest is faster wherever an M-tile has >= 2 N-tiles and the launch has at least two
    // rounds of tiles (QKV 51.9 -> 43.8 us, k=1 FFN data gradient 60.1 -> 51.9, k=9 data gradient 218 -> 209, PostNet k=5
    // 113.5 -> 107.6; the N = 1024 forward conv is unchanged, the 192-tile encoder conv 4 % slower)
    sc.order = (ntn >= 2 && sc.G % 8 == 0 && (long)ntm * ntn * ks >= 2L * cus) ? 1 : 0;
    if (order_env >= 0) sc.order = (order_env == 1 && sc.G % 8 == 0) ? 1 : 0;
    (void)abl;
#ifdef FS2_DEV
    switch (abl) {
        case 1: if (taps == 1) launch_p<true, false, 1>(a, sc, stream); else launch_p<false, false, 1>(a, sc, stream); return;
        case 2: if (taps == 1) launch_p<true, false, 2>(a, sc, stream); else launch_p<false, false, 2>(a, sc, stream); return;
        case 3: if (taps == 1) launch_p<true, false, 3>(a, sc, stream); else launch_p<false, false, 3>(a, sc, stream); return;
        case 4: if (taps == 1) launch_p<true, false, 4>(a, sc, stream); else launch_p<false, false, 4>(a, sc, stream); return;
        default: break;
    }
#endif
    const bool wide = taps > 1 && (taps - 1) * a.dil > 16;
    if (a.in_act == FS2_ACT_LRELU) {
        if (taps == 1) launch_p<true, false, 16>(a, sc, stream);
        else if (wide) launch_p<false, true, 16>(a, sc, stream);
        else launch_p<false, false, 16>(a, sc, stream);
    } else if (taps == 1) launch_p<true, false, 0>(a, sc, stream);
    else if (wide) launch_p<false, true, 0>(a, sc, stream);
    else launch_p<false, false, 0>(a, sc, stream);
    if (ks > 1) {
        const long chunks = (long)a.M * (a.N >> 3);
        FS2_ACT_DISPATCH(a.act, (splitk_finalize_kernel<ACT><<<(unsigned)((chunks + 255) / 256), 256, 0, stream>>>(a, ws, ks)));
    }
    if (sc.tws) FS2_ACT_DISPATCH(a.act, (p_tail_finalize_kernel<ACT><<<dim3(4, (unsigned)((sc.G + 1) / 2)), 256, 0, stream>>>(a, sc)));
}
