// fs2_sched.h - the persistent contraction kernel's SCHEDULE and the epilogue staging-row layout as plain host/device
// functions: fs2_gemm_p.hip / fs2_gemm.hip run them on the device, the launcher runs them on the host, and the host-only test
// aid library (tests/aids/fs2_testaid.cpp -> include/fs2hip_testaid.h) exposes the very same code to tests/test_schedule_cpu.py.
// Nothing here touches memory except PSched::tmap (only p_units_load on the device dereferences it).
#pragma once
#include <stdint.h>
#if defined(__HIPCC__)
#define FS2_HD __host__ __device__ __forceinline__
#else
#define FS2_HD inline
#endif

struct PSched {
    int G, b;                  // workgroups in the launch, this workgroup
    int ntm, ntn;              // M-tiles (real + padded), N-tiles
    int n_real, n_pad;         // real / padded M-tiles (read from the tile map on the device)
    const int32_t* tmap;       // [0] = n_real, [1 .. ntm] = real M-tiles then padded M-tiles; null = identity (no lens)
    int ks, nkc_u;             // K-splits per output tile (1 = none) and Cin chunks per unit (= Cin/64/ks).  ks > 1: every unit
                               // stores its partial 256x128 tile into its split's slab of the f32 workspace `ws` (ks x M x N) and
                               // splitk_finalize_kernel sums the slabs into the bf16 output - for few-tile, long-reduction shapes
                               // (the encoder's k=9 data gradient: 48 tiles x 144 K-steps on 256 CUs)
    float* ws;
    float* tws;                // tail slabs (one 256x128 f32 tile per workgroup) or null.  Non-null (and ks == 1): the LAST, partial
    int tks_max;               // round of output tiles is K-split tks (<= tks_max, a power of two) ways so that it takes 1/tks of a
                               // round instead of a whole one: 300 real tiles on 256 CUs ran as 2 rounds of 144 K-steps (the
                               // k=9 data gradient), now 1 round + 44 tiles x 4 parts of 36 steps.  The parts store f32 partial
                               // tiles into their workgroup's slab, p_tail_finalize_kernel sums them and applies the epilogue.
    int order;                 // 0: M-fastest over all workgroups (neighbours share the weight slice);
                               // 1: per XCD, N-fastest: the G/8 workgroups the dispatcher places on one XCD (b % 8) walk ALL
                               //    N-tiles of the same M-tile together, so the activation tile is fetched into that XCD's L2
                               //    once instead of once per N-tile (QKV re-read its input 6x from the Infinity Cache)
};

// Schedule of one workgroup GROUP (order 0: all G workgroups and all tiles; order 1: the G/8 workgroups of one XCD and the
// tiles of the M-tiles mi = x mod 8): R full rounds, then `tail` tiles left over, each split `tks` ways (1 = not split).
struct PPlan { int Gg, j, x, T, R, tail, tks; };
FS2_HD PPlan p_plan(const PSched& s, int b) {
    PPlan p;
    if (s.order == 0) { p.Gg = s.G; p.j = b; p.x = 0; p.T = s.n_real * s.ntn * s.ks; }
    else {
        p.x = b & 7; p.j = b >> 3; p.Gg = s.G >> 3;
        const int nx = s.n_real > p.x ? (s.n_real - p.x + 7) >> 3 : 0;
        p.T = nx * s.ntn * s.ks;
    }
    p.R = p.T / p.Gg;
    p.tail = p.T - p.R * p.Gg;
    p.tks = 1;
    if (s.tws && p.tail > 0) {
        int t = 1;
        while (2 * t <= s.tks_max && 2 * t * p.tail <= p.Gg) t *= 2;
        p.tks = t;
    }
    return p;
}
// position in the group's unit list -> (index into the real-M-tile list, N-tile, uniform K-split index)
FS2_HD void p_pos(const PSched& s, const PPlan& p, int pos, int& mi, int& nt, int& sp) {
    if (s.order == 0) {
        const int rest = pos / s.ks;
        sp = pos - rest * s.ks;
        nt = rest / s.n_real;
        mi = rest - nt * s.n_real;
    } else {
        const int per = s.ntn * s.ks;
        const int mil = pos / per, r = pos - mil * per;
        nt = r / s.ks;
        sp = r - nt * s.ks;
        mi = p.x + 8 * mil;
    }
}
// k-th unit of workgroup s.b: index into the real-M-tile list, N-tile, first Cin chunk and chunk count, tail part count (1 =
// a whole tile); false when the workgroup has no k-th unit
FS2_HD bool p_unit(const PSched& s, int k, int& mi, int& nt, int& kc0, int& nkc, int& np) {
    const PPlan p = p_plan(s, s.b);
    int pos, part = 0;
    np = 1;
    if (k < p.R) pos = k * p.Gg + p.j;
    else if (k == p.R && p.j < p.tail * p.tks) { pos = p.R * p.Gg + p.j / p.tks; part = p.j % p.tks; np = p.tks; }
    else return false;
    int sp;
    p_pos(s, p, pos, mi, nt, sp);
    nkc = s.nkc_u / np;
    kc0 = sp * s.nkc_u + part * nkc;
    return true;
}

// most units any workgroup of the launch can hold (R full rounds + one tail unit) - the launcher's bound for the two-VGPR unit
// table (lane k = k-th unit, 64 lanes).  order 1 deals per XCD: group x owns the M-tiles mi = x (mod 8), so its tile count is
// ceil((n_real - x) / 8) * ntn * ks over G / 8 workgroups - group 0 is the largest.
FS2_HD int p_max_units(int n_real, int ntn, int ks, int G, int order) {
    long T, Gg;
    if (order == 0) { T = (long)n_real * ntn * ks; Gg = G; }
    else { T = (long)((n_real + 7) >> 3) * ntn * ks; Gg = G >> 3; }
    if (Gg <= 0) return 1 << 30;
    return (int)((T + Gg - 1) / Gg);
}

// position of logical column c (0..127) in a 128-float epilogue staging row (see fs2_gemm.hip, gemm_store_tile): bf16 outputs
// store the row GRANULE-PERMUTED (granule = 4 floats): even granule 2i at position i, odd granule 2i+1 at 16 + ((i + 4) & 15);
// fp32 outputs keep the natural order.
FS2_HD int fs2_tile_col128_bytes(int elem_bytes, int c) {
    if (elem_bytes != 2) return c;
    const int g = c >> 2, i = g >> 1;
    return (((g & 1) ? 16 + ((i + 4) & 15) : i) << 2) | (c & 3);
}
