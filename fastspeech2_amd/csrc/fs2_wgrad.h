// fs2_wgrad.h - launch description and split-K plan of the bf16 weight-gradient kernels, shared by fs2_gemm.hip (register-staged
// kernels, C entry points) and fs2_wgrad.hip (LDS-DMA tap-group kernel).
#pragma once
#include "fs2_common.h"

struct WgradArgs {
    const void* dY; long lddy;
    const void* X; long ldx;
    float* dW;
    float* dbias;          // optional: dbias[n] += sum_m dY[m][n], fused into the blocks that own (c-tile 0, tap group 0)
    const int32_t* lens;   // optional: dY rows t >= lens[seq] are known to be zero -> their K-tiles are skipped
    int M, N, Cin, S, taps, dil, pad, rows_per_split, g3, dbg;
    int n_tiles, n_splits, per_xcd;   // fs2_wgrad.hip's 1-D grid (XCD-aware placement of (split, tile) pairs; see conv_wgrad_tg_kernel)
    float* slab;           // optional split-K scratch: split s stores its partial tile (plain stores) at slab + s * slab_stride in the
    long slab_stride;      // layout of dW, its bias partials behind it (slab_stride = N*taps*Cin + N floats); wgrad_finalize_kernel
                           // sums the splits into dW / dbias.  null: fp32 atomics straight into dW (round-1/2 path)
};

// Split-K plan of a bf16 weight-gradient launch (pure host function: fs2_conv_wgrad_ws_bytes and the launcher must agree).
//   tiles = 128 x 128 output tiles x tap groups; units = 64-row K-tiles (never straddling a sequence)
// With a slab workspace a split costs one plain store of its tile + its share of the finalize pass (no atomics), so the depth is
// chosen to fill the chip ONCE (~1 workgroup per CU) with at least 8 K-tiles per workgroup; the slabs are capped at 96 MB.
#define FS2_WGRAD_WS_CAP (96L << 20)        /* bytes: exported as fs2_conv_wgrad_ws_cap() - the one place the cap is written */
struct WgradPlan { int tiles, units, ups, splits, g_first, n_first, n_rest, share; };
static inline WgradPlan wgrad_plan(int M, int N, int Cin, int S, int taps, int dil, bool has_lens, bool slab) {
    WgradPlan p = {};
    const int S_eff = (taps == 1) ? ((has_lens && M % S == 0) ? S : M) : S;
    p.share = (slab && taps >= 2 && dil == 1) ? 1 : 0;      // the LDS-DMA tap-group kernel (fs2_wgrad.hip)
    if (taps == 1) { p.g_first = 1; p.n_first = 1; p.n_rest = 0; }
    else if (p.share) {                                   // groups of up to 5 taps: 9 = 5 + 4, 5 = 5, 3 = 3, 7 = 4 + 3 ...
        const int ng = (taps + 4) / 5;
        p.n_first = (taps + ng - 1) / ng;                 // 9 -> 5, 7 -> 4, 5 -> 5
        p.g_first = taps / p.n_first;
        p.n_rest = taps - p.g_first * p.n_first;
    } else { p.n_first = 3; p.g_first = taps / 3; p.n_rest = taps - 3 * p.g_first; }
    const int groups = p.g_first + (p.n_rest ? 1 : 0);
    p.tiles = fs2_cdiv(N, 128) * fs2_cdiv(Cin, 128) * groups;
    p.units = (M / S_eff) * ((S_eff + 63) / 64);
    if (slab) {
        // workgroups to aim for: 192, not one per CU - these launches run on the side stream NEXT TO the data-gradient chain, whose
        // persistent kernels need whole CUs (r03d same-box sweep of the whole step, two rounds: 64: 10.56 ms, 96: 9.55, 128: 9.31,
        // 160: 9.23, 192: 9.22, 256: 9.27, 384 .. 1024: 9.41 - 9.45; the round-2 atomic kernels: 9.44; no weight gradients: 7.88)
        static const int cus = fs2_dev_env("FS2_WGRAD_TG_WGS", 192);
        long want = p.tiles >= cus ? 1 : (cus + p.tiles / 2) / p.tiles;       // round(cus / tiles)
        const long max_by_units = p.units / 8 > 0 ? p.units / 8 : 1;
        if (want > max_by_units) want = max_by_units;
        const long slab_bytes = ((long)N * taps * Cin + N) * 4;
        const long max_by_ws = FS2_WGRAD_WS_CAP / slab_bytes > 0 ? FS2_WGRAD_WS_CAP / slab_bytes : 1;
        if (want > max_by_ws) want = max_by_ws;
        if (want < 1) want = 1;
        p.ups = (int)((p.units + want - 1) / want);
        p.splits = fs2_cdiv(p.units, p.ups);
    }
    return p;
}


// fs2_wgrad.hip: the LDS-DMA tap-group kernel for plan p (p.share): false when no instantiation covers (p.n_first, p.n_rest)
bool fs2_wgrad_tg_launch(WgradArgs a, const WgradPlan& p, hipStream_t stream);
