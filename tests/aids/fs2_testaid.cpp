// Host-only test aids over the kernels' own schedule / layout source (see include/fs2hip_testaid.h).  Built with g++.
#include "fs2hip_testaid.h"
#include "fs2_sched.h"

extern "C" int fs2t_stage_tile_col(int c, int elem_bytes) {
    if (c < 0 || c >= 128 || (elem_bytes != 2 && elem_bytes != 4)) return -1;
    return fs2_tile_col128_bytes(elem_bytes, c);
}

extern "C" int fs2t_conv_gemm_p_units(int n_real, int ntn, int G, int order, int ks, int nkc, int tks_max, int b, int* out, int max_units) {
    if (!out || n_real < 0 || ntn <= 0 || G <= 0 || ks <= 0 || nkc <= 0 || nkc % ks != 0 || b < 0 || b >= G || (order != 0 && G % 8 != 0)) return -1;
    PSched s = {};
    s.G = G; s.b = b; s.ntm = n_real; s.ntn = ntn; s.n_real = n_real; s.n_pad = 0; s.tmap = nullptr;
    s.ks = ks; s.nkc_u = nkc / ks; s.ws = nullptr; s.order = order;
    s.tws = tks_max > 1 ? reinterpret_cast<float*>(16) : nullptr; s.tks_max = tks_max;
    int n = 0;
    for (int k = 0;; ++k) {
        int mi, nt, kc0, nk, np;
        if (!p_unit(s, k, mi, nt, kc0, nk, np)) break;
        if (n < max_units) { out[5 * n] = mi; out[5 * n + 1] = nt; out[5 * n + 2] = kc0; out[5 * n + 3] = nk; out[5 * n + 4] = np; }
        ++n;
    }
    return n;
}

extern "C" int fs2t_conv_gemm_p_max_units(int n_real, int ntn, int ks, int G, int order) { return p_max_units(n_real, ntn, ks, G, order); }
