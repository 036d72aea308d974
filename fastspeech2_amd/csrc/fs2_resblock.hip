// fs2_resblock.hip - a whole HiFi-GAN ResBlock1 in ONE launch for the narrow stages (C = 32 / 64 channels, bf16; gfx950, round 5).
//
//   y = x;  for m in 0..2:  t = lrelu(conv1_m(lrelu(y), dilation d_m));  y = conv2_m(t) + y          (hifigan/models.py:96-103)
//   xs = (accumulate ? xs : 0) + out_scale * y                                                           (models.py:155-160: xs / 3)
//
// Why: after the second up-sampling stage the residual blocks are HBM-bound one convolution at a time (C = 32, k = 3: 48 FLOP per
// byte): conv_skinny_kernel moves input + residual + output of EVERY convolution through HBM - 18 row passes per block, 3.2 ms
// of a 9.6 ms batch-synthesis step at ~3.9 TB/s (profiles/r04zzz_kernel_trace_synth.md).  Here a workgroup owns a tile of E
// consecutive rows of one utterance and runs all six convolutions on it: x is read once, xs is read + written once, 3 passes.
//   * the running sum y lives in fp32 REGISTERS of the wave that owns the rows (MFMA accumulators, transposed: D[cout][row], so a
//     lane owns a row and, after v_permlane32_swap, runs of 8 consecutive channels) - the residual add is the accumulator's
//     initial value, and y is never rounded between the three pairs;
//   * what a convolution READS (lrelu(y) resp. t) is written as bf16 into ONE LDS tile [GUARD + E + GUARD rows][C + 8] (rows padded
//     by one 16-byte chunk: ds_read_b128 by 16 consecutive rows is conflict-free); a tap is a row shift of the fragment address;
//     rows outside the utterance are written as zeros (= the reference's zero padding of every convolution's input);
//   * halo by recomputation: the tile covers H = (k-1)/2 * (d0 + d1 + d2 + 3) more rows on each side than it outputs; whatever is
//     computed from outside the tile stays inside that margin (it moves inward by one convolution's reach per convolution);
//   * weights stream L2 -> LDS by LDS-DMA (global_load_lds_dwordx4, source-side swizzle) in groups (RbCfg) through a ring of 2 - 4
//     slots with all but one slot in flight while the current group is multiplied; one raw barrier per group.
// Measured (profiles/r05b-r05j): batch synthesis 9.55 -> 8.13 ms per step with one launch per block, -> 7.47 ms with the three blocks of a
// stage in one launch (RTF 2.29e-4 -> 1.80e-4); MFMA busy 39 % (C = 64) / 33 % (C = 32), 5 - 9 VALU per MFMA - most of them the
// epilogues (leaky-ReLU, pack, store of every convolution's output: a k = 3 convolution at C = 32 is 24 MFMAs and 64 outputs per lane).
// Tried on top and measured neutral or worse: fragments requested one k-slice ahead through two register sets (7.47 -> 7.75 ms: 241
// registers), a 4-deep weight ring with three groups in flight instead of one (no change: the stream is not the limiter), padded
// rows + precomputed fragment addresses (LDS bank conflicts 10-15 % -> 0, VALU per MFMA -1, time unchanged: kept, it is the simpler
// addressing).
// The first attempt at this fusion (round 1: conv1 -> conv2 pair, weight groups behind barrier PAIRS, register-staged) was 24 %
// slower than two launches; this one has no staging registers, no per-group barrier pair and keeps y out of LDS altogether.
#include "fs2_gemm.h"

typedef unsigned rb_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned rb_u32x2 __attribute__((ext_vector_type(2)));

struct ResBlockArgs {
    const void* X; long ldx;            // [M][C] bf16 (M = B * S rows, time-major)
    // up to three blocks on the same x (the kernel sizes 3 / 7 / 11 of one up-sampling stage, models.py:155-160), same dilations
    const void* W1[3]; const void* W2[3];   // per block: [3][C][k][C] bf16 (conv index, cout, tap, cin)
    const float* B1[3]; const float* B2[3]; // per block: [3][C] f32
    int k[3], nblk;
    void* XS; long ldxs;                // [M][C] bf16
    int accumulate; float out_scale, slope, post_slope;
    int S, nbatch, d[3], R, tiles_per_seq;
};

template <int C> struct RbCfg {
    static constexpr int E = C == 32 ? 1024 : (C == 64 ? 512 : 256);      // tile rows (8 waves x MB x 32): the LDS left beside the weight slots
    static constexpr int GUARD = 26;                         // rows in front of / behind the tile a shifted read may touch (|shift| <= 25)
    // activation rows are PADDED by one 16-byte chunk instead of XOR-swizzled: the row stride is an odd number of chunks, so the 16
    // rows of a ds_read_b128 lane group land on 16 different 16-byte slots, and a fragment address is base(row) + tap shift (a scalar)
    // + chunk (an immediate) - the XOR form cost 2-3 VALU per read next to every MFMA (PMC r05i: 6.4 - 10.3 VALU per MFMA, 33 - 38 %
    // MFMA busy: issue-bound)
    static constexpr int STRIDE = C * 2 + 16;
    static constexpr int MB = E / 256, NB = C / 32, KS = C / 16, CPR = C / 8;
    static constexpr int ROWB = C * 2;
    static constexpr int TAPB = C * C * 2;                   // bytes of one tap's weights
    // A weight GROUP (one ring slot, one barrier) = TG taps x 1 / KP of the reduction: C = 32 a whole convolution (<= 11 taps, 22 KB),
    // C = 64 two taps (16 KB), C = 128 half a tap (channels 64 kp .. + 64 of all 128 couts: 16 KB).  NSLOT slots, NSLOT - 1 groups in
    // flight: the L2 -> LDS stream is latency-bound (~2 us issue -> landed), so what counts is bytes in flight per CU - the first
    // C = 128 cut (whole taps, 2 slots, one group in flight) ran 2.7 us per tap for 1 us of MFMAs (profiles/r05g_kernel_trace_synth.md).
    static constexpr int TG = C == 32 ? 11 : (C == 64 ? 2 : 1);
    static constexpr int KP = C == 128 ? 2 : 1;
    static constexpr int NSLOT = C == 32 ? 2 : 4;
    static constexpr int KSG = KS / KP;                      // k-slices per group
    static constexpr int WROWB = ROWB / KP;                  // bytes of a cout's row inside a slot (C = 128: 128)
    static constexpr int SLOT = TG * TAPB / KP;
    static constexpr int PPG = SLOT / 1024;                  // 1 KiB DMA pieces of a full group (22 / 16 / 16)
    static constexpr int PW = C == 32 ? 0 : PPG / 8;         // pieces per wave and group when uniform (C >= 64: 2); 0 = counted by vmcnt(0)
    static constexpr int ACT_BYTES = ((E + 2 * GUARD) * STRIDE + 1023) / 1024 * 1024;
    static constexpr int W_OFF = ACT_BYTES;
    static constexpr int BIAS_OFF = W_OFF + NSLOT * SLOT;
    static constexpr int LDS = BIAS_OFF + 18 * C * 4;
};
// swizzle key of LDS row r: chunk c is stored at chunk c ^ key (16 consecutive rows x one chunk -> 16 distinct 16-byte slots of the
// 256-byte bank window: 4 rows per window at C = 32, 2 at C = 64, 1 at C = 128)
template <int C> __device__ __forceinline__ unsigned rb_key(unsigned r) {
    return C == 32 ? ((r >> 2) & 3u) : (C == 64 ? ((r >> 1) & 7u) : (r & 15u));      // (C = 128: a row IS a bank window)
}
// ... and of a cout's row inside a weight slot (C = 128: half rows of 128 bytes, as C = 64)
template <int C> __device__ __forceinline__ unsigned rb_wkey(unsigned r) { return C == 32 ? ((r >> 2) & 3u) : ((r >> 1) & 7u); }

// raw barrier with this wave's LDS stores (and DMA pieces) retired first; the memory clobber keeps the compiler's LDS accesses on their side
__device__ __forceinline__ void rb_barrier_all() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
template <int N> __device__ __forceinline__ void rb_barrier_vm() { asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" :: "n"(N) : "memory"); }
__device__ __forceinline__ void rb_barrier_lds() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int C>
__global__ void __launch_bounds__(512, 2) resblock_fused_kernel(ResBlockArgs a) {
    typedef RbCfg<C> K;
    constexpr int E = K::E, GUARD = K::GUARD, MB = K::MB, NB = K::NB, STRIDE = K::STRIDE, TG = K::TG, KP = K::KP, NSLOT = K::NSLOT;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, fl = lane & 31, fh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned smem_u = lds_addr(smem);
    float* bias_s = reinterpret_cast<float*>(smem + K::BIAS_OFF);

    const int seq = blockIdx.x / a.tiles_per_seq, tile = blockIdx.x - seq * a.tiles_per_seq;
    const int H = (E - a.R) >> 1;
    const int t_first = tile * a.R - H;                      // utterance row of tile row 0
    const size_t seq_row0 = (size_t)seq * a.S;
    // ---- weight stream over all blocks: group gi -> slot gi % NSLOT.  Block j: convolution c = 2m (conv1 of pair m) / 2m + 1 (conv2), each
    // in gpc_j = ceil(k_j / TG) * KP groups
    int gpc[3], gstart[4];
    gstart[0] = 0;
#pragma unroll
    for (int j = 0; j < 3; ++j) { gpc[j] = j < a.nblk ? ((a.k[j] + TG - 1) / TG) * KP : 0; gstart[j + 1] = gstart[j] + 6 * gpc[j]; }
    const int ngroups = gstart[3];
    auto issue_group = [&](int gi) {
        const int j = gi >= gstart[2] ? 2 : (gi >= gstart[1] ? 1 : 0);
        const int kj = a.k[j], gl = gi - gstart[j];
        const int c = gl / gpc[j], g = gl - c * gpc[j];
        const int tap0 = (g / KP) * TG, kp = g % KP, nt = min(TG, kj - tap0);
        const unsigned char* wbase = reinterpret_cast<const unsigned char*>((c & 1) ? a.W2[j] : a.W1[j]) + (size_t)(c >> 1) * C * kj * C * 2;
        const unsigned slot = smem_u + K::W_OFF + (unsigned)((gi % NSLOT) * K::SLOT);
        // a 1 KiB piece = RPP couts of one tap (rows of WROWB bytes); lane -> (cout, LDS chunk position); the position holds the global
        // chunk kp * chunks-per-part + (pos ^ key(cout)).  C >= 64: every wave issues exactly PW pieces per group (a group of fewer
        // taps re-fetches its last tap into the unused part of the slot: the count a wave waits with stays a compile-time constant)
        constexpr int RPP = 1024 / K::WROWB, CPP = K::WROWB / 16, PPT = K::TAPB / KP / 1024;
        const int npieces = K::PW ? K::PPG : nt * PPT;
        for (int q = wave; q < npieces; q += 8) {
            const int tl = q / PPT, p = q - tl * PPT;
            const int tsrc = min(tl, nt - 1);
            const int cout = p * RPP + lane / CPP;
            const unsigned pos = (unsigned)(lane % CPP);
            const unsigned gch = (unsigned)(kp * CPP) + (pos ^ rb_wkey<C>((unsigned)cout));
            const unsigned voff = (unsigned)(((cout * kj + tap0 + tsrc) * C) * 2) + (gch << 4);
            glds16_sbase(voff, wbase, __builtin_amdgcn_readfirstlane(slot + (unsigned)(tl * (K::TAPB / KP) + p * 1024)));
        }
    };
#pragma unroll
    for (int i = 0; i < NSLOT - 1; ++i)
        if (i < ngroups) issue_group(i);
    for (int i = tid; i < a.nblk * 6 * C; i += 512) {
        const int j = i / (6 * C), r = i - j * 6 * C, c = r / C, n = r - c * C;
        bias_s[i] = ((c & 1) ? a.B2[j] : a.B1[j])[(c >> 1) * C + n];
    }

    // rows of this lane inside the utterance? (bit mb)
    unsigned in_seq = 0;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
        const int t = t_first + wave * (MB * 32) + mb * 32 + fl;
        if (t >= 0 && t < a.S) in_seq |= 1u << mb;
    }
    const bf16_t* X = reinterpret_cast<const bf16_t*>(a.X);
    bf16_t* XS = reinterpret_cast<bf16_t*>(a.XS);
    // this lane's rows in the accumulator layout (lane = row, registers = couts 4 fh + (r & 3) + 8 (r >> 2)): 8-byte pieces
    auto load_rows = [&](const bf16_t* base, long ld, f32x16 (&v)[MB][NB]) {
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
            const int t = min(max(t_first + wave * (MB * 32) + mb * 32 + fl, 0), a.S - 1);
            const bf16_t* row = base + (seq_row0 + (size_t)t) * ld;
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 x4 = ld4<bf16_t>(row + nb * 32 + 8 * q + 4 * fh);
                    v[mb][nb][4 * q + 0] = x4.x; v[mb][nb][4 * q + 1] = x4.y; v[mb][nb][4 * q + 2] = x4.z; v[mb][nb][4 * q + 3] = x4.w;
                }
        }
    };
    // accumulator layout -> a lane's 2 x 8 consecutive couts (chunks nb * 4 + ch * 2 + fh) of its row
    auto to_rows = [&](const f32x16& v, float (&c)[2][8]) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            rb_u32x2 s0 = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[e]), __float_as_uint(v[4 + e]), false, false);
            rb_u32x2 s1 = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[8 + e]), __float_as_uint(v[12 + e]), false, false);
            c[0][e] = __uint_as_float(s0[0]); c[0][4 + e] = __uint_as_float(s0[1]);
            c[1][e] = __uint_as_float(s1[0]); c[1][4 + e] = __uint_as_float(s1[1]);
        }
    };
    // LDS row of tile row tr: GUARD + tr.  Write lrelu(v) (bf16, zeros outside the utterance) for this wave's rows.
    auto write_act = [&](f32x16 (&v)[MB][NB]) {
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
            const unsigned lr = (unsigned)(GUARD + wave * (MB * 32) + mb * 32 + fl);
            const unsigned rowaddr = lr * STRIDE;
            const bool live = (in_seq >> mb) & 1u;
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                float c[2][8];
                to_rows(v[mb][nb], c);
#pragma unroll
                for (int ch = 0; ch < 2; ++ch) {
                    uint4 o;
                    uint32_t* ou = reinterpret_cast<uint32_t*>(&o);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float lo = c[ch][2 * e], hi = c[ch][2 * e + 1];
                        lo = fmaxf(lo, lo * a.slope); hi = fmaxf(hi, hi * a.slope);         // 0 < slope < 1
                        ou[e] = live ? pack_bf16x2(lo, hi) : 0u;
                    }
                    const unsigned chunk = (unsigned)(nb * 4 + ch * 2 + fh);
                    *reinterpret_cast<uint4*>(smem + rowaddr + (chunk << 4)) = o;
                }
            }
        }
    };

    f32x16 y[MB][NB], t_acc[MB][NB];
    uint32_t xs_pk[MB][NB][8];                              // the running xs, bf16 pairs of accumulator registers (2i, 2i + 1)
    if (a.accumulate) {
        load_rows(XS, a.ldxs, t_acc);
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int i = 0; i < 8; ++i) xs_pk[mb][nb][i] = pack_bf16x2(t_acc[mb][nb][2 * i], t_acc[mb][nb][2 * i + 1]);
    } else {
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int i = 0; i < 8; ++i) xs_pk[mb][nb][i] = 0u;
    }
    // loop-invariant parts of the fragment addresses: weights (XOR-swizzled slot image: one address per cout block and k-slice) and this
    // lane's activation rows (padded rows: one base per row block; tap shift and k-slice come in as a scalar and an immediate)
    unsigned wa[NB][K::KSG], xbase[MB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int ks = 0; ks < K::KSG; ++ks)
            wa[nb][ks] = (unsigned)((nb * 32 + fl) * K::WROWB) + ((((unsigned)(2 * ks + fh)) ^ rb_wkey<C>((unsigned)(nb * 32 + fl))) << 4);
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) xbase[mb] = (unsigned)((GUARD + wave * (MB * 32) + mb * 32 + fl) * STRIDE + fh * 16);
    int gi = 0;
    // the groups of convolution c of block j multiplied into `acc` (y for conv2: the residual is already in it; t_acc for conv1)
    auto run_conv = [&](f32x16 (&acc)[MB][NB], int j, int c, bool add_bias) {
        const int kj = a.k[j], pad2 = (kj - 1) >> 1;
        const int dil = (c & 1) ? 1 : a.d[c >> 1];
        const int pad = pad2 * dil;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 bv = *reinterpret_cast<const float4*>(bias_s + (j * 6 + c) * C + nb * 32 + 8 * q + 4 * fh);
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) {
                    if (add_bias) { acc[mb][nb][4 * q] += bv.x; acc[mb][nb][4 * q + 1] += bv.y; acc[mb][nb][4 * q + 2] += bv.z; acc[mb][nb][4 * q + 3] += bv.w; }
                    else { acc[mb][nb][4 * q] = bv.x; acc[mb][nb][4 * q + 1] = bv.y; acc[mb][nb][4 * q + 2] = bv.z; acc[mb][nb][4 * q + 3] = bv.w; }
                }
            }
        for (int g = 0; g < gpc[j]; ++g, ++gi) {
            // my pieces of group gi have landed (only the NSLOT - 2 groups behind it may still be in flight) and my tile rows are stored;
            // then: group gi + the activation tile visible to all, slot (gi - 1) % NSLOT free -> group gi + NSLOT - 1 goes there
            if (K::PW == 0) rb_barrier_all();
            else {
                const int younger = min(NSLOT - 2, ngroups - 1 - gi);
                if (younger >= 2) rb_barrier_vm<2 * (K::PW ? K::PW : 1)>();
                else if (younger == 1) rb_barrier_vm<(K::PW ? K::PW : 1)>();
                else rb_barrier_all();
            }
            if (gi + NSLOT - 1 < ngroups) issue_group(gi + NSLOT - 1);
            const int tap0 = (g / KP) * TG, kp = g % KP, nt = min(TG, kj - tap0);
            const unsigned wslot = (unsigned)(K::W_OFF + (gi % NSLOT) * K::SLOT);
            for (int tl = 0; tl < nt; ++tl) {
                const int shift = (tap0 + tl) * dil - pad;
                const int xoff = shift * STRIDE + kp * (K::KSG * 32);                        // (scalar)
                const unsigned woff = wslot + (unsigned)(tl * (K::TAPB / KP));             // (scalar)
                const unsigned char* xp[MB];
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) xp[mb] = smem + (int)xbase[mb] + xoff;
#pragma unroll
                for (int ks = 0; ks < K::KSG; ++ks) {
                    rb_u32x4 wf[NB], xf[MB];
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) wf[nb] = *reinterpret_cast<const rb_u32x4*>(smem + woff + wa[nb][ks]);
#pragma unroll
                    for (int mb = 0; mb < MB; ++mb) xf[mb] = *reinterpret_cast<const rb_u32x4*>(xp[mb] + ks * 32);
#pragma unroll
                    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb)
                            acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wf[nb]), __builtin_bit_cast(bf16x8, xf[mb]),
                                                                                  acc[mb][nb], 0, 0, 0);
                }
            }
        }
    };
    for (int j = 0; j < a.nblk; ++j) {
        load_rows(X, a.ldx, y);                // y <- x (blocks after the first: from L2)
        // every wave has finished the previous block's last convolution; first block: the bias table (written by whichever threads own
        // its entries) is complete before any wave's first convolution reads it - run_conv reads the biases BEFORE its first group
        // barrier, and without this one a fast wave could pick up what the previous launch left in LDS (seen once in ~1000 launches as
        // an error of the size of a bias: profiles/r05zzz_pytest_flake.log)
        rb_barrier_lds();
        write_act(y);
        for (int m = 0; m < 3; ++m) {
            run_conv(t_acc, j, 2 * m, false);
            rb_barrier_lds();                  // every wave has read lrelu(y): the tile may be overwritten
            write_act(t_acc);
            run_conv(y, j, 2 * m + 1, true);
            if (m == 2) break;
            rb_barrier_lds();
            write_act(y);
        }
        // xs <- bf16(xs + out_scale * y): rounded after every block, as the chain of launches stored it
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const uint32_t u = xs_pk[mb][nb][i];
                    xs_pk[mb][nb][i] = pack_bf16x2(__uint_as_float(u << 16) + a.out_scale * y[mb][nb][2 * i],
                                                   __uint_as_float(u & 0xffff0000u) + a.out_scale * y[mb][nb][2 * i + 1]);
                }
    }

    // ---- store xs for the tile's central R rows
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
        const int tr = wave * (MB * 32) + mb * 32 + fl;
        const int t = t_first + tr;
        const bool store = tr >= H && tr < H + a.R && t < a.S;       // (t >= 0 follows from tr >= H)
        bf16_t* row = XS + (seq_row0 + (size_t)(store ? t : 0)) * a.ldxs;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            f32x16 v;
#pragma unroll
            for (int i = 0; i < 8; ++i) { v[2 * i] = __uint_as_float(xs_pk[mb][nb][i] << 16); v[2 * i + 1] = __uint_as_float(xs_pk[mb][nb][i] & 0xffff0000u); }
            if (a.post_slope > 0.f) {
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = v[r] > 0.f ? v[r] : v[r] * a.post_slope;
            }
            float cc[2][8];
            to_rows(v, cc);
            if (store) {
#pragma unroll
                for (int ch = 0; ch < 2; ++ch) {
                    uint4 o;
                    uint32_t* ou = reinterpret_cast<uint32_t*>(&o);
#pragma unroll
                    for (int e = 0; e < 4; ++e) ou[e] = pack_bf16x2(cc[ch][2 * e], cc[ch][2 * e + 1]);       // (exact without post_slope: the values are bf16 already)
                    *reinterpret_cast<uint4*>(row + nb * 32 + ch * 16 + fh * 8) = o;
                }
            }
        }
    }
}

template <int C>
static void launch_resblock(const ResBlockArgs& a, hipStream_t stream) {
    static Fs2DevOnce once;
    once.run([&] { (void)hipFuncSetAttribute((const void*)resblock_fused_kernel<C>, hipFuncAttributeMaxDynamicSharedMemorySize, RbCfg<C>::LDS); });
    resblock_fused_kernel<C><<<(unsigned)(a.nbatch * a.tiles_per_seq), 512, RbCfg<C>::LDS, stream>>>(a);
}

static int rb_halo(int k, int d0, int d1, int d2) { return ((k - 1) / 2) * (d0 + d1 + d2 + 3); }

extern "C" int fs2_resblock_supported(int C, int k, int d0, int d1, int d2, int dtype) {
    // C = 128 has a configuration above (256-row tiles, half-tap weight groups) and was measured: parity green, 35 % MFMA busy, and NO
    // faster than the chain of single convolutions (stage 1 of HiFi-GAN V1: 2.8 vs 2.4 ms isolated, the batch-synthesis step 7.57 vs
    // 7.59 ms - profiles/r05g-r05j): 256-row tiles pay 10 / 39 / 88 % halo recomputation at k = 3 / 7 / 11 and stream 32 KB of weights
    // per tap for 256 rows.  Not instantiated.
    if (dtype != FS2_BF16 || (C != 32 && C != 64) || k < 1 || k > 11 || !(k & 1)) return 0;
    if (d0 < 1 || d1 < 1 || d2 < 1) return 0;
    const int pad2 = (k - 1) / 2;
    // a shifted read stays inside the guard rows: |shift| <= GUARD - 1 (the bound is the tile's own constant, so the two cannot drift)
    if (pad2 * (d0 > d1 ? (d0 > d2 ? d0 : d2) : (d1 > d2 ? d1 : d2)) > RbCfg<32>::GUARD - 1) return 0;
    const int E = C == 32 ? 1024 : 512;
    return E - 2 * rb_halo(k, d0, d1, d2) >= 64 ? 1 : 0;
}

static int resblocks_impl(const void* x, long ldx, int nblk, const void* const* w1, const void* const* w2, const float* const* b1,
                          const float* const* b2, const int* k, void* xs, long ldxs, int accumulate, float out_scale, float slope,
                          float post_slope, int B, int S, int C, int d0, int d1, int d2, int dtype, hipStream_t stream) {
    FS2_CHECK_ARG(x && xs && nblk >= 1 && nblk <= 3, "resblock_fwd: null pointer / block count");
    FS2_CHECK_ARG(B > 0 && S > 0, "resblock_fwd: bad shape");
    FS2_CHECK_ARG(ldx % 8 == 0 && ldxs % 8 == 0 && (((uintptr_t)x | (uintptr_t)xs) & 15) == 0, "resblock_fwd: rows must be 16-byte addressable");
    FS2_CHECK_ARG(slope > 0.f && slope < 1.f && post_slope >= 0.f && post_slope < 1.f, "resblock_fwd: leaky-ReLU slope in (0, 1)");
    ResBlockArgs a = {};
    int H = 0;
    for (int j = 0; j < nblk; ++j) {
        FS2_CHECK_ARG(w1[j] && w2[j] && b1[j] && b2[j] && (((uintptr_t)w1[j] | (uintptr_t)w2[j]) & 15) == 0, "resblock_fwd: weights");
        FS2_CHECK_ARG(fs2_resblock_supported(C, k[j], d0, d1, d2, dtype), "resblock_fwd: unsupported (C in {32, 64}, odd k <= 11, bf16)");
        a.W1[j] = w1[j]; a.W2[j] = w2[j]; a.B1[j] = b1[j]; a.B2[j] = b2[j]; a.k[j] = k[j];
        const int h = rb_halo(k[j], d0, d1, d2);
        H = h > H ? h : H;                               // one tile geometry for all blocks: the widest halo
    }
    a.nblk = nblk;
    a.X = x; a.ldx = ldx; a.XS = xs; a.ldxs = ldxs;
    a.accumulate = accumulate; a.out_scale = out_scale; a.slope = slope; a.post_slope = post_slope; a.S = S; a.nbatch = B;
    a.d[0] = d0; a.d[1] = d1; a.d[2] = d2;
    const int E = C == 32 ? 1024 : 512;
    a.R = E - 2 * H;
    a.tiles_per_seq = fs2_cdiv(S, a.R);
    if (C == 32) launch_resblock<32>(a, stream);
    else launch_resblock<64>(a, stream);
    FS2_CHECK_LAUNCH("resblock_fwd");
    return FS2_OK;
}

extern "C" int fs2_resblock_fwd(const void* x, long ldx, const void* w1, const void* w2, const float* b1, const float* b2, void* xs,
                                long ldxs, int accumulate, float out_scale, float slope, float post_slope, int B, int S, int C, int k,
                                int d0, int d1, int d2, int dtype, hipStream_t stream) {
    return resblocks_impl(x, ldx, 1, &w1, &w2, &b1, &b2, &k, xs, ldxs, accumulate, out_scale, slope, post_slope, B, S, C, d0, d1, d2, dtype,
                          stream);
}

extern "C" int fs2_resstage_fwd(const void* x, long ldx, const void* w1a, const void* w2a, const float* b1a, const float* b2a, int ka,
                                const void* w1b, const void* w2b, const float* b1b, const float* b2b, int kb, const void* w1c,
                                const void* w2c, const float* b1c, const float* b2c, int kc, void* xs, long ldxs, float out_scale,
                                float slope, float post_slope, int B, int S, int C, int d0, int d1, int d2, int dtype, hipStream_t stream) {
    const void* w1[3] = {w1a, w1b, w1c};
    const void* w2[3] = {w2a, w2b, w2c};
    const float* b1[3] = {b1a, b1b, b1c};
    const float* b2[3] = {b2a, b2b, b2c};
    const int k[3] = {ka, kb, kc};
    return resblocks_impl(x, ldx, 3, w1, w2, b1, b2, k, xs, ldxs, 0, out_scale, slope, post_slope, B, S, C, d0, d1, d2, dtype, stream);
}
