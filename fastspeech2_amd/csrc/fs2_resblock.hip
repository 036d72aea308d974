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
//   * what a convolution READS (lrelu(y) resp. t) is written as bf16 into ONE LDS tile [GUARD + E + GUARD rows][C], 16-byte chunks
//     XOR-swizzled per row so that ds_read_b128 by 16 consecutive rows is conflict-free; a tap is a row shift of the fragment address;
//     rows outside the utterance are written as zeros (= the reference's zero padding of every convolution's input);
//   * halo by recomputation: the tile covers H = (k-1)/2 * (d0 + d1 + d2 + 3) more rows on each side than it outputs; whatever is
//     computed from outside the tile stays inside that margin (it moves inward by one convolution's reach per convolution);
//   * weights stream L2 -> LDS by LDS-DMA (global_load_lds_dwordx4, source-side swizzle) in groups of up to TG taps through two
//     slots, the next group in flight while the current one is multiplied; one raw barrier per group.
// Measured (profiles/r05b-r05d): batch synthesis 9.55 -> 8.13 ms per step with one launch per block, -> 7.47 ms with the three blocks of a
// stage in one launch (RTF 2.29e-4 -> 1.80e-4); MFMA busy 42 % (C = 64) / 29 % (C = 32).  Requesting the fragments one k-slice ahead
// through two register sets (order pinned with sched_barrier) was SLOWER (7.47 -> 7.75 ms: 241 registers, the address arithmetic no
// longer interleaves with the MFMAs); the compiler's own order stays.
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
    int accumulate; float out_scale, slope;
    int S, nbatch, d[3], R, tiles_per_seq;
};

template <int C> struct RbCfg {
    static constexpr int E = C == 32 ? 1024 : 512;          // tile rows (8 waves x MB x 32)
    static constexpr int GUARD = 32;                         // rows in front of / behind the tile a shifted read may touch (|shift| <= 25)
    static constexpr int MB = E / 256, NB = C / 32, KS = C / 16, CPR = C / 8;
    static constexpr int ROWB = C * 2;
    static constexpr int TG = C == 32 ? 11 : 4;              // taps per weight group
    static constexpr int TAPB = C * C * 2;                   // bytes of one tap's weights
    static constexpr int ACT_BYTES = (E + 2 * GUARD) * ROWB;
    static constexpr int W_OFF = ACT_BYTES, SLOT = TG * TAPB;
    static constexpr int BIAS_OFF = W_OFF + 2 * SLOT;
    static constexpr int LDS = BIAS_OFF + 18 * C * 4;
    static constexpr int PPT = TAPB / 1024;                  // 1 KiB DMA pieces per tap (2 / 8)
};
// swizzle key of LDS row r: chunk c is stored at chunk c ^ key (16 consecutive rows x one chunk -> 16 distinct 16-byte slots of the
// 256-byte bank window: 4 rows per window at C = 32, 2 at C = 64)
template <int C> __device__ __forceinline__ unsigned rb_key(unsigned r) { return C == 32 ? ((r >> 2) & 3u) : ((r >> 1) & 7u); }

// raw barrier with this wave's LDS stores (and DMA pieces) retired first; the memory clobber keeps the compiler's LDS accesses on their side
__device__ __forceinline__ void rb_barrier_all() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ void rb_barrier_lds() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int C>
__global__ void __launch_bounds__(512, 2) resblock_fused_kernel(ResBlockArgs a) {
    typedef RbCfg<C> K;
    constexpr int E = K::E, GUARD = K::GUARD, MB = K::MB, NB = K::NB, KS = K::KS, ROWB = K::ROWB, TG = K::TG;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, fl = lane & 31, fh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned smem_u = lds_addr(smem);
    float* bias_s = reinterpret_cast<float*>(smem + K::BIAS_OFF);

    const int seq = blockIdx.x / a.tiles_per_seq, tile = blockIdx.x - seq * a.tiles_per_seq;
    const int H = (E - a.R) >> 1;
    const int t_first = tile * a.R - H;                      // utterance row of tile row 0
    const size_t seq_row0 = (size_t)seq * a.S;
    // ---- weight stream over all blocks: group gi -> slot gi & 1.  Block j: convolution c = 2m (conv1 of pair m) / 2m + 1 (conv2), each
    // in gpc_j = ceil(k_j / TG) groups of <= TG taps
    int gpc[3], gstart[4];
    gstart[0] = 0;
#pragma unroll
    for (int j = 0; j < 3; ++j) { gpc[j] = j < a.nblk ? (a.k[j] + TG - 1) / TG : 0; gstart[j + 1] = gstart[j] + 6 * gpc[j]; }
    const int ngroups = gstart[3];
    auto issue_group = [&](int gi) {
        const int j = gi >= gstart[2] ? 2 : (gi >= gstart[1] ? 1 : 0);
        const int kj = a.k[j], gl = gi - gstart[j];
        const int c = gl / gpc[j], g = gl - c * gpc[j];
        const int tap0 = g * TG, nt = min(TG, kj - tap0);
        const unsigned char* wbase = reinterpret_cast<const unsigned char*>((c & 1) ? a.W2[j] : a.W1[j]) + (size_t)(c >> 1) * C * kj * C * 2;
        const unsigned slot = smem_u + K::W_OFF + (unsigned)((gi & 1) * K::SLOT);
        const int npieces = nt * K::PPT;
        for (int q = wave; q < npieces; q += 8) {
            const int tl = q / K::PPT, p = q - tl * K::PPT;
            // piece p of a tap = C / PPT couts; lane -> (cout, LDS chunk position); the position holds global chunk pos ^ key(cout)
            const int cout = p * (C / K::PPT) + lane / K::CPR;
            const unsigned pos = (unsigned)(lane % K::CPR);
            const unsigned gch = pos ^ rb_key<C>((unsigned)cout);
            const unsigned voff = (unsigned)(((cout * kj + tap0 + tl) * C) * 2) + (gch << 4);
            glds16_sbase(voff, wbase, __builtin_amdgcn_readfirstlane(slot + (unsigned)(tl * K::TAPB + p * 1024)));
        }
    };
    issue_group(0);
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
            const unsigned rowaddr = lr * ROWB, key = rb_key<C>(lr);
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
                    *reinterpret_cast<uint4*>(smem + rowaddr + ((chunk ^ key) << 4)) = o;
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
    unsigned wkey[NB], wrow[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) { wrow[nb] = (unsigned)((nb * 32 + fl) * ROWB); wkey[nb] = rb_key<C>((unsigned)(nb * 32 + fl)); }
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
            rb_barrier_all();                  // my pieces of group gi landed (nothing younger in flight), my tile rows stored; then:
                                               // group gi + the activation tile visible to all, slot (gi + 1) & 1 free
            if (gi + 1 < ngroups) issue_group(gi + 1);
            const int tap0 = g * TG, nt = min(TG, kj - tap0);
            const unsigned wslot = (unsigned)(K::W_OFF + (gi & 1) * K::SLOT);
            for (int tl = 0; tl < nt; ++tl) {
                const int shift = (tap0 + tl) * dil - pad;
                unsigned xrow[MB], xkey[MB];
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) {
                    const unsigned lr = (unsigned)(GUARD + wave * (MB * 32) + mb * 32 + fl + shift);
                    xrow[mb] = lr * ROWB; xkey[mb] = rb_key<C>(lr);
                }
                const unsigned wtap = wslot + (unsigned)(tl * K::TAPB);
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const unsigned chunk = (unsigned)(2 * ks + fh);
                    rb_u32x4 wf[NB], xf[MB];
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb)
                        wf[nb] = *reinterpret_cast<const rb_u32x4*>(smem + wtap + wrow[nb] + ((chunk ^ wkey[nb]) << 4));
#pragma unroll
                    for (int mb = 0; mb < MB; ++mb)
                        xf[mb] = *reinterpret_cast<const rb_u32x4*>(smem + xrow[mb] + ((chunk ^ xkey[mb]) << 4));
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
        if (j > 0) rb_barrier_lds();           // every wave has finished the previous block's last convolution
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
            float cc[2][8];
            to_rows(v, cc);
            if (store) {
#pragma unroll
                for (int ch = 0; ch < 2; ++ch) {
                    uint4 o;
                    uint32_t* ou = reinterpret_cast<uint32_t*>(&o);
#pragma unroll
                    for (int e = 0; e < 4; ++e) ou[e] = pack_bf16x2(cc[ch][2 * e], cc[ch][2 * e + 1]);       // (exact: the values are bf16 already)
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
    if (dtype != FS2_BF16 || (C != 32 && C != 64) || k < 1 || k > 11 || !(k & 1)) return 0;
    if (d0 < 1 || d1 < 1 || d2 < 1) return 0;
    const int pad2 = (k - 1) / 2;
    if (pad2 * (d0 > d1 ? (d0 > d2 ? d0 : d2) : (d1 > d2 ? d1 : d2)) > 32) return 0;                 // a shifted read stays inside the guard rows
    const int E = C == 32 ? 1024 : 512;
    return E - 2 * rb_halo(k, d0, d1, d2) >= 64 ? 1 : 0;
}

static int resblocks_impl(const void* x, long ldx, int nblk, const void* const* w1, const void* const* w2, const float* const* b1,
                          const float* const* b2, const int* k, void* xs, long ldxs, int accumulate, float out_scale, float slope, int B,
                          int S, int C, int d0, int d1, int d2, int dtype, hipStream_t stream) {
    FS2_CHECK_ARG(x && xs && nblk >= 1 && nblk <= 3, "resblock_fwd: null pointer / block count");
    FS2_CHECK_ARG(B > 0 && S > 0, "resblock_fwd: bad shape");
    FS2_CHECK_ARG(ldx % 8 == 0 && ldxs % 8 == 0 && (((uintptr_t)x | (uintptr_t)xs) & 15) == 0, "resblock_fwd: rows must be 16-byte addressable");
    FS2_CHECK_ARG(slope > 0.f && slope < 1.f, "resblock_fwd: leaky-ReLU slope in (0, 1)");
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
    a.accumulate = accumulate; a.out_scale = out_scale; a.slope = slope; a.S = S; a.nbatch = B;
    a.d[0] = d0; a.d[1] = d1; a.d[2] = d2;
    const int E = C == 32 ? 1024 : 512;
    a.R = E - 2 * H;
    a.tiles_per_seq = fs2_cdiv(S, a.R);
    if (C == 32) launch_resblock<32>(a, stream); else launch_resblock<64>(a, stream);
    FS2_CHECK_LAUNCH("resblock_fwd");
    return FS2_OK;
}

extern "C" int fs2_resblock_fwd(const void* x, long ldx, const void* w1, const void* w2, const float* b1, const float* b2, void* xs,
                                long ldxs, int accumulate, float out_scale, float slope, int B, int S, int C, int k, int d0, int d1,
                                int d2, int dtype, hipStream_t stream) {
    return resblocks_impl(x, ldx, 1, &w1, &w2, &b1, &b2, &k, xs, ldxs, accumulate, out_scale, slope, B, S, C, d0, d1, d2, dtype, stream);
}

extern "C" int fs2_resstage_fwd(const void* x, long ldx, const void* w1a, const void* w2a, const float* b1a, const float* b2a, int ka,
                                const void* w1b, const void* w2b, const float* b1b, const float* b2b, int kb, const void* w1c,
                                const void* w2c, const float* b1c, const float* b2c, int kc, void* xs, long ldxs, float out_scale,
                                float slope, int B, int S, int C, int d0, int d1, int d2, int dtype, hipStream_t stream) {
    const void* w1[3] = {w1a, w1b, w1c};
    const void* w2[3] = {w2a, w2b, w2c};
    const float* b1[3] = {b1a, b1b, b1c};
    const float* b2[3] = {b2a, b2b, b2c};
    const int k[3] = {ka, kb, kc};
    return resblocks_impl(x, ldx, 3, w1, w2, b1, b2, k, xs, ldxs, 0, out_scale, slope, B, S, C, d0, d1, d2, dtype, stream);
}
