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
// The first attempt at this fusion (round 1: conv1 -> conv2 pair, weight groups behind barrier PAIRS, register-staged) was 24 %
// slower than two launches; this one has no staging registers, no per-group barrier pair and keeps y out of LDS altogether.
#include "fs2_gemm.h"

typedef unsigned rb_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned rb_u32x2 __attribute__((ext_vector_type(2)));

struct ResBlockArgs {
    const void* X; long ldx;            // [M][C] bf16 (M = B * S rows, time-major)
    const void* W1; const void* W2;     // [3][C][k][C] bf16 each (conv index, cout, tap, cin)
    const float* B1; const float* B2;   // [3][C] f32
    void* XS; long ldxs;                // [M][C] bf16
    int accumulate; float out_scale, slope;
    int S, nbatch, k, d[3], R, tiles_per_seq;
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
    static constexpr int LDS = BIAS_OFF + 6 * C * 4;
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
    const int k = a.k, pad2 = (k - 1) >> 1;

    // ---- weight stream: group gi -> slot gi & 1.  conv c = 2m (conv1 of pair m) / 2m + 1 (conv2); groups of <= TG taps
    const int gpc = (k + TG - 1) / TG;                       // groups per convolution
    const int ngroups = 6 * gpc;
    auto issue_group = [&](int gi) {
        const int c = gi / gpc, g = gi - c * gpc;
        const int tap0 = g * TG, nt = min(TG, k - tap0);
        const unsigned char* wbase = reinterpret_cast<const unsigned char*>((c & 1) ? a.W2 : a.W1) + (size_t)(c >> 1) * C * k * C * 2;
        const unsigned slot = smem_u + K::W_OFF + (unsigned)((gi & 1) * K::SLOT);
        const int npieces = nt * K::PPT;
        for (int q = wave; q < npieces; q += 8) {
            const int tl = q / K::PPT, p = q - tl * K::PPT;
            // piece p of a tap = C / PPT couts; lane -> (cout, LDS chunk position); the position holds global chunk pos ^ key(cout)
            const int cout = p * (C / K::PPT) + lane / K::CPR;
            const unsigned pos = (unsigned)(lane % K::CPR);
            const unsigned gch = pos ^ rb_key<C>((unsigned)cout);
            const unsigned voff = (unsigned)(((cout * k + tap0 + tl) * C) * 2) + (gch << 4);
            glds16_sbase(voff, wbase, __builtin_amdgcn_readfirstlane(slot + (unsigned)(tl * K::TAPB + p * 1024)));
        }
    };
    issue_group(0);
    for (int i = tid; i < 6 * C; i += 512) {
        const int c = i / C, n = i - c * C;
        bias_s[i] = ((c & 1) ? a.B2 : a.B1)[(c >> 1) * C + n];
    }

    // ---- y <- x: this wave's rows, accumulator layout (lane = row, registers = couts 4 fh + (r & 3) + 8 (r >> 2))
    f32x16 y[MB][NB];
    const bf16_t* X = reinterpret_cast<const bf16_t*>(a.X);
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
        const int tr = wave * (MB * 32) + mb * 32 + fl;
        const int t = min(max(t_first + tr, 0), a.S - 1);
        const bf16_t* row = X + (seq_row0 + (size_t)t) * a.ldx;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 v = ld4<bf16_t>(row + nb * 32 + 8 * q + 4 * fh);
                y[mb][nb][4 * q + 0] = v.x; y[mb][nb][4 * q + 1] = v.y; y[mb][nb][4 * q + 2] = v.z; y[mb][nb][4 * q + 3] = v.w;
            }
    }
    // rows of this lane inside the utterance? (bit mb)
    unsigned in_seq = 0;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
        const int t = t_first + wave * (MB * 32) + mb * 32 + fl;
        if (t >= 0 && t < a.S) in_seq |= 1u << mb;
    }

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
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    rb_u32x2 s0 = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[mb][nb][e]), __float_as_uint(v[mb][nb][4 + e]), false, false);
                    rb_u32x2 s1 = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[mb][nb][8 + e]), __float_as_uint(v[mb][nb][12 + e]), false, false);
                    c[0][e] = __uint_as_float(s0[0]); c[0][4 + e] = __uint_as_float(s0[1]);
                    c[1][e] = __uint_as_float(s1[0]); c[1][4 + e] = __uint_as_float(s1[1]);
                }
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
    write_act(y);

    // ---- the six convolutions
    f32x16 t_acc[MB][NB];
    unsigned wkey[NB], wrow[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) { wrow[nb] = (unsigned)((nb * 32 + fl) * ROWB); wkey[nb] = rb_key<C>((unsigned)(nb * 32 + fl)); }
    int gi = 0;
    // the groups of convolution c multiplied into `acc` (y for conv2: the residual is already in it; t_acc for conv1)
    auto run_conv = [&](f32x16 (&acc)[MB][NB], int c, bool add_bias) {
        const int dil = (c & 1) ? 1 : a.d[c >> 1];
        const int pad = pad2 * dil;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 bv = *reinterpret_cast<const float4*>(bias_s + c * C + nb * 32 + 8 * q + 4 * fh);
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) {
                    if (add_bias) { acc[mb][nb][4 * q] += bv.x; acc[mb][nb][4 * q + 1] += bv.y; acc[mb][nb][4 * q + 2] += bv.z; acc[mb][nb][4 * q + 3] += bv.w; }
                    else { acc[mb][nb][4 * q] = bv.x; acc[mb][nb][4 * q + 1] = bv.y; acc[mb][nb][4 * q + 2] = bv.z; acc[mb][nb][4 * q + 3] = bv.w; }
                }
            }
        for (int g = 0; g < gpc; ++g, ++gi) {
            rb_barrier_all();                  // my pieces of group gi landed (nothing younger in flight), my tile rows stored; then:
                                               // group gi + the activation tile visible to all, slot (gi + 1) & 1 free
            if (gi + 1 < ngroups) issue_group(gi + 1);
            const int tap0 = g * TG, nt = min(TG, k - tap0);
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
    for (int m = 0; m < 3; ++m) {
        run_conv(t_acc, 2 * m, false);
        rb_barrier_lds();                      // every wave has read lrelu(y): the tile may be overwritten
        write_act(t_acc);
        run_conv(y, 2 * m + 1, true);
        if (m == 2) break;
        rb_barrier_lds();
        write_act(y);
    }

    // ---- xs (+)= out_scale * y for the tile's central R rows
    bf16_t* XS = reinterpret_cast<bf16_t*>(a.XS);
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
        const int tr = wave * (MB * 32) + mb * 32 + fl;
        const int t = t_first + tr;
        const bool store = tr >= H && tr < H + a.R && t < a.S;       // (t >= 0 follows from tr >= H)
        bf16_t* row = XS + (seq_row0 + (size_t)(store ? t : 0)) * a.ldxs;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            float cc[2][8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                rb_u32x2 s0 = __builtin_amdgcn_permlane32_swap(__float_as_uint(y[mb][nb][e]), __float_as_uint(y[mb][nb][4 + e]), false, false);
                rb_u32x2 s1 = __builtin_amdgcn_permlane32_swap(__float_as_uint(y[mb][nb][8 + e]), __float_as_uint(y[mb][nb][12 + e]), false, false);
                cc[0][e] = __uint_as_float(s0[0]); cc[0][4 + e] = __uint_as_float(s0[1]);
                cc[1][e] = __uint_as_float(s1[0]); cc[1][4 + e] = __uint_as_float(s1[1]);
            }
            if (store) {
#pragma unroll
                for (int ch = 0; ch < 2; ++ch) {
                    bf16_t* p = row + nb * 32 + ch * 16 + fh * 8;
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = cc[ch][e] * a.out_scale;
                    if (a.accumulate) {
                        const uint4 old = *reinterpret_cast<const uint4*>(p);
                        const uint32_t* u = reinterpret_cast<const uint32_t*>(&old);
#pragma unroll
                        for (int e = 0; e < 4; ++e) { v[2 * e] += __uint_as_float(u[e] << 16); v[2 * e + 1] += __uint_as_float(u[e] & 0xffff0000u); }
                    }
                    uint4 o;
                    uint32_t* ou = reinterpret_cast<uint32_t*>(&o);
#pragma unroll
                    for (int e = 0; e < 4; ++e) ou[e] = pack_bf16x2(v[2 * e], v[2 * e + 1]);
                    *reinterpret_cast<uint4*>(p) = o;
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

extern "C" int fs2_resblock_supported(int C, int k, int d0, int d1, int d2, int dtype) {
    if (dtype != FS2_BF16 || (C != 32 && C != 64) || k < 1 || k > 11 || !(k & 1)) return 0;
    if (d0 < 1 || d1 < 1 || d2 < 1) return 0;
    const int pad2 = (k - 1) / 2;
    if (pad2 * (d0 > d1 ? (d0 > d2 ? d0 : d2) : (d1 > d2 ? d1 : d2)) > 32) return 0;                 // a shifted read stays inside the guard rows
    const int H = pad2 * (d0 + d1 + d2 + 3);
    const int E = C == 32 ? 1024 : 512;
    return E - 2 * H >= 64 ? 1 : 0;
}

extern "C" int fs2_resblock_fwd(const void* x, long ldx, const void* w1, const void* w2, const float* b1, const float* b2, void* xs,
                                long ldxs, int accumulate, float out_scale, float slope, int B, int S, int C, int k, int d0, int d1,
                                int d2, int dtype, hipStream_t stream) {
    FS2_CHECK_ARG(x && w1 && w2 && b1 && b2 && xs, "resblock_fwd: null pointer");
    FS2_CHECK_ARG(B > 0 && S > 0, "resblock_fwd: bad shape");
    FS2_CHECK_ARG(fs2_resblock_supported(C, k, d0, d1, d2, dtype), "resblock_fwd: unsupported (C in {32, 64}, odd k <= 11, bf16)");
    FS2_CHECK_ARG(ldx % 8 == 0 && ldxs % 8 == 0 && (((uintptr_t)x | (uintptr_t)xs | (uintptr_t)w1 | (uintptr_t)w2) & 15) == 0,
                  "resblock_fwd: rows must be 16-byte addressable");
    FS2_CHECK_ARG(slope > 0.f && slope < 1.f, "resblock_fwd: leaky-ReLU slope in (0, 1)");
    ResBlockArgs a;
    a.X = x; a.ldx = ldx; a.W1 = w1; a.W2 = w2; a.B1 = b1; a.B2 = b2; a.XS = xs; a.ldxs = ldxs;
    a.accumulate = accumulate; a.out_scale = out_scale; a.slope = slope; a.S = S; a.nbatch = B; a.k = k;
    a.d[0] = d0; a.d[1] = d1; a.d[2] = d2;
    const int H = ((k - 1) / 2) * (d0 + d1 + d2 + 3);
    const int E = C == 32 ? 1024 : 512;
    a.R = E - 2 * H;
    a.tiles_per_seq = fs2_cdiv(S, a.R);
    if (C == 32) launch_resblock<32>(a, stream); else launch_resblock<64>(a, stream);
    FS2_CHECK_LAUNCH("resblock_fwd");
    return FS2_OK;
}
