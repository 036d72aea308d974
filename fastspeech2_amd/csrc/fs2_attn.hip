// fs2_attn.hip — fused scaled-dot-product attention with key-padding mask (forward + backward), dk = dv = 128.
// Reference: transformer/Modules.py:14-25 (bmm, /sqrt(dk), masked_fill(-inf), softmax, bmm) and the head
// split/merge of transformer/SubLayers.py:39-52.  The S x S score matrix never reaches HBM: online softmax
// over 32-key tiles, only the per-row log-sum-exp is saved for backward (flash-style).
//
// Layout: the fused QKV projection writes one buffer qkv[M][3*H*128] = (q | k | v), head h in columns
// h*128..h*128+127 of each third; the context is written as ctx[M][H*128] (== the reference's merged
// (B, S, H*dv) tensor), so no permute/contiguous copy exists anywhere.
//
// MFMA mapping (v_mfma_f32_32x32x2_f32, exact f32): scores are computed TRANSPOSED where that makes the
// probability tile land directly in the A-operand layout of the following product (C layout: lane = col,
// regs = rows (r&3)+8*(r>>2)+4*(lane>>5);  A layout: lane = row, k-slot = lane>>5) -> P never leaves registers.
// Storage dtype T may be bf16; tiles are widened to f32 when staged into LDS (first revision: f32 MFMA rate).
#include "fs2_common.h"
#include "fs2_gemm.h"
#include <type_traits>

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define DK 128
// row r of a [32][128] f32 tile, 16-B chunk c (0..31): XOR-swizzled so ds_read_b128 by 16 distinct rows is conflict-free
__device__ __forceinline__ int swz(int row, int chunk) { return row * 512 + ((chunk ^ (row & 15)) << 4); }
__device__ __forceinline__ int crow(int r, int h2) { return (r & 3) + 8 * (r >> 2) + 4 * h2; }

// stage a [32 rows][128] tile (rows row0.., column offset col0 of a ld-strided matrix) into LDS as f32, swizzled.
// rows >= nrows_valid are zero-filled. 256 threads (or NT threads).
template <typename T, int NT>
__device__ __forceinline__ void stage_tile(unsigned char* lds, const T* __restrict__ base, long ld, int row0, int nrows_valid,
                                           int tid) {
#pragma unroll
    for (int i = 0; i < 1024 / NT; ++i) {
        int idx = tid + NT * i;
        int r = idx >> 5, c = idx & 31;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < nrows_valid) v = ld4<T>(base + (size_t)(row0 + r) * ld + c * 4);
        *reinterpret_cast<float4*>(lds + swz(r, c)) = v;
    }
}

// ------------------------------------------------------------------ forward
template <typename T>
__global__ void __launch_bounds__(256, 1) attn_fwd_kernel(const T* __restrict__ qkv, long ld, T* __restrict__ ctx, long ldo,
                                                          float* __restrict__ lse, const int32_t* __restrict__ lens, int S,
                                                          int H, float scale) {
    __shared__ __attribute__((aligned(16))) unsigned char sK[32 * 512];
    __shared__ __attribute__((aligned(16))) unsigned char sV[32 * 512];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, fl = lane & 31, h2 = lane >> 5;
    const int qt = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    const int len = lens ? min(lens[b], S) : S;
    const int q0 = qt * 128;
    const size_t rowbase = (size_t)b * S;
    T* out = ctx + rowbase * ldo + h * DK;
    float* lse_o = lse + ((size_t)b * H + h) * S;
    if (q0 >= len) {  // fully padded query tile: zeros (the rows are masked to zero downstream anyway)
        for (int i = tid; i < 128 * 32; i += 256) {
            int r = i >> 5, c = (i & 31) * 4;
            if (q0 + r < S) st4<T>(out + (size_t)(q0 + r) * ldo + c, make_float4(0.f, 0.f, 0.f, 0.f));
        }
        if (tid < 128 && q0 + tid < S) lse_o[q0 + tid] = 0.f;
        return;
    }
    const T* Q = qkv + rowbase * ld + h * DK;
    const T* K = qkv + rowbase * ld + (size_t)H * DK + h * DK;
    const T* V = qkv + rowbase * ld + (size_t)2 * H * DK + h * DK;

    const int myq = q0 + w * 32 + fl;
    float qf[64];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (myq < S) v = ld4<T>(Q + (size_t)myq * ld + h2 * 64 + u * 4);
        qf[u * 4 + 0] = v.x * scale; qf[u * 4 + 1] = v.y * scale; qf[u * 4 + 2] = v.z * scale; qf[u * 4 + 3] = v.w * scale;
    }
    f32x16 o[4];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[nb][r] = 0.f;
    float m = -INFINITY, l = 0.f;

    for (int k0 = 0; k0 < len; k0 += 32) {
        __syncthreads();
        stage_tile<T, 256>(sK, K, ld, k0, min(32, S - k0), tid);
        stage_tile<T, 256>(sV, V, ld, k0, min(32, S - k0), tid);
        __syncthreads();
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            float4 kf = *reinterpret_cast<const float4*>(sK + swz(fl, h2 * 16 + u));
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.x, qf[u * 4 + 0], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.y, qf[u * 4 + 1], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.z, qf[u * 4 + 2], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.w, qf[u * 4 + 3], s, 0, 0, 0);
        }
        // s[r] = score(q = lane&31, key = k0 + crow(r,h2))
        float mx = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (k0 + crow(r, h2) >= len) s[r] = -INFINITY;
            mx = fmaxf(mx, s[r]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float mn = fmaxf(m, mx);
        float alpha = __expf(m - mn);   // m = -inf on the first tile -> 0
        float rs = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = __expf(s[r] - mn); rs += s[r]; }
        rs += __shfl_xor(rs, 32, 64);
        l = l * alpha + rs;
        m = mn;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float ar = __shfl(alpha, crow(r, h2), 64);
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) o[nb][r] *= ar;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            int key = crow(r, h2);
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) {
                int col = nb * 32 + fl;
                float vb = *reinterpret_cast<const float*>(sV + swz(key, col >> 2) + (col & 3) * 4);
                o[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(s[r], vb, o[nb], 0, 0, 0);
            }
        }
    }
    float linv = l > 0.f ? 1.f / l : 0.f;
    if (h2 == 0 && myq < S) lse_o[myq] = (l > 0.f) ? m + __logf(l) : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float li = __shfl(linv, crow(r, h2), 64);
        int q = q0 + w * 32 + crow(r, h2);
        if (q < S) {
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) Elem<T>::st(out + (size_t)q * ldo + nb * 32 + fl, o[nb][r] * li);
        }
    }
}


// ====================================================================================================
// bf16 path: v_mfma_f32_32x32x16_bf16 (2.5 PF roof), same register-resident-P structure as the f32 kernels.
// Tiles are [rows][128] bf16 (256-B rows) with ONE swizzle that serves both access patterns:
//   16-B chunk c of row r lives at chunk c ^ S(r),  S(r) = ((r&3)<<2) | ((r>>2)&3)
//   * ds_read_b128 of one chunk by 16 distinct rows (K-contiguous operand)      -> 16 distinct slots
//   * ds_read_b64_tr_b16 of 4 consecutive rows x 64 B (row-major operand, transposing) -> 4 distinct bank quarters
// k-slot convention of every MFMA here: lane half h = lane>>5, element e = 0..7.
//   K-contiguous operands: d = 16*s + 8*h + e.
//   row-indexed operands (P / dS as A, V / dO / Q / K as B): row = 16*u + 4*h + (e&3) + 8*(e>>2)  — exactly the rows
//   a lane owns in the 32x32 C layout, so probabilities feed the next MFMA straight from registers.
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) s16x4* lds_s4p;

__device__ __forceinline__ int swzb(int row, int c16) { return row * 256 + ((c16 ^ (((row & 3) << 2) | ((row >> 2) & 3))) << 4); }

template <int NT, int ROWS>
__device__ __forceinline__ void stage_tile_bf16(unsigned char* lds, const bf16_t* __restrict__ base, long ld, int row0,
                                                int nrows_valid, int tid) {
#pragma unroll
    for (int i = 0; i < ROWS * 16 / NT; ++i) {
        int idx = tid + NT * i;
        int r = idx >> 4, c = idx & 15;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (r < nrows_valid) v = *reinterpret_cast<const uint4*>(base + (size_t)(row0 + r) * ld + c * 8);
        *reinterpret_cast<uint4*>(lds + swzb(r, c)) = v;
    }
}
// K-contiguous fragment: row `row`, d-slots 16*s + 8*h .. +7
__device__ __forceinline__ bf16x8 frag_k(const unsigned char* lds, int row, int s, int h) {
    return __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(lds + swzb(row, 2 * s + h)));
}
// transposed fragment: rows rowbase + 4h + {0..3, 8..11}, column col (= 32*nb + lane&31)
__device__ __forceinline__ bf16x8 frag_t(const unsigned char* lds, int rowbase, int nb, int lane) {
    int li = lane & 15, g = lane >> 4, h = g >> 1;
    int row = rowbase + 4 * h + (li >> 2);
    int col = nb * 32 + 16 * (g & 1) + 4 * (li & 3);       // 8-byte piece inside 16-B chunk col>>3
    int off0 = swzb(row, col >> 3) + ((col & 7) << 1);
    int off1 = swzb(row + 8, col >> 3) + ((col & 7) << 1);
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4p)(lds + off0));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4p)(lds + off1));
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}
__device__ __forceinline__ bf16x8 pack8(const float* p) {
    uint4 v;
    v.x = pack_bf16x2(p[0], p[1]); v.y = pack_bf16x2(p[2], p[3]); v.z = pack_bf16x2(p[4], p[5]); v.w = pack_bf16x2(p[6], p[7]);
    return __builtin_bit_cast(bf16x8, v);
}
// load this lane's K-contiguous register fragments of one row (8 steps x 8 bf16), optionally scaled.  (Round 6 measured the alternative - the
// 128 rows read row-wise in full lines, parked in a free LDS image, fragments by ds_read_b128: the dQ prologue is 15 000 cycles either
// way (tools/attn_phases.py 1): a workgroup's cold start is a memory round trip under a whole-chip burst, not a coalescing problem)
__device__ __forceinline__ void load_row_frags(bf16x8* f, const bf16_t* row, int h, bool ok) {
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        uint4 v = make_uint4(0, 0, 0, 0);
        if (ok) v = *reinterpret_cast<const uint4*>(row + 16 * s + 8 * h);
        f[s] = __builtin_bit_cast(bf16x8, v);
    }
}

// Register-staged prefetch of a [64 rows][128] bf16 tile (256 threads: 4 x 16 B per thread).  Rows are CLAMPED to the
// sequence's last row instead of zero-filled: every consumer already multiplies rows >= len by an exact 0 probability /
// score gradient, so only finiteness matters, and the load needs no predicate (r01h: the synchronous
// load -> ds_write -> barrier staging left the MFMA pipe idle for the whole L2 round trip of every tile).
// (32-bit element offsets from the wave-uniform base: the 64-bit form cost ~6 vector instructions per load)
#define TILE_LD1(t, base, ld_, row0, last_row, I) \
    (t##I) = *reinterpret_cast<const uint4*>((base) + (unsigned)(min((row0) + ((tid + 256 * I) >> 4), (last_row)) * (int)(ld_) + (tid & 15) * 8))
#define TILE_LOAD_REGS(t, base, ld_, row0, last_row) do { TILE_LD1(t, base, ld_, row0, last_row, 0); TILE_LD1(t, base, ld_, row0, last_row, 1); \
    TILE_LD1(t, base, ld_, row0, last_row, 2); TILE_LD1(t, base, ld_, row0, last_row, 3); } while (0)
#define TILE_ST1(lds, t, I) *reinterpret_cast<uint4*>((lds) + swzb((tid + 256 * I) >> 4, tid & 15)) = (t##I)
#define TILE_STORE_REGS(lds, t) do { TILE_ST1(lds, t, 0); TILE_ST1(lds, t, 1); TILE_ST1(lds, t, 2); TILE_ST1(lds, t, 3); } while (0)


#ifdef FS2_DEV
// dev builds: s_memtime stamps of workgroup (0, 0, 0), wave 0 at the phase boundaries of its first tiles (tools/attn_phases.py)
__device__ unsigned long long fs2_attn_stamps[16 * 12];
__device__ int fs2_attn_stamp_kernel = 2;      // 0 = forward, 1 = dQ, 2 = dK/dV (set with fs2_dev_attn_stamp_select)
extern "C" int fs2_dev_attn_stamp_select(int k) { return hipMemcpyToSymbol(HIP_SYMBOL(fs2_attn_stamp_kernel), &k, sizeof(int)) == hipSuccess ? 0 : 1; }
extern "C" int fs2_dev_attn_stamps(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(fs2_attn_stamps), sizeof(fs2_attn_stamps)) == hipSuccess ? 0 : 1;
}
// (the selector is read ONCE, at the top of the kernel: a load beside every stamp is a vmcnt(0) wait beside every stamp)
#define FS2_STAMP_INIT() const bool fs2_stamp_on = fs2_attn_stamp_kernel == FS2_STAMP_KERNEL && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0
#define FS2_STAMP_AT(t0_, i) do { if (fs2_stamp_on && (t0_) < 16 * 64) fs2_attn_stamps[((t0_) >> 6) * 12 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#define FS2_STAMP(i) FS2_STAMP_AT(q0, i)
#define FS2_STAMP_RAW(i) do { if (fs2_stamp_on) fs2_attn_stamps[(i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define FS2_STAMP_INIT() do {} while (0)
#define FS2_STAMP_AT(t0_, i) do {} while (0)
#define FS2_STAMP(i) do {} while (0)
#define FS2_STAMP_RAW(i) do {} while (0)
#endif

// XCD-AWARE BLOCK MAP (round 6).  The grid is (tiles, H, B) and the dispatcher deals consecutive workgroups round-robin to the 8
// XCDs - with the tile index fastest, the 8 query (key) tiles of ONE (sequence, head) landed on 8 different XCDs and each of them
// pulled that head's whole K / V (Q / dO) stream through its own L2: 8 x the fabric traffic (r05zzz PMC: 143-167 MB fetched per
// decoder launch for ~91 MB of operands; r06p phase stamps: the forward's 8 tile loads alone held a wave for 1 300 - 2 500 cycles of a
// 5 400-cycle tile).  Here the dispatch-order index L is re-read as: groups of 8 (sequence, head) pairs, inside a group the PAIR is
// the fastest index (L mod 8 = the XCD) and the tile the slower one - the tiles of a pair share an XCD and start together.
__device__ __forceinline__ void attn_block_map(int H, int& t, int& h, int& b) {
    const int nt = (int)gridDim.x, nbh = (int)(gridDim.y * gridDim.z);
    const int L = (int)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z));
    const int G = L / (8 * nt), r = L - G * (8 * nt);
    const int rem = min(8, nbh - 8 * G);                    // pairs in this group (< 8 only in the last one)
    const int bh = 8 * G + r % rem;
    t = r / rem;
    b = bh / H; h = bh - b * H;
}

// Lane-constant fragment addresses of a [64 rows][256 B] swizzled tile image pair (round 6): absolute LDS byte addresses of this
// lane's pieces in the CURRENT buffer's first image; the second image (+16 KiB), the 32-row block (+8 KiB) and the 16-row group of a
// transposing read (+4 KiB) enter as immediate offsets, and flip() moves all of them to the other buffer (+-32 KiB) once per tile.
// The first form recomputed swzb() for every read: ~80 of a tile's ~240 vector instructions in the forward kernel.
// IMG: bytes of one tile image (64 rows x 256 B = 16 KiB; 8 KiB in the 32-key-tile kernels); NTHR: threads of the workgroup
template <int IMG, int NTHR> struct FragAddrT {
    unsigned k[8], t0[4], t1[4], st[4];
    int step;
    __device__ __forceinline__ void init(unsigned base, int tid) {
        const int lane = tid & 63, fl = lane & 31, h2 = lane >> 5;
#pragma unroll
        for (int i = 0; i < 8; ++i) k[i] = base + (unsigned)swzb(fl, 2 * i + h2);
        const int li = lane & 15, g = lane >> 4, hh = g >> 1, rr = 4 * hh + (li >> 2);
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
            const int col = nb * 32 + 16 * (g & 1) + 4 * (li & 3);
            t0[nb] = base + (unsigned)(swzb(rr, col >> 3) + ((col & 7) << 1));
            t1[nb] = base + (unsigned)(swzb(rr + 8, col >> 3) + ((col & 7) << 1));
        }
#pragma unroll
        for (int I = 0; I < 4; ++I) st[I] = base + (unsigned)(2 * IMG) + (unsigned)swzb((tid + NTHR * I) >> 4, tid & 15);   // stores go to the OTHER buffer
        step = 2 * IMG;
    }
    __device__ __forceinline__ void flip() {
#pragma unroll
        for (int i = 0; i < 8; ++i) k[i] += (unsigned)step;
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) { t0[nb] += (unsigned)step; t1[nb] += (unsigned)step; }
#pragma unroll
        for (int I = 0; I < 4; ++I) st[I] -= (unsigned)step;
        step = -step;
    }
    // K-contiguous fragment: image img (0 / 1), 32-row block blk, d-slice i
    __device__ __forceinline__ bf16x8 rk(int img, int blk, int i) const {
        typedef unsigned lds_v4u __attribute__((ext_vector_type(4)));
        typedef __attribute__((address_space(3))) const lds_v4u* lds_u4p;
        return __builtin_bit_cast(bf16x8, *(lds_u4p)(size_t)(k[i] + (unsigned)(img * IMG + blk * 8192)));
    }
    // transposed fragment: image img, rows rb .. rb + 15 (rb a multiple of 16), column block nb
    __device__ __forceinline__ bf16x8 rt(int img, int rb, int nb) const {
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4p)(size_t)(t0[nb] + (unsigned)(img * IMG + rb * 256)));
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4p)(size_t)(t1[nb] + (unsigned)(img * IMG + rb * 256)));
        return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    }
    // register-staged tile store into the other buffer's image img
    __device__ __forceinline__ void store(int img, const uint4& v, int I) const {
#if defined(__HIP_DEVICE_COMPILE__)
        typedef unsigned lds_v4u __attribute__((ext_vector_type(4)));
        typedef __attribute__((address_space(3))) lds_v4u* lds_u4w;
        *(lds_u4w)(size_t)(st[I] + (unsigned)(img * IMG)) = lds_v4u{v.x, v.y, v.z, v.w};
#else
        (void)img; (void)v; (void)I;
#endif
    }
};
typedef FragAddrT<16384, 256> FragAddr;                  // 64-row tile images, 256-thread workgroups
#define TILE_STORE_FA(fa, img, t) do { (fa).store(img, t##0, 0); (fa).store(img, t##1, 1); (fa).store(img, t##2, 2); (fa).store(img, t##3, 3); } while (0)


// Row-per-lane epilogue (round 6).  The P V / dS K / P^T dO products are issued with their operands SWAPPED, so the accumulators hold
// the TRANSPOSED result: lane = output ROW (query / key), registers = columns d = 32 nb + 8 (r >> 2) + 4 h2 + (r & 3).  One
// v_permlane32_swap per register pair then gives each half-wave 8 consecutive columns = one 16-byte bf16 store (8 per lane instead
// of the 64 two-byte stores of the column-per-lane form - a store-issue-bound tail of the order of a tenth of a workgroup's life),
// and every per-row factor (the online-softmax rescale, 1 / l, the softmax scale) is a per-LANE scalar: no cross-lane shuffles.
typedef unsigned attn_u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void attn_store_row(bf16_t* rowp, const f32x16 (&acc)[4], float f, int h2, bool ok) {
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
        float c[2][8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            attn_u32x2 s0 = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[nb][e]), __float_as_uint(acc[nb][4 + e]), false, false);
            attn_u32x2 s1 = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[nb][8 + e]), __float_as_uint(acc[nb][12 + e]), false, false);
            c[0][e] = __uint_as_float(s0[0]); c[0][4 + e] = __uint_as_float(s0[1]);
            c[1][e] = __uint_as_float(s1[0]); c[1][4 + e] = __uint_as_float(s1[1]);
        }
#pragma unroll
        for (int ch = 0; ch < 2; ++ch) {
            uint4 o;
            o.x = pack_bf16x2(c[ch][0] * f, c[ch][1] * f); o.y = pack_bf16x2(c[ch][2] * f, c[ch][3] * f);
            o.z = pack_bf16x2(c[ch][4] * f, c[ch][5] * f); o.w = pack_bf16x2(c[ch][6] * f, c[ch][7] * f);
            if (ok) *reinterpret_cast<uint4*>(rowp + nb * 32 + ch * 16 + h2 * 8) = o;
        }
    }
}

// Raised wave priority around the MFMA runs of the two-waves-per-SIMD kernels (MI355X_MICROARCH: +4-7 % on an attention loop; here the
// forward 60.6 -> 58.8 us same process, tools/attn_prio.py): the wave that is in a matrix run is not held up by its partner's softmax VALU.
// Dev builds can switch it off (fs2_dev_attn_prio).
#ifdef FS2_DEV
__device__ int fs2_attn_prio_on = 1;
extern "C" int fs2_dev_attn_prio(int on) { return hipMemcpyToSymbol(HIP_SYMBOL(fs2_attn_prio_on), &on, sizeof(int)) == hipSuccess ? 0 : 1; }
#define FS2_ATTN_PRIO_INIT() const bool fs2_prio = fs2_attn_prio_on != 0
#define FS2_ATTN_PRIO(v) do { if (fs2_prio) { if (v) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); } } while (0)
#else
#define FS2_ATTN_PRIO_INIT() do {} while (0)
#define FS2_ATTN_PRIO(v) __builtin_amdgcn_s_setprio(v)
#endif
#define FS2_STAMP_KERNEL 0
__global__ void __launch_bounds__(256, 2) attn_fwd_bf16_kernel(const bf16_t* __restrict__ qkv, long ld, bf16_t* __restrict__ ctx,
                                                               long ldo, float* __restrict__ lse,
                                                               const int32_t* __restrict__ lens, int S, int H, float scale) {
    __shared__ __attribute__((aligned(16))) unsigned char sKV[2][2][64 * 256];   // [buffer][K | V]
    FS2_ATTN_PRIO_INIT();
    FS2_STAMP_INIT();
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, fl = lane & 31, h2 = lane >> 5;
    int qt, h, b;
    attn_block_map(H, qt, h, b);
    const int len = lens ? min(lens[b], S) : S;
    const int q0 = qt * 128;
    const size_t rowbase = (size_t)b * S;
    bf16_t* out = ctx + rowbase * ldo + h * DK;
    float* lse_o = lse + ((size_t)b * H + h) * S;
    if (q0 >= len) {
        for (int i = tid; i < 128 * 32; i += 256) {
            int r = i >> 5, c = (i & 31) * 4;
            if (q0 + r < S) st4<bf16_t>(out + (size_t)(q0 + r) * ldo + c, make_float4(0.f, 0.f, 0.f, 0.f));
        }
        if (tid < 128 && q0 + tid < S) lse_o[q0 + tid] = 0.f;
        return;
    }
    const bf16_t* Q = qkv + rowbase * ld + h * DK;
    const bf16_t* K = qkv + rowbase * ld + (size_t)H * DK + h * DK;
    const bf16_t* V = qkv + rowbase * ld + (size_t)2 * H * DK + h * DK;
    const int myq = q0 + w * 32 + fl;
    bf16x8 qf[8];
    load_row_frags(qf, Q + (size_t)min(myq, S - 1) * ld, h2, myq < S);
    f32x16 o[4];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[nb][r] = 0.f;
    float m = -INFINITY, l = 0.f;                        // m: running maximum of s * scale * log2(e)
    const float sc2 = scale * 1.4426950408889634f;

    uint4 tk0, tk1, tk2, tk3, tv0, tv1, tv2, tv3;
    TILE_LOAD_REGS(tk, K, ld, 0, S - 1);
    TILE_LOAD_REGS(tv, V, ld, 0, S - 1);
    TILE_STORE_REGS(sKV[0][0], tk);
    TILE_STORE_REGS(sKV[0][1], tv);
    __syncthreads();
    FragAddr fa;
    fa.init(lds_addr(&sKV[0][0][0]), tid);
    for (int k0 = 0; k0 < len; k0 += 64) {
        const bool more = k0 + 64 < len;
        FS2_STAMP_AT(k0, 0);
        if (more) {                                       // next tile travels while this one is multiplied
            TILE_LOAD_REGS(tk, K, ld, k0 + 64, S - 1);
            TILE_LOAD_REGS(tv, V, ld, k0 + 64, S - 1);
        }
        // software pipeline over the 8 d-slices: the two K fragments of slice st+1 are in flight while slice st is
        // multiplied (two independent accumulator chains); sched_barrier pins the order - unpinned, the scheduler sinks
        // each read to just above its MFMA (r01i ISA: ds_read / s_waitcnt lgkmcnt(0) / v_mfma, 16 times in a row)
        FS2_STAMP_AT(k0, 1);
        f32x16 s[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) s[kb][r] = 0.f;
        {
            FS2_ATTN_PRIO(1);
            bf16x8 c0 = fa.rk(0, 0, 0), c1 = fa.rk(0, 1, 0);
#pragma unroll
            for (int st = 0; st < 8; ++st) {
                bf16x8 n0 = c0, n1 = c1;
                if (st < 7) { n0 = fa.rk(0, 0, st + 1); n1 = fa.rk(0, 1, st + 1); }
                __builtin_amdgcn_sched_barrier(0);
                s[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(c0, qf[st], s[0], 0, 0, 0);
                s[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(c1, qf[st], s[1], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                c0 = n0; c1 = n1;
            }
            FS2_ATTN_PRIO(0);
        }
        // Softmax bookkeeping in the log2 domain (p = exp2(s * scale * log2 e - m): one fma + one v_exp_f32 per score), the
        // key-padding mask only on the sequence's last tile, and a LAZY running maximum: the accumulators are rescaled only
        // when some query's maximum grew by more than 2^8 - after the first tiles that is almost never, and a rescale costs
        // 16 cross-lane permutes + 64 multiplies per wave and tile (r02s PMC: 15.7 VALU instructions per MFMA made this kernel
        // VALU-bound 2:1).  A stale maximum only scales p and l by the same factor <= 2^8: o / l is unchanged.
        FS2_STAMP_AT(k0, 2);
        if (k0 + 64 > len) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (k0 + kb * 32 + crow(r, h2) >= len) s[kb][r] = -INFINITY;
        }
        float mx = -INFINITY;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[kb][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64)) * sc2;                       // scale > 0: the maximum of the raw scores, scaled once
        if (__builtin_amdgcn_ballot_w64(mx > m + 8.f) != 0ull) {           // wave-uniform; always taken on the first tile (m = -inf)
            const float mn = fmaxf(m, mx);
            const float alpha = __builtin_amdgcn_exp2f(m - mn);
            l *= alpha;
            m = mn;
#pragma unroll
            for (int r = 0; r < 16; ++r) {               // (transposed accumulators: lane = query, so alpha is this lane's own)
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) o[nb][r] *= alpha;
            }
        }
        float rs = 0.f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[kb][r] = __builtin_amdgcn_exp2f(fmaf(s[kb][r], sc2, -m)); rs += s[kb][r]; }
        rs += __shfl_xor(rs, 32, 64);
        l += rs;
        FS2_STAMP_AT(k0, 3);
        {   // P V: the four V fragments of the next 16-key group are fetched while the current group is multiplied
            bf16x8 cv[4], nv[4];
            FS2_ATTN_PRIO(1);
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) cv[nb] = fa.rt(1, 0, nb);
#pragma unroll
            for (int g = 0; g < 4; ++g) {                 // g = 2*kb + u: keys 16g .. 16g+15 of the tile
                if (g < 3) {
#pragma unroll
                    for (int nb = 0; nb < 4; ++nb) nv[nb] = fa.rt(1, 16 * (g + 1), nb);
                }
                float pv[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) pv[e] = s[g >> 1][8 * (g & 1) + e];
                bf16x8 pa = pack8(pv);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) o[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cv[nb], pa, o[nb], 0, 0, 0);     // O^T[d][q] += V^T P^T
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) cv[nb] = nv[nb];
            }
            FS2_ATTN_PRIO(0);
        }
        FS2_STAMP_AT(k0, 4);
        if (more) {
            TILE_STORE_FA(fa, 0, tk);
            TILE_STORE_FA(fa, 1, tv);
        }
        fa.flip();
        FS2_STAMP_AT(k0, 5);
        __syncthreads();
        FS2_STAMP_AT(k0, 6);
    }
    float linv = l > 0.f ? 1.f / l : 0.f;
    if (h2 == 0 && myq < S) lse_o[myq] = (l > 0.f) ? m * 0.6931471805599453f + __logf(l) : 0.f;      // natural-log lse for the backward
    attn_store_row(out + (size_t)min(myq, S - 1) * ldo, o, linv, h2, myq < S);
}
#undef FS2_STAMP_KERNEL

// (Round 6 measured a SMALL-ITEM forward - 64 queries per two-wave workgroup, 32-key tiles, 32 KiB of LDS, four workgroups per CU - to
// even out the dealing of ~660 live items over the CU slots (a launch lasts two item lives for 1.3 items per slot): parity green, 70 us
// against 65 (profiles/r06xc_bench_attn.log).  Every K / V tile is staged by twice as many workgroups, there is a barrier per 32 keys and
// twice as many ~15 000-cycle cold starts: the per-item loss outweighs the better dealing.  Not kept; git history has the kernel.)

// dK, dV: one wave owns 32 keys (K, V fragments in registers); the block streams 64-query tiles of Q / dO.
// Per tile the log-sum-exp and delta values of its 64 queries travel with the Q / dO prefetch into LDS: the first
// version fetched them from GLOBAL memory element by element inside the `key_ok && q < len` branch - 32 dependent,
// individually waited loads per 32-query block (r01i ISA: global_load_dword / s_waitcnt vmcnt(0) / v_exp, x32), which is
// why dK/dV ran at 323 TF next to dQ's 724 TF.  The probability / score-gradient math is branch-free (invalid pairs are
// multiplied by an exact 0), and the fragment reads of each MFMA run are issued together in front of it (sched_barrier
// keeps the scheduler from sinking them back to their consumers).
__global__ void __launch_bounds__(256, 1) attn_bwd_dkv_bf16_kernel(const bf16_t* __restrict__ qkv, long ld,
                                                                   const bf16_t* __restrict__ dctx, long ldo,
                                                                   const float* __restrict__ lse, const float* __restrict__ delta,
                                                                   bf16_t* __restrict__ dqkv, const int32_t* __restrict__ lens,
                                                                   int S, int H, float scale) {
    __shared__ __attribute__((aligned(16))) unsigned char sQD[2][2][64 * 256];   // [buffer][Q | dO]
    __shared__ __attribute__((aligned(16))) float sLD[2][2][64];                 // [buffer][lse | delta] of the tile's queries
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, fl = lane & 31, h2 = lane >> 5;
    const int kt = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    const int len = lens ? min(lens[b], S) : S;
    const size_t rowbase = (size_t)b * S;
    const int kbase = kt * 128 + w * 32;
    bf16_t* dK = dqkv + rowbase * ld + (size_t)H * DK + h * DK;
    bf16_t* dV = dqkv + rowbase * ld + (size_t)2 * H * DK + h * DK;
    if (kt * 128 >= len) {
        for (int i = tid; i < 128 * 32; i += 256) {
            int r = i >> 5, c = (i & 31) * 4;
            if (kt * 128 + r < S) {
                st4<bf16_t>(dK + (size_t)(kt * 128 + r) * ld + c, make_float4(0.f, 0.f, 0.f, 0.f));
                st4<bf16_t>(dV + (size_t)(kt * 128 + r) * ld + c, make_float4(0.f, 0.f, 0.f, 0.f));
            }
        }
        return;
    }
    const bf16_t* Q = qkv + rowbase * ld + h * DK;
    const bf16_t* K = qkv + rowbase * ld + (size_t)H * DK + h * DK;
    const bf16_t* V = qkv + rowbase * ld + (size_t)2 * H * DK + h * DK;
    const bf16_t* dO = dctx + rowbase * ldo + h * DK;
    const float* lse_b = lse + ((size_t)b * H + h) * S;
    const float* del_b = delta + ((size_t)b * H + h) * S;
    const int mykey = kbase + fl;
    const float key_ok = mykey < len ? 1.f : 0.f;
    bf16x8 kf[8], vf[8];
    load_row_frags(kf, K + (size_t)min(mykey, S - 1) * ld, h2, mykey < S);
    load_row_frags(vf, V + (size_t)min(mykey, S - 1) * ld, h2, mykey < S);
    f32x16 dk[4], dv[4];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dk[nb][r] = 0.f; dv[nb][r] = 0.f; }

    // threads 0..63 carry lse, 64..127 delta of query (tile start + tid & 63); a query >= len gets lse = +inf -> p = 0
    const float* ld_src = tid < 64 ? lse_b : del_b;
    // (lse travels in the log2 domain: p = exp2(s * scale * log2 e - lse * log2 e) is one fma + one v_exp_f32 per score)
    const float sc2 = scale * 1.4426950408889634f;
    auto load_ld = [&](int q0) -> float {
        int q = q0 + (tid & 63);
        float v = ld_src[min(q, S - 1)];
        if (tid < 64) v *= 1.4426950408889634f;
        if (q >= len) v = tid < 64 ? INFINITY : 0.f;
        return v;
    };

    uint4 tq0, tq1, tq2, tq3, td0, td1, td2, td3;
    float tl = 0.f;
    TILE_LOAD_REGS(tq, Q, ld, 0, S - 1);
    TILE_LOAD_REGS(td, dO, ldo, 0, S - 1);
    if (tid < 128) tl = load_ld(0);
    TILE_STORE_REGS(sQD[0][0], tq);
    TILE_STORE_REGS(sQD[0][1], td);
    if (tid < 128) sLD[0][tid >> 6][tid & 63] = tl;
    __syncthreads();
    int buf = 0;
    for (int q0 = 0; q0 < len; q0 += 64, buf ^= 1) {
        const bool more = q0 + 64 < len;
        if (more) {
            TILE_LOAD_REGS(tq, Q, ld, q0 + 64, S - 1);
            TILE_LOAD_REGS(td, dO, ldo, q0 + 64, S - 1);
            if (tid < 128) tl = load_ld(q0 + 64);
        }
        const unsigned char* sQ = sQD[buf][0];
        const unsigned char* sdO = sQD[buf][1];
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            if (q0 + qb * 32 >= len) break;
            // ---- S^T = Q K^T and dP^T = dO V^T for 32 queries x my 32 keys, then dV += P^T dO and dK += dS^T Q.  The 32 LDS
            // fragments of a query block travel in four batches of eight through TWO register sets, each batch requested one
            // 8-MFMA run (256 cycles) ahead of its use.  r03m: with all 32 fragments requested up front (fq/fd/tO/tQ = 128
            // registers next to the 64 of K/V, the 32 prefetch registers and the softmax values) the kernel needed ~350 vector
            // registers; the compiler parked 96 of them in AGPRs and moved them with 337 v_accvgpr_read/write per 64 MFMAs -
            // 40 % of the loop's instructions at ~5 issue cycles each.
            const int rq = qb * 32 + fl;
            bf16x8 fa[4], fb[4], ga[4], gb[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { fa[i] = frag_k(sQ, rq, i, h2); fb[i] = frag_k(sdO, rq, i, h2); }
            // this lane's 16 queries: rows crow(r, h2) = (r&3) + 8*(r>>2) + 4*h2 -> four float4 per array
            float4 l4[4], d4[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                l4[g] = *reinterpret_cast<const float4*>(&sLD[buf][0][qb * 32 + 8 * g + 4 * h2]);
                d4[g] = *reinterpret_cast<const float4*>(&sLD[buf][1][qb * 32 + 8 * g + 4 * h2]);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) { ga[i] = frag_k(sQ, rq, 4 + i, h2); gb[i] = frag_k(sdO, rq, 4 + i, h2); }
            __builtin_amdgcn_sched_barrier(0);
            f32x16 s, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], kf[i], s, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[i], vf[i], dp, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) { fa[nb] = frag_t(sdO, qb * 32, nb, lane); fb[nb] = frag_t(sQ, qb * 32, nb, lane); }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ga[i], kf[4 + i], s, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gb[i], vf[4 + i], dp, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) { ga[nb] = frag_t(sdO, qb * 32 + 16, nb, lane); gb[nb] = frag_t(sQ, qb * 32 + 16, nb, lane); }
            __builtin_amdgcn_sched_barrier(0);
            float pv[16], dsv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float lq = reinterpret_cast<const float*>(&l4[r >> 2])[r & 3];
                const float dq_ = reinterpret_cast<const float*>(&d4[r >> 2])[r & 3];
                const float p = __builtin_amdgcn_exp2f(fmaf(s[r], sc2, -lq)) * key_ok;   // lq = +inf for q >= len -> exp2(-inf) = 0
                pv[r] = p;
                dsv[r] = p * (dp[r] - dq_);                                  // (the softmax scale multiplies dK once, at the store)
            }
            {
                const bf16x8 pa = pack8(pv), da = pack8(dsv);
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) {
                    dv[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa, fa[nb], dv[nb], 0, 0, 0);
                    dk[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(da, fb[nb], dk[nb], 0, 0, 0);
                }
            }
            {
                const bf16x8 pa = pack8(pv + 8), da = pack8(dsv + 8);
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) {
                    dv[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa, ga[nb], dv[nb], 0, 0, 0);
                    dk[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(da, gb[nb], dk[nb], 0, 0, 0);
                }
            }
        }
        if (more) {
            TILE_STORE_REGS(sQD[buf ^ 1][0], tq);
            TILE_STORE_REGS(sQD[buf ^ 1][1], td);
            if (tid < 128) sLD[buf ^ 1][tid >> 6][tid & 63] = tl;
        }
        // the eight accumulators are pinned to the accumulation registers across the loop edge: left to itself the allocator carried
        // them in VGPRs from one tile to the next and moved all 128 values into AGPRs and back around every tile's MFMAs
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) asm volatile("" : "+a"(dk[nb]), "+a"(dv[nb]));
        __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        int key = kbase + crow(r, h2);
        if (key < S) {
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) {
                dK[(size_t)key * ld + nb * 32 + fl] = f32_to_bf16(dk[nb][r] * scale);
                dV[(size_t)key * ld + nb * 32 + fl] = f32_to_bf16(dv[nb][r]);
            }
        }
    }
}

// ---- dK, dV, second form (round 6): the same tiles, fragments and arithmetic, SOFTWARE-PIPELINED inside the wave.
// The first form runs one wave per SIMD (128 accumulator registers + 64 of K / V fragments leave no room for a second), and that
// wave walked S^T / dP^T MFMAs -> exponentials -> dV / dK MFMAs strictly one after the other, each phase behind the LDS round
// trip of its fragments: 133 us per decoder layer at 17 % MFMA busy (r05zzz PMC) - 7 350 cycles per 64-query tile for 2 048 cycles
// of MFMA.  Here the two 32-query blocks of a tile are skewed by one phase, so that every VALU phase has an independent MFMA run
// to hide under and every fragment batch is requested one 8-MFMA run (256 cycles) before its use:
//     A0a | A0b | A1a + B0' | A1b + B0" | C0a + B1' | C0b + B1" | C1a | C1b          (A: S^T, dP^T; B: p, dS; C: dV, dK)
// The interleave inside a run is pinned with sched_group_barrier (one MFMA, then its share of the VALU / transcendental / LDS
// instructions); sched_barrier(0) separates the runs.  A block whose queries all lie beyond the sequence is multiplied anyway
// (its lse is +inf -> p = 0 exactly): no branch inside the tile.  Fragment addresses are lane constants (16 registers) with the
// buffer / array / row-block as immediate offsets - the tile loop is unrolled over the two LDS buffers.
#define FS2_SGB(mask, n) __builtin_amdgcn_sched_group_barrier((mask), (n), 0)
#define FS2_SG_MFMA 0x008
#define FS2_SG_VALU 0x002
#define FS2_SG_TRANS 0x400
#define FS2_SG_DSR 0x100
#define FS2_SG_DSW 0x200
#define FS2_STAMP_KERNEL 2
__global__ void __launch_bounds__(256, 1) attn_bwd_dkv2_bf16_kernel(const bf16_t* __restrict__ qkv, long ld,
                                                                    const bf16_t* __restrict__ dctx, long ldo,
                                                                    const float* __restrict__ lse, const float* __restrict__ delta,
                                                                    bf16_t* __restrict__ dqkv, const int32_t* __restrict__ lens,
                                                                    int S, int H, float scale) {
    __shared__ __attribute__((aligned(16))) unsigned char sQD[2][2][64 * 256];   // [buffer][Q | dO]
    __shared__ __attribute__((aligned(16))) float sLD[2][2][64];                 // [buffer][lse | delta] of the tile's queries
    FS2_STAMP_INIT();
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, fl = lane & 31, h2 = lane >> 5;
    int kt, h, b;
    attn_block_map(H, kt, h, b);
    const int len = lens ? min(lens[b], S) : S;
    const size_t rowbase = (size_t)b * S;
    const int kbase = kt * 128 + w * 32;
    bf16_t* dK = dqkv + rowbase * ld + (size_t)H * DK + h * DK;
    bf16_t* dV = dqkv + rowbase * ld + (size_t)2 * H * DK + h * DK;
    if (kt * 128 >= len) {
        for (int i = tid; i < 128 * 32; i += 256) {
            int r = i >> 5, c = (i & 31) * 4;
            if (kt * 128 + r < S) {
                st4<bf16_t>(dK + (size_t)(kt * 128 + r) * ld + c, make_float4(0.f, 0.f, 0.f, 0.f));
                st4<bf16_t>(dV + (size_t)(kt * 128 + r) * ld + c, make_float4(0.f, 0.f, 0.f, 0.f));
            }
        }
        return;
    }
    const bf16_t* Q = qkv + rowbase * ld + h * DK;
    const bf16_t* K = qkv + rowbase * ld + (size_t)H * DK + h * DK;
    const bf16_t* V = qkv + rowbase * ld + (size_t)2 * H * DK + h * DK;
    const bf16_t* dO = dctx + rowbase * ldo + h * DK;
    const float* lse_b = lse + ((size_t)b * H + h) * S;
    const float* del_b = delta + ((size_t)b * H + h) * S;
    const int mykey = kbase + fl;
    bf16x8 kf[8], vf[8];
    load_row_frags(kf, K + (size_t)min(mykey, S - 1) * ld, h2, mykey < S);
    load_row_frags(vf, V + (size_t)min(mykey, S - 1) * ld, h2, mykey < S);
    f32x16 dk[4], dv[4];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dk[nb][r] = 0.f; dv[nb][r] = 0.f; }
    const float* ld_src = tid < 64 ? lse_b : del_b;
    const float sc2 = scale * 1.4426950408889634f;
    // the raw value is LOADED at the top of a tile and CONVERTED where it is stored (end of the tile): the first form multiplied it
    // right behind the load - an s_waitcnt vmcnt(0) at the top of every tile, which also waited for the tile's own prefetch: a full
    // memory round trip per tile in front of everything else (r06n PMC: 41 % of the wave time parked)
    // (an asm load: the compiler sinks a plain load down to its use - issued late, waited for at once; nothing but the explicit
    // s_waitcnt vmcnt(0) in front of the tile's closing barrier may wait on vector memory inside the loop, because the counter also
    // holds the tile's LDS-DMA pieces)
    auto load_ld = [&](int q0) -> float {
        float v;
        const float* pa_ = ld_src + min(q0 + (tid & 63), S - 1);
        asm volatile("global_load_dword %0, %1, off" : "=v"(v) : "v"(pa_) : "memory");
        return v;
    };
    auto conv_ld = [&](float v, int q0) -> float {
        const int q = q0 + (tid & 63);
        if (tid < 64) v *= 1.4426950408889634f;
        if (q >= len) v = tid < 64 ? INFINITY : 0.f;
        return v;
    };
    // fragment addresses: LDS byte addresses of this lane's pieces in the CURRENT buffer's Q image (the dO image, the row block and the
    // d-slice enter as immediate offsets); they flip between the two buffers by +- 32 KiB at the end of every tile (16 adds per tile
    // instead of ~80 address instructions per tile in the first form)
    const unsigned lds_qd = lds_addr(&sQD[0][0][0]);
    unsigned ak[8], at0[4], at1[4];
#pragma unroll
    for (int i = 0; i < 8; ++i) ak[i] = lds_qd + (unsigned)swzb(fl, 2 * i + h2);
    {
        const int li = lane & 15, g = lane >> 4, hh = g >> 1, rr = 4 * hh + (li >> 2);
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
            const int col = nb * 32 + 16 * (g & 1) + 4 * (li & 3);
            at0[nb] = lds_qd + (unsigned)(swzb(rr, col >> 3) + ((col & 7) << 1));
            at1[nb] = lds_qd + (unsigned)(swzb(rr + 8, col >> 3) + ((col & 7) << 1));
        }
    }
    typedef __attribute__((address_space(3))) const uint4* lds_u4p;
    // arr: 0 = Q, 1 = dO; qb: 32-query block; i: d-slice
    auto rdk = [&](int arr, int qb, int i) -> bf16x8 {
        return __builtin_bit_cast(bf16x8, *(lds_u4p)(size_t)(ak[i] + (unsigned)(arr * 16384 + qb * 8192)));
    };
    auto rdt = [&](int arr, int rb, int nb) -> bf16x8 {                            // rb: first of the 16 rows (a multiple of 16)
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4p)(size_t)(at0[nb] + (unsigned)(arr * 16384 + rb * 256)));
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4p)(size_t)(at1[nb] + (unsigned)(arr * 16384 + rb * 256)));
        return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    };
    // Tile staging by LDS-DMA (no staging registers, no ds_write pass): a 1 KiB piece = 4 tile rows, lane l -> row l >> 4, LDS chunk
    // position l & 15, which must hold global chunk (l & 15) ^ S(row) (the swizzle is an involution on the chunk index); wave w moves
    // pieces w, w + 4, w + 8, w + 12 of Q and of dO.  Rows are clamped to the sequence's last row (finite data times an exact 0).
    unsigned voq[4], vod[4];            // per piece: byte offset of this lane's 16 bytes relative to (tile row 0) - the row part is added per tile
    int prow[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = 4 * (w + 4 * j) + (lane >> 4);
        const int c = (lane & 15) ^ (((row & 3) << 2) | ((row >> 2) & 3));
        prow[j] = row;
        voq[j] = (unsigned)c * 16u; vod[j] = (unsigned)c * 16u;
    }
    auto dma_piece = [&](int q0, int buf, int j) {          // piece w + 4 j of Q and of dO
        const unsigned r = (unsigned)min(q0 + prow[j], S - 1);
        const unsigned dst = __builtin_amdgcn_readfirstlane(lds_qd + (unsigned)(buf * 32768 + (w + 4 * j) * 1024));
        glds16_sbase(r * (unsigned)(ld * 2) + voq[j], Q, dst);
        glds16_sbase(r * (unsigned)(ldo * 2) + vod[j], dO, dst + 16384u);
    };
    auto dma_tile = [&](int q0, int buf) {
#pragma unroll
        for (int j = 0; j < 4; ++j) dma_piece(q0, buf, j);
    };
    float tl = 0.f;
    dma_tile(0, 0);
    tl = load_ld(0);
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(tl) :: "memory");
    if (tid < 128) sLD[0][tid >> 6][tid & 63] = conv_ld(tl, 0);
    __syncthreads();

    // p, dS of eight (query, my key) pairs -> packed A operands (lse / delta of the lane's queries come from LDS where they are used)
    auto softmax8 = [&](const f32x16& s, const f32x16& dp, const float* sl, const float* sd, int qb, int u, bf16x8& pa, bf16x8& da) {
        float4 l4[2], d4[2];
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            l4[g] = *reinterpret_cast<const float4*>(sl + qb * 32 + 8 * (2 * u + g) + 4 * h2);
            d4[g] = *reinterpret_cast<const float4*>(sd + qb * 32 + 8 * (2 * u + g) + 4 * h2);
        }
        float pv[8], dsv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float lq = reinterpret_cast<const float*>(&l4[e >> 2])[e & 3];
            const float dq_ = reinterpret_cast<const float*>(&d4[e >> 2])[e & 3];
            const float p = __builtin_amdgcn_exp2f(fmaf(s[8 * u + e], sc2, -lq));     // (a padded KEY only pollutes its own dK / dV rows: zeroed at the store)
            pv[e] = p;
            dsv[e] = p * (dp[8 * u + e] - dq_);
        }
        pa = pack8(pv); da = pack8(dsv);
        // (the compiler sinks part of this arithmetic towards its first use, the dV / dK run two phases later; pinning the packed operands
        // here with an empty asm measured SLOWER - 191.7 vs 185 us for the whole backward, r06m - so the placement is left to it)
    };

    // dK / dV stay in their accumulation registers at every phase boundary (left alone, the allocator lent 32 of them to the score
    // accumulators and moved them out and back with ~107 v_accvgpr copies per tile)
#define FS2_PIN_ACC() asm volatile("" : "+a"(dk[0]), "+a"(dk[1]), "+a"(dk[2]), "+a"(dk[3]), "+a"(dv[0]), "+a"(dv[1]), "+a"(dv[2]), "+a"(dv[3]))
    // K / V fragments: landed, and known to the compiler's wait-count pass to have landed, before the loop (otherwise it waits for
    // them - vmcnt(0), i.e. for the tile's DMA too - in front of the second MFMA of EVERY tile)
#pragma unroll
    for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(kf[i]), "+v"(vf[i]));
    int buf = 0, flip = 32768;
    for (int q0 = 0; q0 < len; q0 += 64, buf ^= 1) {
        const bool more = q0 + 64 < len;
        FS2_STAMP(0);
        constexpr int sQ = 0, sdO = 1;
        const float* sl = sLD[buf][0];
        const float* sd = sLD[buf][1];
        bf16x8 x0q[4], x0d[4], y0q[4], y0d[4], x1q[4], x1d[4], y1q[4], y1d[4];      // K-contiguous batches: block 0 / 1, d-slices 0-3 / 4-7
        bf16x8 t0d[4], t0q[4], u0d[4], u0q[4], t1d[4], t1q[4], u1d[4], u1q[4];      // transposed batches: block 0 / 1, queries 0-15 / 16-31
        f32x16 s0, p0, s1, p1;
        bf16x8 pa0a, da0a, pa0b, da0b, pa1a, da1a, pa1b, da1b;
        // the tile's first fragments are requested FIRST (nothing else can hide their round trip right behind the barrier); the next
        // tile's DMA pieces (the other buffer's last readers passed that barrier) and its lse / delta load follow one per MFMA pair
#pragma unroll
        for (int i = 0; i < 4; ++i) { x0q[i] = rdk(sQ, 0, i); x0d[i] = rdk(sdO, 0, i); }
#pragma unroll
        for (int i = 0; i < 4; ++i) { y0q[i] = rdk(sQ, 0, 4 + i); y0d[i] = rdk(sdO, 0, 4 + i); }
        tl = load_ld(q0 + 64);                            // unconditional (clamped address): a load inside a branch is waited for at the join
#pragma unroll
        for (int r = 0; r < 16; ++r) { s0[r] = 0.f; p0[r] = 0.f; }
        FS2_PIN_ACC(); __builtin_amdgcn_sched_barrier(0);
        // ---- A0a | fetch X1a | DMA pieces 0, 1
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x0q[i], kf[i], s0, 0, 0, 0);
            p0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x0d[i], vf[i], p0, 0, 0, 0);
            x1q[i] = rdk(sQ, 1, i); x1d[i] = rdk(sdO, 1, i);
            if ((i & 1) && more) dma_piece(q0 + 64, buf ^ 1, i >> 1);
            __builtin_amdgcn_sched_barrier(0);
        }
        FS2_STAMP(1);
        // ---- A0b | fetch X1b | DMA pieces 2, 3
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(y0q[i], kf[4 + i], s0, 0, 0, 0);
            p0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(y0d[i], vf[4 + i], p0, 0, 0, 0);
            y1q[i] = rdk(sQ, 1, 4 + i); y1d[i] = rdk(sdO, 1, 4 + i);
            if ((i & 1) && more) dma_piece(q0 + 64, buf ^ 1, 2 + (i >> 1));
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) { s1[r] = 0.f; p1[r] = 0.f; }
        FS2_PIN_ACC(); __builtin_amdgcn_sched_barrier(0);
        FS2_STAMP(2);
        // ---- A1a | p, dS of block 0, queries 0-15 | fetch T0a
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x1q[i], kf[i], s1, 0, 0, 0);
            p1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x1d[i], vf[i], p1, 0, 0, 0);
        }
        softmax8(s0, p0, sl, sd, 0, 0, pa0a, da0a);
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) { t0d[nb] = rdt(sdO, 0, nb); t0q[nb] = rdt(sQ, 0, nb); }
#pragma unroll
        for (int k = 0; k < 8; ++k) { FS2_SGB(FS2_SG_MFMA, 1); FS2_SGB(FS2_SG_DSR, 2); FS2_SGB(FS2_SG_VALU, 6); FS2_SGB(FS2_SG_TRANS, 1); }
        FS2_PIN_ACC(); __builtin_amdgcn_sched_barrier(0);
        FS2_STAMP(3);
        // ---- A1b | p, dS of block 0, queries 16-31 | fetch T0b
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(y1q[i], kf[4 + i], s1, 0, 0, 0);
            p1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(y1d[i], vf[4 + i], p1, 0, 0, 0);
        }
        softmax8(s0, p0, sl, sd, 0, 1, pa0b, da0b);
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) { u0d[nb] = rdt(sdO, 16, nb); u0q[nb] = rdt(sQ, 16, nb); }
#pragma unroll
        for (int k = 0; k < 8; ++k) { FS2_SGB(FS2_SG_MFMA, 1); FS2_SGB(FS2_SG_DSR, 2); FS2_SGB(FS2_SG_VALU, 6); FS2_SGB(FS2_SG_TRANS, 1); }
        FS2_PIN_ACC(); __builtin_amdgcn_sched_barrier(0);
        FS2_STAMP(4);
        // ---- C0a | p, dS of block 1, queries 0-15 | fetch T1a
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
            dv[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(t0d[nb], pa0a, dv[nb], 0, 0, 0);
            dk[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(t0q[nb], da0a, dk[nb], 0, 0, 0);
        }
        softmax8(s1, p1, sl, sd, 1, 0, pa1a, da1a);
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) { t1d[nb] = rdt(sdO, 32, nb); t1q[nb] = rdt(sQ, 32, nb); }
#pragma unroll
        for (int k = 0; k < 8; ++k) { FS2_SGB(FS2_SG_MFMA, 1); FS2_SGB(FS2_SG_DSR, 2); FS2_SGB(FS2_SG_VALU, 6); FS2_SGB(FS2_SG_TRANS, 1); }
        FS2_PIN_ACC(); __builtin_amdgcn_sched_barrier(0);
        FS2_STAMP(5);
        // ---- C0b | p, dS of block 1, queries 16-31 | fetch T1b
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
            dv[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(u0d[nb], pa0b, dv[nb], 0, 0, 0);
            dk[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(u0q[nb], da0b, dk[nb], 0, 0, 0);
        }
        softmax8(s1, p1, sl, sd, 1, 1, pa1b, da1b);
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) { u1d[nb] = rdt(sdO, 48, nb); u1q[nb] = rdt(sQ, 48, nb); }
#pragma unroll
        for (int k = 0; k < 8; ++k) { FS2_SGB(FS2_SG_MFMA, 1); FS2_SGB(FS2_SG_DSR, 2); FS2_SGB(FS2_SG_VALU, 6); FS2_SGB(FS2_SG_TRANS, 1); }
        FS2_PIN_ACC(); __builtin_amdgcn_sched_barrier(0);
        FS2_STAMP(6);
        // ---- C1a, C1b
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
            dv[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(t1d[nb], pa1a, dv[nb], 0, 0, 0);
            dk[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(t1q[nb], da1a, dk[nb], 0, 0, 0);
        }
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
            dv[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(u1d[nb], pa1b, dv[nb], 0, 0, 0);
            dk[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(u1q[nb], da1b, dk[nb], 0, 0, 0);
        }
        FS2_STAMP(7);
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(tl) :: "memory");      // my pieces of the next tile and its lse / delta value have landed
        FS2_STAMP(8);
        if (tid < 128) sLD[buf ^ 1][tid >> 6][tid & 63] = conv_ld(tl, q0 + 64);       // (after the last tile: into a buffer nobody reads)
#pragma unroll
        for (int i = 0; i < 8; ++i) ak[i] += (unsigned)flip;
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) { at0[nb] += (unsigned)flip; at1[nb] += (unsigned)flip; }
        flip = -flip;
        FS2_PIN_ACC();
        FS2_STAMP(9);
        __syncthreads();
        FS2_STAMP(10);
    }
    // (transposed accumulators dK^T / dV^T [d][key]: lane = key)
    {
        const float kz = mykey < len ? 1.f : 0.f;           // a padded key's rows come out as zeros
        attn_store_row(dK + (size_t)min(mykey, S - 1) * ld, dk, scale * kz, h2, mykey < S);
        attn_store_row(dV + (size_t)min(mykey, S - 1) * ld, dv, kz, h2, mykey < S);
    }
}
#undef FS2_STAMP_KERNEL
// (also computes delta[q] = sum_d dO[q][d] O[q][d] for its queries - a lane already holds half of its query's dO row - and
// writes it for the dK/dV kernel, which therefore runs AFTER this one: the separate delta launch, 12 us per layer, is gone)
#define FS2_STAMP_KERNEL 1
__global__ void __launch_bounds__(256, 2) attn_bwd_dq_bf16_kernel(const bf16_t* __restrict__ qkv, long ld,
                                                                  const bf16_t* __restrict__ ctx,
                                                                  const bf16_t* __restrict__ dctx, long ldo,
                                                                  const float* __restrict__ lse, float* __restrict__ delta,
                                                                  bf16_t* __restrict__ dqkv, const int32_t* __restrict__ lens,
                                                                  int S, int H, float scale) {
    __shared__ __attribute__((aligned(16))) unsigned char sKV[2][2][64 * 256];   // [buffer][K | V]
    FS2_ATTN_PRIO_INIT();
    FS2_STAMP_INIT();
    FS2_STAMP_RAW(0);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, fl = lane & 31, h2 = lane >> 5;
    int qt, h, b;
    attn_block_map(H, qt, h, b);
    const int len = lens ? min(lens[b], S) : S;
    const int q0 = qt * 128;
    const size_t rowbase = (size_t)b * S;
    bf16_t* dQ = dqkv + rowbase * ld + h * DK;
    if (q0 >= len) {
        for (int i = tid; i < 128 * 32; i += 256) {
            int r = i >> 5, c = (i & 31) * 4;
            if (q0 + r < S) st4<bf16_t>(dQ + (size_t)(q0 + r) * ld + c, make_float4(0.f, 0.f, 0.f, 0.f));
        }
        if (tid < 128 && q0 + tid < S) delta[((size_t)b * H + h) * S + q0 + tid] = 0.f;      // (never read: queries >= len)
        return;
    }
    const bf16_t* Q = qkv + rowbase * ld + h * DK;
    const bf16_t* K = qkv + rowbase * ld + (size_t)H * DK + h * DK;
    const bf16_t* V = qkv + rowbase * ld + (size_t)2 * H * DK + h * DK;
    const bf16_t* dO = dctx + rowbase * ldo + h * DK;
    const int myq = q0 + w * 32 + fl;
    const bool q_ok = myq < len;
    float my_lse = 0.f, my_del = 0.f;
    if (myq < S) my_lse = lse[((size_t)b * H + h) * S + myq];
    // log2 domain (one fma + one v_exp_f32 per score); a padded query gets lse = +inf -> p = 0 without a mask multiply
    const float sc2 = scale * 1.4426950408889634f;
    const float my_lse2 = q_ok ? my_lse * 1.4426950408889634f : INFINITY;
    bf16x8 qf[8], df[8];
    load_row_frags(qf, Q + (size_t)min(myq, S - 1) * ld, h2, myq < S);
    load_row_frags(df, dO + (size_t)min(myq, S - 1) * ldo, h2, myq < S);
    {   // delta of my query from the dO fragments just loaded (this lane: d = 16 s + 8 h2 + e; the other half-wave has the rest)
        bf16x8 of[8];
        load_row_frags(of, ctx + (rowbase + (size_t)min(myq, S - 1)) * ldo + h * DK, h2, myq < S);
        float acc = 0.f;
#pragma unroll
        for (int st = 0; st < 8; ++st) {
            const uint4 ov = __builtin_bit_cast(uint4, of[st]), dv_ = __builtin_bit_cast(uint4, df[st]);
            const uint32_t* ou = reinterpret_cast<const uint32_t*>(&ov);
            const uint32_t* du = reinterpret_cast<const uint32_t*>(&dv_);
#pragma unroll
            for (int e = 0; e < 4; ++e)
                acc += __uint_as_float(ou[e] << 16) * __uint_as_float(du[e] << 16) + __uint_as_float(ou[e] & 0xffff0000u) * __uint_as_float(du[e] & 0xffff0000u);
        }
        acc += __shfl_xor(acc, 32, 64);
        my_del = acc;                                        // (rows >= S load zeros)
        if (h2 == 0 && myq < S) delta[((size_t)b * H + h) * S + myq] = my_del;
    }
    f32x16 dq[4];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[nb][r] = 0.f;

    uint4 tk0, tk1, tk2, tk3, tv0, tv1, tv2, tv3;
    TILE_LOAD_REGS(tk, K, ld, 0, S - 1);
    TILE_LOAD_REGS(tv, V, ld, 0, S - 1);
    TILE_STORE_REGS(sKV[0][0], tk);
    TILE_STORE_REGS(sKV[0][1], tv);
    __syncthreads();
    FragAddr fa;
    fa.init(lds_addr(&sKV[0][0][0]), tid);
    FS2_STAMP_RAW(1);
    for (int k0 = 0; k0 < len; k0 += 64) {
        const bool more = k0 + 64 < len;
        FS2_STAMP_AT(k0 + 64, 0);
        if (more) {
            TILE_LOAD_REGS(tk, K, ld, k0 + 64, S - 1);
            TILE_LOAD_REGS(tv, V, ld, k0 + 64, S - 1);
        }
        FS2_STAMP_AT(k0 + 64, 1);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            if (k0 + kb * 32 >= len) break;
            // software pipeline over the 8 d-slices (K and V fragments of slice st+1 in flight while slice st is multiplied);
            // sched_barrier pins it (hoisting the 8 transposed K fragments too spills at 2 waves per SIMD)
            f32x16 s, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
            {
                FS2_ATTN_PRIO(1);
                bf16x8 ck = fa.rk(0, kb, 0), cv = fa.rk(1, kb, 0);
#pragma unroll
                for (int st = 0; st < 8; ++st) {
                    bf16x8 nk = ck, nv = cv;
                    if (st < 7) { nk = fa.rk(0, kb, st + 1); nv = fa.rk(1, kb, st + 1); }
                    __builtin_amdgcn_sched_barrier(0);
                    s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ck, qf[st], s, 0, 0, 0);
                    dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cv, df[st], dp, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    ck = nk; cv = nv;
                }
                FS2_ATTN_PRIO(0);
            }
            FS2_STAMP_AT(k0 + 64, 2 + 3 * kb);
            float dsv[16];
            // the only key block with padded keys gets its scores masked BEFORE the exponential, behind a wave-uniform branch (as a
            // per-element select in the common loop it was 94 of the loop's 310 VALU instructions: r03m instruction mix)
            if (k0 + kb * 32 + 32 > len) {
                int thr = len - k0 - kb * 32 - 4 * h2;
                asm volatile("" : "+v"(thr));
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if ((r & 3) + 8 * (r >> 2) >= thr) s[r] = -INFINITY;           // exp2(-inf) = 0
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = __builtin_amdgcn_exp2f(fmaf(s[r], sc2, -my_lse2));
                dsv[r] = p * (dp[r] - my_del);                                  // (the softmax scale multiplies dQ once, at the store)
            }
            FS2_STAMP_AT(k0 + 64, 3 + 3 * kb);
            FS2_ATTN_PRIO(1);
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                bf16x8 da = pack8(dsv + 8 * u);
#pragma unroll
                for (int nb = 0; nb < 4; ++nb)
                    dq[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa.rt(0, kb * 32 + 16 * u, nb), da, dq[nb], 0, 0, 0);       // dQ^T[d][q] += K^T dS^T
            }
            FS2_ATTN_PRIO(0);
            FS2_STAMP_AT(k0 + 64, 4 + 3 * kb);
        }
        if (more) {
            TILE_STORE_FA(fa, 0, tk);
            TILE_STORE_FA(fa, 1, tv);
        }
        fa.flip();
        FS2_STAMP_AT(k0 + 64, 8);
        __syncthreads();
        FS2_STAMP_AT(k0 + 64, 9);
    }
    attn_store_row(dQ + (size_t)min(myq, S - 1) * ld, dq, scale, h2, myq < S);
}
#undef FS2_STAMP_KERNEL

extern "C" int fs2_attn_fwd(const void* qkv, void* ctx, float* lse, const int32_t* lens, int B, int S, int H, int dk,
                            float scale, int dtype, hipStream_t stream) {
    FS2_CHECK_ARG(qkv && ctx && lse, "attn_fwd: null pointer");
    FS2_CHECK_ARG(dk == DK, "attn_fwd: only d_k = 128 is supported (got %d)", dk);
    FS2_CHECK_ARG(B >= 0 && S > 0 && H > 0, "attn_fwd: bad shape");
    if (B == 0) return FS2_OK;
    dim3 grid(fs2_cdiv(S, 128), H, B);
    long ld = 3L * H * DK, ldo = (long)H * DK;
    if (dtype == FS2_F32) attn_fwd_kernel<float><<<grid, 256, 0, stream>>>((const float*)qkv, ld, (float*)ctx, ldo, lse, lens, S, H, scale);
    else if (dtype == FS2_BF16) attn_fwd_bf16_kernel<<<grid, 256, 0, stream>>>((const bf16_t*)qkv, ld, (bf16_t*)ctx, ldo, lse, lens, S, H, scale);
    else { fs2_set_error("attn_fwd: dtype"); return FS2_EDTYPE; }
    FS2_CHECK_LAUNCH("attn_fwd");
    return FS2_OK;
}

// ------------------------------------------------------------------ backward, part 0: D[b,h,q] = sum_d dO*O
template <typename T>
__global__ void attn_delta_kernel(const T* __restrict__ ctx, const T* __restrict__ dctx, float* __restrict__ delta, int rows,
                                  int S, int H) {
    int idx = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);  // idx = row*H + h
    if (idx >= rows * H) return;
    int lane = threadIdx.x & 63;
    int row = idx / H, h = idx - row * H;
    size_t off = (size_t)row * H * DK + h * DK + lane * 2;
    float s = Elem<T>::ld(ctx + off) * Elem<T>::ld(dctx + off) + Elem<T>::ld(ctx + off + 1) * Elem<T>::ld(dctx + off + 1);
    s = wave_sum(s);
    if (lane == 0) {
        int b = row / S, q = row - b * S;
        delta[((size_t)b * H + h) * S + q] = s;
    }
}

// ------------------------------------------------------------------ backward, part 1: dK, dV
// One wave owns 32 keys; the block's 4 waves share the streamed Q / dO tiles.
// dynamic LDS: [4 waves][K 16K | V 16K] + Q 16K + dO 16K = 160 KiB.
template <typename T>
__global__ void __launch_bounds__(256, 1) attn_bwd_dkv_kernel(const T* __restrict__ qkv, long ld, const T* __restrict__ dctx,
                                                              long ldo, const float* __restrict__ lse,
                                                              const float* __restrict__ delta, T* __restrict__ dqkv,
                                                              const int32_t* __restrict__ lens, int S, int H, float scale) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, fl = lane & 31, h2 = lane >> 5;
    const int kt = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    const int len = lens ? min(lens[b], S) : S;
    const size_t rowbase = (size_t)b * S;
    const int kbase = kt * 128 + w * 32;
    T* dK = dqkv + rowbase * ld + (size_t)H * DK + h * DK;
    T* dV = dqkv + rowbase * ld + (size_t)2 * H * DK + h * DK;
    if (kt * 128 >= len) {   // all keys of this block are padding: gradients are zero
        for (int i = tid; i < 128 * 32; i += 256) {
            int r = i >> 5, c = (i & 31) * 4;
            if (kt * 128 + r < S) {
                st4<T>(dK + (size_t)(kt * 128 + r) * ld + c, make_float4(0.f, 0.f, 0.f, 0.f));
                st4<T>(dV + (size_t)(kt * 128 + r) * ld + c, make_float4(0.f, 0.f, 0.f, 0.f));
            }
        }
        return;
    }
    unsigned char* sKw = smem + w * 32768;
    unsigned char* sVw = sKw + 16384;
    unsigned char* sQ = smem + 131072;
    unsigned char* sdO = sQ + 16384;
    const T* Q = qkv + rowbase * ld + h * DK;
    const T* K = qkv + rowbase * ld + (size_t)H * DK + h * DK;
    const T* V = qkv + rowbase * ld + (size_t)2 * H * DK + h * DK;
    const T* dO = dctx + rowbase * ldo + h * DK;
    const float* lse_b = lse + ((size_t)b * H + h) * S;
    const float* del_b = delta + ((size_t)b * H + h) * S;
    // each wave stages its own K / V sub-tiles (64 threads)
    stage_tile<T, 64>(sKw, K, ld, kbase, max(0, min(32, S - kbase)), lane);
    stage_tile<T, 64>(sVw, V, ld, kbase, max(0, min(32, S - kbase)), lane);
    const bool key_ok = (kbase + fl) < len;

    f32x16 dk[4], dv[4];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dk[nb][r] = 0.f; dv[nb][r] = 0.f; }

    for (int q0 = 0; q0 < len; q0 += 32) {
        __syncthreads();
        stage_tile<T, 256>(sQ, Q, ld, q0, min(32, S - q0), tid);
        stage_tile<T, 256>(sdO, dO, ldo, q0, min(32, S - q0), tid);
        __syncthreads();
        f32x16 s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
        // S[q][key] : A = Q (lane row = q), B = K^T (lane col = key)
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            float4 qa = *reinterpret_cast<const float4*>(sQ + swz(fl, h2 * 16 + u));
            float4 kb = *reinterpret_cast<const float4*>(sKw + swz(fl, h2 * 16 + u));
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(qa.x, kb.x, s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(qa.y, kb.y, s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(qa.z, kb.z, s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(qa.w, kb.w, s, 0, 0, 0);
        }
        // dP[q][key] : A = dO, B = V^T
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            float4 da = *reinterpret_cast<const float4*>(sdO + swz(fl, h2 * 16 + u));
            float4 vb = *reinterpret_cast<const float4*>(sVw + swz(fl, h2 * 16 + u));
            dp = __builtin_amdgcn_mfma_f32_32x32x2f32(da.x, vb.x, dp, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x2f32(da.y, vb.y, dp, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x2f32(da.z, vb.z, dp, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x2f32(da.w, vb.w, dp, 0, 0, 0);
        }
        // lane = key (col), reg r = query row crow(r,h2)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            int q = q0 + crow(r, h2);
            float p = 0.f, ds = 0.f;
            if (key_ok && q < len) {
                p = __expf(s[r] * scale - lse_b[q]);
                ds = p * (dp[r] - del_b[q]) * scale;
            }
            s[r] = p; dp[r] = ds;
        }
        // dV[key][d] += P^T dO ; dK[key][d] += dS^T Q     (A = regs, lane row = key, k-slot = q)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            int qr = crow(r, h2);
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) {
                int col = nb * 32 + fl;
                float ob = *reinterpret_cast<const float*>(sdO + swz(qr, col >> 2) + (col & 3) * 4);
                float qb = *reinterpret_cast<const float*>(sQ + swz(qr, col >> 2) + (col & 3) * 4);
                dv[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(s[r], ob, dv[nb], 0, 0, 0);
                dk[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(dp[r], qb, dk[nb], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        int key = kbase + crow(r, h2);
        if (key < S) {
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) {
                Elem<T>::st(dK + (size_t)key * ld + nb * 32 + fl, dk[nb][r]);
                Elem<T>::st(dV + (size_t)key * ld + nb * 32 + fl, dv[nb][r]);
            }
        }
    }
}

// ------------------------------------------------------------------ backward, part 2: dQ
template <typename T>
__global__ void __launch_bounds__(256, 1) attn_bwd_dq_kernel(const T* __restrict__ qkv, long ld, const T* __restrict__ dctx,
                                                             long ldo, const float* __restrict__ lse,
                                                             const float* __restrict__ delta, T* __restrict__ dqkv,
                                                             const int32_t* __restrict__ lens, int S, int H, float scale) {
    __shared__ __attribute__((aligned(16))) unsigned char sK[32 * 512];
    __shared__ __attribute__((aligned(16))) unsigned char sV[32 * 512];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, fl = lane & 31, h2 = lane >> 5;
    const int qt = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    const int len = lens ? min(lens[b], S) : S;
    const int q0 = qt * 128;
    const size_t rowbase = (size_t)b * S;
    T* dQ = dqkv + rowbase * ld + h * DK;
    if (q0 >= len) {
        for (int i = tid; i < 128 * 32; i += 256) {
            int r = i >> 5, c = (i & 31) * 4;
            if (q0 + r < S) st4<T>(dQ + (size_t)(q0 + r) * ld + c, make_float4(0.f, 0.f, 0.f, 0.f));
        }
        return;
    }
    const T* Q = qkv + rowbase * ld + h * DK;
    const T* K = qkv + rowbase * ld + (size_t)H * DK + h * DK;
    const T* V = qkv + rowbase * ld + (size_t)2 * H * DK + h * DK;
    const T* dO = dctx + rowbase * ldo + h * DK;
    const int myq = q0 + w * 32 + fl;
    const bool q_ok = myq < len;
    float my_lse = 0.f, my_del = 0.f;
    if (myq < S) { my_lse = lse[((size_t)b * H + h) * S + myq]; my_del = delta[((size_t)b * H + h) * S + myq]; }
    float qf[64], df[64];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f), d = v;
        if (myq < S) { v = ld4<T>(Q + (size_t)myq * ld + h2 * 64 + u * 4); d = ld4<T>(dO + (size_t)myq * ldo + h2 * 64 + u * 4); }
        qf[u * 4 + 0] = v.x; qf[u * 4 + 1] = v.y; qf[u * 4 + 2] = v.z; qf[u * 4 + 3] = v.w;
        df[u * 4 + 0] = d.x; df[u * 4 + 1] = d.y; df[u * 4 + 2] = d.z; df[u * 4 + 3] = d.w;
    }
    f32x16 dq[4];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[nb][r] = 0.f;

    for (int k0 = 0; k0 < len; k0 += 32) {
        __syncthreads();
        stage_tile<T, 256>(sK, K, ld, k0, min(32, S - k0), tid);
        stage_tile<T, 256>(sV, V, ld, k0, min(32, S - k0), tid);
        __syncthreads();
        f32x16 s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
        // S^T[key][q] : A = K (lane row = key), B = Q^T (lane col = q) ; dP^T[key][q] : A = V, B = dO^T
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            float4 ka = *reinterpret_cast<const float4*>(sK + swz(fl, h2 * 16 + u));
            float4 va = *reinterpret_cast<const float4*>(sV + swz(fl, h2 * 16 + u));
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(ka.x, qf[u * 4 + 0], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(ka.y, qf[u * 4 + 1], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(ka.z, qf[u * 4 + 2], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(ka.w, qf[u * 4 + 3], s, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x2f32(va.x, df[u * 4 + 0], dp, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x2f32(va.y, df[u * 4 + 1], dp, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x2f32(va.z, df[u * 4 + 2], dp, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x2f32(va.w, df[u * 4 + 3], dp, 0, 0, 0);
        }
        // lane = q (col), reg r = key row crow(r,h2)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            int key = k0 + crow(r, h2);
            float ds = 0.f;
            if (q_ok && key < len) {
                float p = __expf(s[r] * scale - my_lse);
                ds = p * (dp[r] - my_del) * scale;
            }
            s[r] = ds;
        }
        // dQ[q][d] += dS K   (A = regs: lane row = q, k-slot = key)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            int kr = crow(r, h2);
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) {
                int col = nb * 32 + fl;
                float kb = *reinterpret_cast<const float*>(sK + swz(kr, col >> 2) + (col & 3) * 4);
                dq[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(s[r], kb, dq[nb], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        int q = q0 + w * 32 + crow(r, h2);
        if (q < S) {
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) Elem<T>::st(dQ + (size_t)q * ld + nb * 32 + fl, dq[nb][r]);
        }
    }
}

extern "C" int fs2_attn_bwd(const void* qkv, const void* ctx, const void* dctx, const float* lse, float* delta, void* dqkv,
                            const int32_t* lens, int B, int S, int H, int dk, float scale, int dtype, hipStream_t stream) {
    FS2_CHECK_ARG(qkv && ctx && dctx && lse && delta && dqkv, "attn_bwd: null pointer");
    FS2_CHECK_ARG(dk == DK, "attn_bwd: only d_k = 128 is supported (got %d)", dk);
    if (B == 0) return FS2_OK;
    long ld = 3L * H * DK, ldo = (long)H * DK;
    int rows = B * S;
    dim3 grid(fs2_cdiv(S, 128), H, B);
    const int dyn = 163840;
    if (dtype == FS2_F32) {
        attn_delta_kernel<float><<<fs2_cdiv(rows * H, 4), 256, 0, stream>>>((const float*)ctx, (const float*)dctx, delta, rows, S, H);
        static Fs2DevOnce once;
        once.run([&] { (void)hipFuncSetAttribute((const void*)attn_bwd_dkv_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, dyn); });
        attn_bwd_dkv_kernel<float><<<grid, 256, dyn, stream>>>((const float*)qkv, ld, (const float*)dctx, ldo, lse, delta, (float*)dqkv, lens, S, H, scale);
        attn_bwd_dq_kernel<float><<<grid, 256, 0, stream>>>((const float*)qkv, ld, (const float*)dctx, ldo, lse, delta, (float*)dqkv, lens, S, H, scale);
    } else if (dtype == FS2_BF16) {
        // dQ first: it also produces delta (row sums of dO * O) for the dK/dV kernel
        attn_bwd_dq_bf16_kernel<<<grid, 256, 0, stream>>>((const bf16_t*)qkv, ld, (const bf16_t*)ctx, (const bf16_t*)dctx, ldo, lse, delta, (bf16_t*)dqkv, lens, S, H, scale);
        static const int dkv_form = fs2_dev_env("FS2_ATTN_DKV", 2);          // dev A/B: 1 = the phase-serial first form
        if (dkv_form == 1) attn_bwd_dkv_bf16_kernel<<<grid, 256, 0, stream>>>((const bf16_t*)qkv, ld, (const bf16_t*)dctx, ldo, lse, delta, (bf16_t*)dqkv, lens, S, H, scale);
        else attn_bwd_dkv2_bf16_kernel<<<grid, 256, 0, stream>>>((const bf16_t*)qkv, ld, (const bf16_t*)dctx, ldo, lse, delta, (bf16_t*)dqkv, lens, S, H, scale);
    } else { fs2_set_error("attn_bwd: dtype"); return FS2_EDTYPE; }
    FS2_CHECK_LAUNCH("attn_bwd");
    return FS2_OK;
}
