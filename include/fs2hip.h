/* fs2hip.h — C ABI of libfs2hip.so: hand-written gfx950 (CDNA4) kernels for the FastSpeech 2 hot path.
 *
 * The reference (ming024/FastSpeech2) has no FFI layer: its hot path is torch op call sites inside Python
 * nn.Modules.  Each entry point below replaces the torch ops at the cited reference location (paths relative
 * to the reference tree).  Host code above this ABI (fastspeech2_amd/*.py) mirrors the reference's Python
 * surface and binds these symbols with ctypes.
 *
 * Conventions
 *   - plain pointers + sizes only; every pointer is DEVICE memory owned by the caller (incl. workspaces);
 *   - functions enqueue work on `stream` and return immediately: no allocation, no synchronisation,
 *     no global mutable state (re-entrant; safe from autograd / DDP hook threads);
 *   - return 0 on success; <0 on error (FS2_EINVAL -1 bad shape/null, FS2_EDTYPE -2, FS2_ELAUNCH -3),
 *     message via fs2_last_error() (thread-local);
 *   - dtype: 0 = float32, 1 = bfloat16 (storage of activations / packed weights; accumulation, statistics,
 *     parameters, gradients of parameters and optimiser state are always float32);
 *   - activations are time-major rows [B*S][C] with C contiguous (== the reference's (B, S, C) tensors);
 *   - `lens` (int32 per sequence) replaces the reference's bool masks: row t of sequence b is padding iff
 *     t >= lens[b] (utils/tools.py:91-99 get_mask_from_lengths).
 */
#ifndef FS2HIP_H
#define FS2HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* fs2_stream_t; /* == hipStream_t */

enum { FS2_F32 = 0, FS2_BF16 = 1 };
enum { FS2_ACT_NONE = 0, FS2_ACT_RELU = 1, FS2_ACT_TANH = 2, FS2_ACT_LRELU = 3,
       FS2_ACT_GATE = 4 /* Y = (R > 0) ? acc+bias : 0 : ReLU backward fused into the data-gradient GEMM */ };

int fs2_version(void);
const char* fs2_last_error(void);

/* ---- dense contractions (MFMA) -------------------------------------------------------------------- */
/* Y[m][n] = ((act(sum_{j<taps} sum_c X[m + j*dil - pad][c] * W[n][j][c] + bias[n])) + R[m][n]) * out_scale
 * (+= Y if accumulate); taps shifted outside their sequence [0,S) contribute 0; rows t >= lens[b] -> 0.
 * Replaces nn.Linear / nn.Conv1d forward AND data-gradient (with the tap-flipped pack):
 *   transformer/SubLayers.py:39-41,54 (w_qs/w_ks/w_vs/fc), :87-88 (w_1 k=9, w_2 k=1);
 *   model/modules.py:209-240,291-296 (VariancePredictor Conv k=3); model/fastspeech2.py:95 (mel_linear);
 *   transformer/Layers.py:129-137 (PostNet k=5); hifigan/models.py:96-103,150-163; audio/stft.py:66-72. */
int fs2_conv_gemm(const void* X, long ldx, const void* Wpacked, const float* bias, const void* R, long ldr, void* Y,
                  long ldy, const int32_t* lens, const int32_t* tile_map, int M, int N, int Cin, int S, int taps, int dil,
                  int pad, int act, float slope, int in_act, float in_slope, int accumulate, float out_scale, int dtype,
                  fs2_stream_t stream);
/* tile_map (optional, from fs2_tile_map with rows = 256 for the same lens / S / M): lets the persistent kernel deal only
 * the M-tiles that hold valid rows to its workgroups.  With lens != NULL and tile_map == NULL the non-persistent kernels
 * run (same results).
 * fs2_tile_map: out[0] = number of REAL `rows`-row M-tiles of the [B*S] row space, out[1..] = their indices (ascending)
 * followed by the fully padded ones (all rows in one sequence's tail t >= lens[b]); out holds 1 + ceil(B*S/rows) ints. */
int fs2_tile_map(const int32_t* lens, int B, int S, int rows, int32_t* out, fs2_stream_t stream);
/* Everything a step derives from a lengths vector (int64, as the reference's batches carry them) in one launch:
 * lens32[b] = min(len[b], S); mask[b][t] = (t >= len[b]) as bytes (the bool masks FastSpeech2.forward returns, True = padding,
 * reference utils/tools.py:91-99); count[0] = sum_b lens32[b] (FastSpeech2Loss's valid-position count); tile_map as fs2_tile_map
 * (optional). */
int fs2_lens_prep(const int64_t* lens, int B, int S, int rows, int32_t* lens32, void* mask, float* count, int32_t* tile_map,
                  fs2_stream_t stream);
/* fs2_conv_gemm with scratch for the persistent kernel's TAIL SPLIT: when the output tiles do not fill a whole number of rounds
 * over the CUs, the last partial round is K-split (2 / 4 / 8 ways) so that it costs a fraction of a round instead of a whole
 * one; the parts store f32 partial tiles into tail_ws and one more launch sums them and applies the epilogue.  tail_ws:
 * fs2_conv_gemm_tail_ws_bytes() bytes, 16-byte aligned, any contents, not shared by launches that may run concurrently; NULL =
 * fs2_conv_gemm.  Launches the persistent kernel does not take ignore it. */
int fs2_conv_gemm_tail_ws_bytes(void);   /* a size, not a status */
int fs2_conv_gemm_tail(const void* X, long ldx, const void* Wpacked, const float* bias, const void* R, long ldr, void* Y,
                       long ldy, const int32_t* lens, const int32_t* tile_map, float* tail_ws, int M, int N, int Cin, int S,
                       int taps, int dil, int pad, int act, float slope, int in_act, float in_slope, int accumulate,
                       float out_scale, int dtype, fs2_stream_t stream);
/* K-split form for few-tile, long-reduction contractions: `ksplit` workgroups per output tile store partial tiles into ws
 * (f32 scratch, ksplit x M x N, any contents), one more launch sums them and finalises (bias, activation, residual, bf16).
 * bf16 only, Cin % (64 ksplit) == 0; FS2_EINVAL for unsupported shapes (fall back to fs2_conv_gemm). */
int fs2_conv_gemm_splitk(const void* X, long ldx, const void* Wpacked, const float* bias, const void* R, long ldr, void* Y,
                         long ldy, const int32_t* lens, const int32_t* tile_map, float* ws, int ksplit, int M, int N, int Cin,
                         int S, int taps, int dil, int pad, int act, float slope, float out_scale, int dtype,
                         fs2_stream_t stream);
/* fs2_conv_gemm_tail for HiFi-GAN's pre-activation chains when the activations are STORED leaky-ReLU'd (hifigan/models.py:96-103:
 * every convolution's input is leaky_relu(x), the residual is the raw x): the producer stores lrelu(value) (post_slope = slope,
 * applied last: after residual, out_scale and accumulate), the next convolution reads it with no prologue, and a residual operand
 * stored that way is undone on the fly (res_unlrelu = 1 / slope: r > 0 ? r : r * res_unlrelu).  0 switches either off. */
int fs2_conv_gemm_lrelu_io(const void* X, long ldx, const void* Wpacked, const float* bias, const void* R, long ldr, void* Y,
                           long ldy, float* tail_ws, int M, int N, int Cin, int S, int taps, int dil, int pad, int act, float slope,
                           int accumulate, float out_scale, float res_unlrelu, float post_slope, int dtype, fs2_stream_t stream);
/* Measurement aid: a back-to-back v_mfma_f32_32x32x16_bf16 stream on every SIMD of the chip (one 256-thread workgroup per CU,
 * iters x 8 MFMAs per wave); *flops receives the FLOPs of the launch.  bench.py times it to report what the matrix pipes SUSTAIN
 * under the chip's power management beside the nominal peak.  sink: one float of device memory (never written in practice). */
int fs2_mfma_calibrate(int iters, float* sink, double* flops, fs2_stream_t stream);
/* The memory-side calibrations reported beside it (roofline.hbm_copy / roofline.l2_to_lds): a 16-byte-per-lane copy of `bytes`
 * (src -> dst), and an LDS-DMA stream of an L2-resident window on every CU (8 waves x 4 KiB in flight; *bytes = bytes moved). */
int fs2_hbm_calibrate(const void* src, void* dst, size_t bytes, fs2_stream_t stream);
int fs2_ldsdma_calibrate(const void* src, size_t src_bytes, int iters, float* sink, double* bytes, fs2_stream_t stream);
/* Which kernel fs2_conv_gemm dispatches a launch description to: a pure function (no state) - a measurement aid that
 * lets bench.py attribute HIP-event durations to the kernel names rocprofv3 reports.  has_lens / has_map: whether
 * lens / tile_map would be non-NULL; ldr = 0 without a residual operand. */
#define FS2_GEMM_PLAIN 1   /* conv_gemm_kernel: 128x128 register-staged */
#define FS2_GEMM_DMA 2     /* conv_gemm_dma_kernel: 128x128 LDS-DMA, halo reuse (incl. in-workgroup split-K) */
#define FS2_GEMM_RING 3    /* conv_gemm_ring_kernel: 256x128 wave-specialised (shapes the persistent kernel declines; lrelu_io launches with a residual) */
#define FS2_GEMM_SKINNY 4  /* conv_skinny_kernel: C = 32 / 64 */
#define FS2_GEMM_PERSIST 5 /* conv_gemm_p_kernel<false>: persistent 256x128 convolution (taps >= 3), MFMA-bound */
#define FS2_GEMM_PERSIST_1TAP 6 /* conv_gemm_p_kernel<true>: the same kernel for taps == 1 (Linear / k=1 conv): HBM-bound at K <= 1024 */
#define FS2_GEMM_WIDE_1TAP 7 /* conv_gemm_w_kernel: persistent 256x256 tiles, every wave loads and multiplies (taps == 1, N % 256 == 0) */
/* (8 was round 4's 512x128 tall-tile kernel: measured slower at every shape of the model and removed from the library in round 5;
 *  git history: fastspeech2_amd/csrc/fs2_gemm_t.hip at ff5fda0) */
#define FS2_GEMM_STREAM_K256 9 /* conv_gemm_s_kernel: one tap, K = 256, N % 256 == 0 - weights in registers, X streamed through LDS (HBM-bound) */
int fs2_conv_gemm_variant(long ldx, long ldy, long ldr, int has_lens, int has_map, int M, int N, int Cin, int S, int taps,
                          int dil, int in_act, float in_slope, int dtype);
/* ... and of a fs2_conv_gemm_lrelu_io launch (its dispatch also looks at the epilogue operands: residual / accumulate launches with
 * a short reduction go to the ring kernel's LDS-staged epilogue) */
int fs2_conv_gemm_lrelu_io_variant(long ldx, long ldy, long ldr, int accumulate, int M, int N, int Cin, int S, int taps, int dil,
                                   int act, float res_unlrelu, float post_slope, int dtype);
/* Master conv weights are stored tap-major W[n][j][c] f32 (the (Cout,Cin,k) nn.Parameter is a permuted view of it):
 * -> Wf[n][j][c] (forward: dtype cast) and/or Wd[c][j][n] = W[n][k-1-j][c] (data gradient: tap flip + transpose). */
int fs2_pack_weight(const float* w, void* wf, void* wd, int Cout, int Cin, int k, int dtype, fs2_stream_t stream);
/* All data-gradient packs of a model in ONE launch.  table (device, int64 [n_entries][6]) = {src element offset into
 * flat, dst element offset into wd_base, Cout, Cin, k, first 64x64 tile index}; total_tiles = sum over entries of
 * ceil(Cout/64)*ceil(Cin/64)*k.  Each entry is packed as fs2_pack_weight's Wd (tap flip + transpose, LDS-tiled). */
int fs2_pack_dgrad_multi(const float* flat, void* wd_base, const int64_t* table, int n_entries, int total_tiles, int dtype,
                         fs2_stream_t stream);
/* dW[n][j][c] += sum_m dY[m][n] * X[m + j*dil - pad][c]   (tap-major master layout, f32, atomic accumulate);
 * dbias (optional): dbias[n] += sum_m dY[m][n], fused into the same pass over dY (the conv/linear bias gradient).
 * lens (optional): the caller guarantees dY rows t >= lens[b] are zero, so their K-tiles are skipped. */
int fs2_conv_wgrad(const void* dY, long lddy, const void* X, long ldx, float* dW, float* dbias, const int32_t* lens, int M,
                   int N, int Cin, int S, int taps, int dil, int pad, int dtype, fs2_stream_t stream);
/* The same with a split-K workspace (round 3): every split stores its partial tile into its own f32 copy of dW inside `ws`
 * (plain stores) and one more launch sums the splits into dW / dbias in index order - no atomics, bit-reproducible gradients,
 * and (dil == 1) tap groups of up to 5 taps that share one set of X fragment reads.  ws: fs2_conv_wgrad_ws_bytes(...) bytes for
 * the same shape, 16-byte aligned, any contents, not shared by launches that may run concurrently; NULL or too small = the
 * atomic path of fs2_conv_wgrad.  bf16 only (fp32 ignores ws).  fs2_conv_wgrad_ws_bytes returns a size, not a status. */
int fs2_conv_wgrad_ws_bytes(int M, int N, int Cin, int S, int taps, int dil, int has_lens, int dtype);
int fs2_conv_wgrad_ws_cap(void);     /* upper bound of fs2_conv_wgrad_ws_bytes over all shapes that get a workspace at all (a size, not a status) */
int fs2_conv_wgrad_ws(const void* dY, long lddy, const void* X, long ldx, float* dW, float* dbias, const int32_t* lens, int M,
                      int N, int Cin, int S, int taps, int dil, int pad, int dtype, float* ws, long ws_bytes, fs2_stream_t stream);
/* out[n] += sum_m x[m][n]  (bias gradients) */
int fs2_colsum(const void* x, long ldx, float* out, int M, int N, int dtype, fs2_stream_t stream);

/* ---- attention: transformer/Modules.py:14-25 + head split/merge SubLayers.py:39-52 ----------------- */
/* qkv [B*S][3*H*128] = (q|k|v); ctx [B*S][H*128]; lse [B][H][S] f32 (saved for backward). */
int fs2_attn_fwd(const void* qkv, void* ctx, float* lse, const int32_t* lens, int B, int S, int H, int dk, float scale,
                 int dtype, fs2_stream_t stream);
/* delta [B][H][S] f32 workspace; dqkv [B*S][3*H*128] receives (dq|dk|dv). */
int fs2_attn_bwd(const void* qkv, const void* ctx, const void* dctx, const float* lse, float* delta, void* dqkv,
                 const int32_t* lens, int B, int S, int H, int dk, float scale, int dtype, fs2_stream_t stream);

/* ---- LayerNorm family: SubLayers.py:54-55,90-91 + Layers.py:25,28; model/modules.py:222-240 -------- */
/* z = drop_pre(y) + res (written back into y); out = mask(drop_post(LN(z)*gamma+beta)); saves mean/rstd. */
int fs2_ln_fwd(void* y, const void* res, const float* gamma, const float* beta, const int32_t* lens, void* out,
               float* mean, float* rstd, int B, int S, int C, float eps, float p_pre, uint64_t seed_pre, float p_post,
               uint64_t seed_post, const uint64_t* seed_dev, int dtype, fs2_stream_t stream);
/* gemm_res_ln: the N = 256 projection of a transformer sub-layer AND its LayerNorm in one launch (SubLayers.py:54-55, 90-93 +
 * Layers.py:25,28): Z = dropout_pre(X W^T + bias) + R (bf16, saved for backward: what fs2_ln_fwd leaves in y),
 * out = mask(LN(Z) * gamma + beta), mean / rstd saved.  Same dropout stream and statistics as fs2_conv_gemm + fs2_ln_fwd; the
 * projection's output is never stored (one bf16 rounding less).  bf16, N == 256, Cin % 32 == 0; fs2_gemm_res_ln_supported
 * returns 1 / 0 (a flag, not a status) - otherwise call the two entry points. */
int fs2_gemm_res_ln_supported(int M, int N, int Cin, int S, int dtype);
/* 1 when the shape runs on the streaming K = 256 kernel (weights in registers, 64-row tiles: every CU busy at the decoder's M) */
int fs2_gemm_res_ln_streams(int M, int N, int Cin, int S, int dtype);
int fs2_gemm_res_ln_fwd(const void* X, long ldx, const void* Wpacked, const float* bias, const void* R, long ldr, void* Z, long ldz,
                        void* out, long ldo, const float* gamma, const float* beta, float* mean, float* rstd, const int32_t* lens,
                        const int32_t* tile_map, int M, int N, int Cin, int S, float eps, float p_pre, uint64_t seed_pre,
                        const uint64_t* seed_dev, int dtype, fs2_stream_t stream);
/* d1 = dz (+ d1_add), d2 = dz * dropmask_pre * (relu_bwd ? z>0 : 1); dgamma/dbeta += column sums.
 * partial_ws: caller workspace of per-block partial sums, reduced by a 2nd launch: FS2_LN_BWD_GRID*2*C floats when dgamma /
 * dbeta are given (nothing is written beyond that).  dgamma = dbeta = NULL DEFERS the reduction: the workspace must then hold
 * FS2_LN_BWD_GRID*2*C + 4 floats (the block count is stored behind the partials) and the caller runs
 * fs2_ln_bwd_reduce(partial_ws, ...) later, on any stream ordered after this call, before partial_ws is reused. */
#define FS2_LN_BWD_GRID 1024
int fs2_ln_bwd(const void* z, const void* dout, const float* gamma, const int32_t* lens, const float* mean,
               const float* rstd, const void* d1_add, void* d1, void* d2, float* dgamma, float* dbeta,
               float* partial_ws, int B, int S, int C, float p_pre, uint64_t seed_pre, float p_post, uint64_t seed_post, const uint64_t* seed_dev,
               int relu_bwd, int dtype, fs2_stream_t stream);
int fs2_ln_bwd_reduce(const float* partial_ws, int C, float* dgamma /*+=*/, float* dbeta /*+=*/, fs2_stream_t stream);
/* fs2_ln_bwd with the upstream gradient given as a SUM of two tensors, dout + dout2 (dout2 may be NULL = fs2_ln_bwd): the residual
 * branch's gradient reaches the LayerNorm below a sub-layer as "data gradient of the sub-layer + gradient that bypassed it"
 * (autograd of `output + residual`, transformer/SubLayers.py:55,91).  Rounds 1-5 added the bypass term in the epilogue of the
 * contraction that produced the first (16 dependent residual loads behind the tile's own stores: 25-45 us per launch); here it is one
 * more coalesced row read of an HBM-bound kernel. */
int fs2_ln_bwd_sum(const void* z, const void* dout, const void* dout2, const float* gamma, const int32_t* lens, const float* mean,
                   const float* rstd, const void* d1_add, void* d1, void* d2, float* dgamma, float* dbeta,
                   float* partial_ws, int B, int S, int C, float p_pre, uint64_t seed_pre, float p_post, uint64_t seed_post, const uint64_t* seed_dev,
                   int relu_bwd, int dtype, fs2_stream_t stream);

/* ---- BatchNorm1d (+tanh, +dropout) of PostNet: transformer/Layers.py:129-137 ------------------------ */
/* Column sums are bit-reproducible: every reducing launch stores per-workgroup partial sums in a workspace and a second tiny
 * launch adds them in index order (no float atomics, no device-scope fences).  A BN workspace holds fs2_bn_ws_floats(C)
 * floats (any contents); its first 2C floats receive the sums.  Every reducing entry point takes the workspace's size in floats
 * (`ws_floats`) right behind the pointer and refuses one smaller than fs2_bn_ws_floats(C) - rounds 1-3 needed 2C floats, so a caller
 * built against the old prototype now fails at the call instead of being written out of bounds. */
int fs2_bn_ws_floats(int C);
int fs2_bn_stats(const void* x, float* stats /*workspace; [0,2C) = sum | sum of squared deviations*/, long ws_floats, int M, int C,
                 int dtype, fs2_stream_t stream);
/* train-mode statistics without housekeeping launches; also updates the running statistics and increments
 * num_batches_tracked (int64, optional) - nn.BatchNorm1d's buffers. */
int fs2_bn_train_stats(const void* x, float* stats_ws, long ws_floats, float* running_mean, float* running_var,
                       int64_t* num_batches_tracked, float* mean_rstd, int M, int C, float eps, float momentum, int dtype,
                       fs2_stream_t stream);
int fs2_bn_finalize(const float* stats, float* running_mean, float* running_var, float* mean_rstd /*2C*/, int M, int C,
                    float eps, float momentum, fs2_stream_t stream);
/* out = drop(act(BN(x))) + res ; seed_dev (optional, device) is added to every dropout seed so that a replayed
 * hipGraph draws a fresh mask each step. */
int fs2_bn_apply(const void* x, const float* mean_rstd, const float* gamma, const float* beta, const void* res, void* out,
                 int M, int C, int act, float p, uint64_t seed, const uint64_t* seed_dev, int dtype, fs2_stream_t stream);
int fs2_bn_bwd(const void* x, const void* dout, const float* mean_rstd, const float* gamma, const float* beta,
               float* sums /*workspace; [0,2C) = dbeta|dgamma*/, long ws_floats, void* dx, int M, int C, int act, float p,
               uint64_t seed, const uint64_t* seed_dev, int dtype, fs2_stream_t stream);
/* the same with dgamma_acc / dbeta_acc (parameter-gradient buffers) += the reduced sums; successive calls on one stream may
 * share one workspace. */
int fs2_bn_bwd_acc(const void* x, const void* dout, const float* mean_rstd, const float* gamma, const float* beta, float* sums,
                   long ws_floats, void* dx, float* dgamma_acc, float* dbeta_acc, int M, int C, int act, float p, uint64_t seed,
                   const uint64_t* seed_dev, int dtype, fs2_stream_t stream);

/* ---- gathers / index kernels ----------------------------------------------------------------------- */
/* transformer/Models.py:89-91: out = src_word_emb[tok] + position_enc[t] */
int fs2_embed_pe_fwd(const int64_t* tokens, const float* emb, const float* pe, void* out, int B, int L, int C, int V,
                     int dtype, fs2_stream_t stream);
int fs2_embed_bwd(const int64_t* tokens, const void* dy, float* demb, int rows, int C, int V, int pad_idx, int dtype,
                  fs2_stream_t stream);
/* model/fastspeech2.py:68-71: x[b,t,:] += speaker_emb[speakers[b]] */
int fs2_add_rowvec(void* x, const float* table, const int64_t* idx, int B, int S, int C, int V, int dtype,
                   fs2_stream_t stream);
int fs2_rowvec_bwd(const void* dy, float* dtable, const int64_t* idx, int B, int S, int C, int V, int dtype,
                   fs2_stream_t stream);
/* model/modules.py:80-100,121,126: out = x + emb[bucketize(vals*scale, bins)] ; idx_out saved for backward */
int fs2_bucket_embed_add_fwd(const void* x, const float* vals, float scale, const float* bins, int nbins,
                             const float* emb, void* out, int32_t* idx_out, int rows, int C, int dtype,
                             fs2_stream_t stream);
/* demb[n_bins][C] (fp32) += rows of dy grouped by bucket index (gather-reduce per bin, one atomic per channel and row split) */
int fs2_bucket_embed_bwd(const int32_t* idx, const void* dy, float* demb, int rows, int n_bins, int C, int dtype,
                         fs2_stream_t stream);
/* model/modules.py:167-194 LengthRegulator (+ utils/tools.py:299-317 pad): integer index map, bit-exact.
 * cum [B][L+1] int32 exclusive prefix sums of max((int)d,0); idx [B][T] frame->phoneme (-1 = padding);
 * mel_len [B] int64 = un-cropped total (Appendix A #4). */
int fs2_lr_index(const void* durations, int dur_is_float, int B, int L, int T, int32_t* cum, int32_t* idx,
                 int64_t* mel_len, fs2_stream_t stream);
/* out[b,t,:] = x[b,idx[b,t],:] (0 if idx<0) + (pe ? pe[t,:] : 0)  — pe fuses Decoder's position_enc add
 * (transformer/Models.py:154-162). */
int fs2_lr_gather_fwd(const void* x, const int32_t* idx, const float* pe, void* out, int B, int L, int T, int C,
                      int dtype, fs2_stream_t stream);
int fs2_lr_gather_bwd(const void* dy, const int32_t* cum, void* dx, int B, int L, int T, int C, int accumulate,
                      int dtype, fs2_stream_t stream);
/* model/modules.py:132-135: clamp(round_half_even(exp(log_d)-1) * d_control, min=0) */
int fs2_duration_round(const float* logd, float d_control, float* out, int n, fs2_stream_t stream);
/* model/modules.py:243-249: out[r] = mask(dot(x[r,:], w) + b) */
int fs2_rowdot_fwd(const void* x, const float* w, const float* bias, const int32_t* lens, float* out, int B, int S, int C,
                   int dtype, fs2_stream_t stream);
int fs2_rowdot_bwd(const void* x, const float* w, const float* g, const int32_t* lens, void* dx, float* dw, float* db,
                   int B, int S, int C, int dtype, fs2_stream_t stream);
/* transformer/Layers.py:25,28 masked_fill(mask, 0) in place */
int fs2_mask_rows(void* x, const int32_t* lens, int B, int S, int C, int dtype, fs2_stream_t stream);
int fs2_cast(const void* in, int in_dtype, void* out, int out_dtype, size_t n, fs2_stream_t stream);
int fs2_add(const void* a, const void* b, void* out, size_t n, int dtype, fs2_stream_t stream);
/* transformer/Models.py:154-162 (frame-level variance configs): x[b,t,:] += position_enc[t,:] */
int fs2_add_pe(void* x, const float* pe, int B, int S, int C, int dtype, fs2_stream_t stream);
int fs2_bump_counter(uint64_t* ctr, uint64_t inc, fs2_stream_t stream);

/* ---- vocoder (hifigan/models.py:113-174, utils/model.py:74-92) and mel extraction (audio/stft.py) ---- */
/* The dense layers run in fs2_conv_gemm:
 *   conv_pre / ResBlock convs   taps=k, dil, in_act = leaky_relu(0.1), residual / accumulate / out_scale=1/3 epilogue
 *                               (models.py:96-103,150,155-160);
 *   ConvTranspose1d(k=2u, stride u, pad u/2) as its polyphase form: a 3-tap conv with N = u*Cout whose output rows
 *                               [B*T][u*Cout] ARE the up-sampled time-major rows [B*T*u][Cout] (models.py:124-135,152-153);
 *   framed DFT                  X = reflect-padded wav viewed as rows [B*rows][hop], taps = filter/hop (stft.py:66-72).
 * (B, C, T) float32 -> rows [B*T][C] in `dtype` (vocoder_infer receives mels channel-major, utils/model.py:74-80). */
int fs2_chan_to_rows(const float* in, void* out, int B, int C, int T, int dtype, fs2_stream_t stream);
/* models.py:161-163 + utils/model.py:82-85: y = tanh(conv(leaky_relu(x, in_slope), w[taps][C]) + bias);
 * wav[m] = y (optional), pcm[m] = (int16)(y * max_wav_value) with numpy's astype semantics (optional). */
int fs2_conv_post_pcm(const void* x, long ldx, const float* w, const float* bias, float in_slope, float* wav,
                      int16_t* pcm, float max_wav_value, int M, int S, int C, int taps, int pad, int dtype,
                      fs2_stream_t stream);
/* hifigan/models.py:96-103 + 155-160: a whole ResBlock1 of the narrow stages (C = 32 / 64, bf16) in ONE launch -
 *   y = x; for m in 0..2: t = lrelu(conv1_m(lrelu(y), dilation d_m)); y = conv2_m(t) + y;  xs = (accumulate ? xs : 0) + out_scale * y
 * x / xs: rows [B*S][C] (time-major); w1 / w2: [3][C][k][C] (convs1 / convs2 of the block, cout-major, tap, cin), b1 / b2: [3][C] f32.
 * The running sum stays in fp32 registers, the convolutions' operands in LDS: x is read once, xs read + written once.
 * post_slope > 0: the xs written is leaky_relu(xs, post_slope) (the next up-sampling convolution then needs no prologue).
 * fs2_resblock_supported: 1 when a (C, k, dilations, dtype) combination has an instantiation (else run the convolutions one by one). */
int fs2_resblock_supported(int C, int k, int d0, int d1, int d2, int dtype);
int fs2_resblock_fwd(const void* x, long ldx, const void* w1, const void* w2, const float* b1, const float* b2, void* xs,
                     long ldxs, int accumulate, float out_scale, float slope, float post_slope, int B, int S, int C, int k, int d0,
                     int d1, int d2, int dtype, fs2_stream_t stream);
/* models.py:155-160 for a whole up-sampling stage: the three residual blocks (kernel sizes ka / kb / kc, same dilations) of one x
 * in ONE launch - xs = out_scale * (block_a(x) + block_b(x) + block_c(x)), the running xs rounded to the storage dtype after
 * every block (as the per-block launches stored it): x is read once from HBM (twice more from L2), xs written once. */
int fs2_resstage_fwd(const void* x, long ldx, const void* w1a, const void* w2a, const float* b1a, const float* b2a, int ka,
                     const void* w1b, const void* w2b, const float* b1b, const float* b2b, int kb, const void* w1c,
                     const void* w2c, const float* b1c, const float* b2c, int kc, void* xs, long ldxs, float out_scale,
                     float slope, float post_slope, int B, int S, int C, int d0, int d1, int d2, int dtype, fs2_stream_t stream);
/* stft.py:60-66: xp[b][i] = y[b][reflect(i - P)], i < N + 2P; zero-filled up to row_len. */
int fs2_reflect_pad(const float* y, float* xp, int B, int N, int P, long row_len, fs2_stream_t stream);
/* The same for a ragged batch (preprocessor/preprocessor.py:194 over a corpus): row b holds lens[b] samples (row stride ldy,
 * lens[b] > P else the row is zero-filled) and is reflected at its own end. */
int fs2_reflect_pad_ragged(const float* y, long ldy, const int32_t* lens, float* xp, int B, int P, long row_len,
                           fs2_stream_t stream);
/* stft.py:74-78,174-176 + audio_processing.py:91: ft rows [B*S][2*NF] (Re|Im) -> mel (B, n_mel, frames) =
 * log(clamp(mel_basis[n_mel][NF] . |ft|, clamp_min)), energy (B, frames) = ||ft|||_2; span[n_mel][2] = non-zero
 * band [lo, hi) of each filter. */
int fs2_stft_mel_epilogue(const float* ft, long ldft, const float* mel_basis, const int32_t* span, float* mel,
                          float* energy, int B, int S, int frames, int NF, int n_mel, float clamp_min,
                          fs2_stream_t stream);

/* ---- loss (model/loss.py:19-92): masked L1 (mel, post-net mel) + masked MSE (pitch, energy, log-duration) ----
 * mel / post: [B][T][n_mel] f32 predictions; mel_t: target with batch stride ld_t_b (its own padded length >= T);
 * lens int64 (valid = t < min(len, T)); p/e predictions [B][L] (phoneme level) or [B][T] (p_frame / e_frame = 1) with
 * target row strides ld_pt / ld_et; logd [B][L], dur int64 [B][L] (row stride ld_dur; target = log(dur + 1));
 * cnt (device) = {valid phonemes, valid frames} = the divisors of the means (data-parallel: global counts / world).
 * fwd: sums[5] workspace, losses[6] = {total, mel, postnet, pitch, energy, duration}.
 * bwd: g[6] (device) upstream gradients of the 6 outputs; writes dense gradients (exact zeros on padding). */
int fs2_loss_fwd(const float* mel, const float* post, const float* mel_t, long ld_t_b, const int64_t* mel_lens,
                 const int64_t* src_lens, const float* p_pred, const float* p_t, long ld_pt, const float* e_pred,
                 const float* e_t, long ld_et, const float* logd, const int64_t* dur, long ld_dur, const float* cnt,
                 int B, int T, int L, int n_mel, int p_frame, int e_frame, float* sums, float* losses, fs2_stream_t stream);
int fs2_loss_bwd(const float* mel, const float* post, const float* mel_t, long ld_t_b, const int64_t* mel_lens,
                 const int64_t* src_lens, const float* p_pred, const float* p_t, long ld_pt, const float* e_pred,
                 const float* e_t, long ld_et, const float* logd, const int64_t* dur, long ld_dur, const float* cnt,
                 const float* g, int B, int T, int L, int n_mel, int p_frame, int e_frame, float* dmel, float* dpost,
                 float* dp, float* de, float* dlogd, fs2_stream_t stream);

/* ---- optimiser: train.py:93 clip_grad_norm_ + model/optimizer.py:10-51 Adam ------------------------- */
/* ws: FS2_SUMSQ_BLOCKS floats of workspace (two-stage reduction in a fixed order: bit-reproducible norm) */
#define FS2_SUMSQ_BLOCKS 1024
int fs2_sumsq(const float* x, size_t n, float* out /*+=*/, float* ws, fs2_stream_t stream);
/* hyper (device) = {lr, 1-beta1^t, 1-beta2^t}; clip = min(1, max_norm/(sqrt(*gnorm_sq)+1e-6)).
 * p_lowp (optional, bf16): compute-dtype copy of the updated parameters written in the same pass;
 * zero_grad: clear g as it is consumed (optimizer.zero_grad(), model/optimizer.py:30-31).  n % 4 == 0. */
int fs2_adam_step(float* p, float* g, float* m, float* v, size_t n, const float* gnorm_sq, float max_norm,
                  const float* hyper, float b1, float b2, float eps, float wd, void* p_lowp, int lowp_dtype, int zero_grad,
                  fs2_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
