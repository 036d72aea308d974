// fs2_loss.hip — FastSpeech2Loss (reference model/loss.py:19-92) as two launches forward (+ one backward).
//   mel_loss      = mean |mel      - mel_t| over valid frames x n_mel        (L1, masked_select by ~mel_mask)
//   postnet_loss  = mean |postnet  - mel_t|
//   pitch / energy loss = mean (pred - target)^2 over valid phonemes (phoneme_level) or frames (frame_level)
//   duration_loss = mean (log_d_pred - log(d + 1))^2 over valid phonemes
//   total         = their sum
// "valid" is t < lens[b] (the reference's ~mask), the means divide by the number of valid positions, which the caller
// passes in DEVICE memory (cnt[0] = valid phonemes, cnt[1] = valid frames) so that data-parallel runs can substitute the
// all-reduced global counts / world (fastspeech2_amd/ddp.py) without a host round trip.
// The reference needs ~9 masked_select gathers + ~25 elementwise / reduction launches forward and as many backward; here
// the padded tensors are read once forward and once backward, padding contributes exact zeros, no mask is materialised.
#include "fs2_common.h"

struct LossArgs {
    const float* mel; const float* post; const float* mel_t; long ld_t_b;     // mel/post: [B][T][n_mel]; target batch stride (its own T_t)
    const int64_t* mel_lens; const int64_t* src_lens;
    const float* p_pred; const float* p_t; const float* e_pred; const float* e_t;   // [B][Sp] / [B][Se]; targets with own row stride
    long ld_pt, ld_et;
    const float* logd; const int64_t* dur;                                         // [B][L]
    long ld_dur;
    int B, T, L, n_mel, p_frame, e_frame;       // p_frame / e_frame: 1 = frame-level feature (mask = mel mask, length T)
    float* sums;                                // [5]: mel, post, pitch, energy, duration (atomically accumulated)
};

__device__ __forceinline__ float block_sum_256(float v, float* s) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = v;
    __syncthreads();
    return s[0] + s[1] + s[2] + s[3];
}

// grid (blocks_per_seq, B): the block strides over one sequence's valid (frame, channel) elements and valid phonemes.
__global__ void __launch_bounds__(256) loss_fwd_kernel(LossArgs a) {
    __shared__ float s[4];
    const int b = blockIdx.y;
    const int mlen = (int)min((int64_t)a.T, a.mel_lens[b]);
    const int slen = (int)min((int64_t)a.L, a.src_lens[b]);
    float am = 0.f, ap = 0.f;
    {   // mel L1 terms
        const size_t n = (size_t)mlen * a.n_mel;                       // valid elements are a contiguous prefix of the row block
        const float* m = a.mel + (size_t)b * a.T * a.n_mel;
        const float* p = a.post + (size_t)b * a.T * a.n_mel;
        const float* t = a.mel_t + (size_t)b * a.ld_t_b;
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
            float tv = t[i];
            am += fabsf(m[i] - tv);
            ap += fabsf(p[i] - tv);
        }
    }
    float apit = 0.f, aen = 0.f, adu = 0.f;
    if (blockIdx.x == 0) {
        const int pl = a.p_frame ? mlen : slen, el = a.e_frame ? mlen : slen;
        const int ps = a.p_frame ? a.T : a.L, es = a.e_frame ? a.T : a.L;
        for (int i = threadIdx.x; i < pl; i += 256) { float d = a.p_pred[(size_t)b * ps + i] - a.p_t[(size_t)b * a.ld_pt + i]; apit += d * d; }
        for (int i = threadIdx.x; i < el; i += 256) { float d = a.e_pred[(size_t)b * es + i] - a.e_t[(size_t)b * a.ld_et + i]; aen += d * d; }
        for (int i = threadIdx.x; i < slen; i += 256) {
            float d = a.logd[(size_t)b * a.L + i] - logf((float)a.dur[(size_t)b * a.ld_dur + i] + 1.f);
            adu += d * d;
        }
    }
    am = block_sum_256(am, s);
    ap = block_sum_256(ap, s);
    if (threadIdx.x == 0) { atomicAdd(a.sums + 0, am); atomicAdd(a.sums + 1, ap); }
    if (blockIdx.x == 0) {
        apit = block_sum_256(apit, s); aen = block_sum_256(aen, s); adu = block_sum_256(adu, s);
        if (threadIdx.x == 0) { atomicAdd(a.sums + 2, apit); atomicAdd(a.sums + 3, aen); atomicAdd(a.sums + 4, adu); }
    }
}

// losses[6] = {total, mel, postnet, pitch, energy, duration}; cnt = {valid phonemes, valid frames}
__global__ void loss_finalize_kernel(const float* __restrict__ sums, const float* __restrict__ cnt, int n_mel, int p_frame,
                                     int e_frame, float* __restrict__ losses) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const float ns = cnt[0], nm = cnt[1];
    const float mel = sums[0] / (nm * (float)n_mel), post = sums[1] / (nm * (float)n_mel);
    const float pit = sums[2] / (p_frame ? nm : ns), en = sums[3] / (e_frame ? nm : ns), du = sums[4] / ns;
    losses[1] = mel; losses[2] = post; losses[3] = pit; losses[4] = en; losses[5] = du;
    losses[0] = mel + post + du + pit + en;
}

extern "C" int fs2_loss_fwd(const float* mel, const float* post, const float* mel_t, long ld_t_b, const int64_t* mel_lens,
                            const int64_t* src_lens, const float* p_pred, const float* p_t, long ld_pt, const float* e_pred,
                            const float* e_t, long ld_et, const float* logd, const int64_t* dur, long ld_dur, const float* cnt,
                            int B, int T, int L, int n_mel, int p_frame, int e_frame, float* sums, float* losses,
                            hipStream_t stream) {
    FS2_CHECK_ARG(mel && post && mel_t && mel_lens && src_lens && p_pred && p_t && e_pred && e_t && logd && dur && cnt && sums && losses,
                  "loss_fwd: null pointer");
    FS2_CHECK_ARG(B > 0 && T > 0 && L > 0 && n_mel > 0, "loss_fwd: bad shape B=%d T=%d L=%d n_mel=%d", B, T, L, n_mel);
    LossArgs a;
    a.mel = mel; a.post = post; a.mel_t = mel_t; a.ld_t_b = ld_t_b; a.mel_lens = mel_lens; a.src_lens = src_lens;
    a.p_pred = p_pred; a.p_t = p_t; a.e_pred = e_pred; a.e_t = e_t; a.ld_pt = ld_pt; a.ld_et = ld_et; a.logd = logd; a.dur = dur;
    a.ld_dur = ld_dur; a.B = B; a.T = T; a.L = L; a.n_mel = n_mel; a.p_frame = p_frame; a.e_frame = e_frame; a.sums = sums;
    (void)hipMemsetAsync(sums, 0, 5 * sizeof(float), stream);
    int bps = fs2_cdiv((long)T * n_mel, 256 * 32);          // few blocks per sequence: each ends with 2-5 same-address atomics
    if (bps > 8) bps = 8;
    loss_fwd_kernel<<<dim3(bps, B), 256, 0, stream>>>(a);
    loss_finalize_kernel<<<1, 64, 0, stream>>>(sums, cnt, n_mel, p_frame, e_frame, losses);
    FS2_CHECK_LAUNCH("loss_fwd");
    return FS2_OK;
}

// Gradients of the 5 terms w.r.t. the predictions, scaled by the upstream gradients g[6] (device: d total, d mel, d postnet,
// d pitch, d energy, d duration; a tensor's factor is g[0] + g[its own term]).  Padded positions get exact zeros.
struct LossBwdArgs {
    LossArgs f;
    const float* cnt; const float* g;
    float* dmel; float* dpost; float* dp; float* de; float* dlogd;
};
__global__ void __launch_bounds__(256) loss_bwd_kernel(LossBwdArgs q) {
    const LossArgs& a = q.f;
    const int b = blockIdx.y;
    const int mlen = (int)min((int64_t)a.T, a.mel_lens[b]);
    const int slen = (int)min((int64_t)a.L, a.src_lens[b]);
    const float ns = q.cnt[0], nm = q.cnt[1];
    {
        const float km = (q.g[0] + q.g[1]) / (nm * (float)a.n_mel), kp = (q.g[0] + q.g[2]) / (nm * (float)a.n_mel);
        const size_t n = (size_t)mlen * a.n_mel, nall = (size_t)a.T * a.n_mel;
        const size_t o = (size_t)b * a.T * a.n_mel;
        const float* t = a.mel_t + (size_t)b * a.ld_t_b;
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nall; i += (size_t)gridDim.x * 256) {
            float gm = 0.f, gp = 0.f;
            if (i < n) {
                float tv = t[i];
                float dm = a.mel[o + i] - tv, dq = a.post[o + i] - tv;
                gm = dm > 0.f ? km : (dm < 0.f ? -km : 0.f);        // sign(0) = 0, as torch's l1_loss backward
                gp = dq > 0.f ? kp : (dq < 0.f ? -kp : 0.f);
            }
            q.dmel[o + i] = gm;
            q.dpost[o + i] = gp;
        }
    }
    if (blockIdx.x == 0) {
        const int pl = a.p_frame ? mlen : slen, el = a.e_frame ? mlen : slen;
        const int ps = a.p_frame ? a.T : a.L, es = a.e_frame ? a.T : a.L;
        const float kp = 2.f * (q.g[0] + q.g[3]) / (a.p_frame ? nm : ns), ke = 2.f * (q.g[0] + q.g[4]) / (a.e_frame ? nm : ns);
        const float kd = 2.f * (q.g[0] + q.g[5]) / ns;
        for (int i = threadIdx.x; i < ps; i += 256)
            q.dp[(size_t)b * ps + i] = i < pl ? kp * (a.p_pred[(size_t)b * ps + i] - a.p_t[(size_t)b * a.ld_pt + i]) : 0.f;
        for (int i = threadIdx.x; i < es; i += 256)
            q.de[(size_t)b * es + i] = i < el ? ke * (a.e_pred[(size_t)b * es + i] - a.e_t[(size_t)b * a.ld_et + i]) : 0.f;
        for (int i = threadIdx.x; i < a.L; i += 256)
            q.dlogd[(size_t)b * a.L + i] =
                i < slen ? kd * (a.logd[(size_t)b * a.L + i] - logf((float)a.dur[(size_t)b * a.ld_dur + i] + 1.f)) : 0.f;
    }
}
extern "C" int fs2_loss_bwd(const float* mel, const float* post, const float* mel_t, long ld_t_b, const int64_t* mel_lens,
                            const int64_t* src_lens, const float* p_pred, const float* p_t, long ld_pt, const float* e_pred,
                            const float* e_t, long ld_et, const float* logd, const int64_t* dur, long ld_dur, const float* cnt,
                            const float* g, int B, int T, int L, int n_mel, int p_frame, int e_frame, float* dmel, float* dpost,
                            float* dp, float* de, float* dlogd, hipStream_t stream) {
    FS2_CHECK_ARG(mel && post && mel_t && mel_lens && src_lens && p_pred && p_t && e_pred && e_t && logd && dur && cnt && g && dmel &&
                  dpost && dp && de && dlogd, "loss_bwd: null pointer");
    FS2_CHECK_ARG(B > 0 && T > 0 && L > 0 && n_mel > 0, "loss_bwd: bad shape");
    LossBwdArgs q;
    LossArgs& a = q.f;
    a.mel = mel; a.post = post; a.mel_t = mel_t; a.ld_t_b = ld_t_b; a.mel_lens = mel_lens; a.src_lens = src_lens;
    a.p_pred = p_pred; a.p_t = p_t; a.e_pred = e_pred; a.e_t = e_t; a.ld_pt = ld_pt; a.ld_et = ld_et; a.logd = logd; a.dur = dur;
    a.ld_dur = ld_dur; a.B = B; a.T = T; a.L = L; a.n_mel = n_mel; a.p_frame = p_frame; a.e_frame = e_frame; a.sums = nullptr;
    q.cnt = cnt; q.g = g; q.dmel = dmel; q.dpost = dpost; q.dp = dp; q.de = de; q.dlogd = dlogd;
    int bps = fs2_cdiv((long)T * n_mel, 256 * 8);
    if (bps > 64) bps = 64;
    loss_bwd_kernel<<<dim3(bps, B), 256, 0, stream>>>(q);
    FS2_CHECK_LAUNCH("loss_bwd");
    return FS2_OK;
}
