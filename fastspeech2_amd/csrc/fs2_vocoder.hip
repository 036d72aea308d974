// fs2_vocoder.hip — HBM-bound kernels around the MFMA contractions of the vocoder and of mel extraction.
//   (B,C,T) -> time-major rows            (reference utils/model.py:74-80: mels arrive as (B, 80, T))
//   conv_post + tanh + PCM16              (reference hifigan/models.py:161-163, utils/model.py:82-85)
//   reflect pad + framing rows            (reference audio/stft.py:60-66)
//   |DFT| -> mel filterbank -> log, energy (reference audio/stft.py:74-78,159-178, audio/audio_processing.py:85-91)
// The dense work (conv_pre, up-samplers as polyphase GEMMs, dilated ResBlock convs, the framed DFT) runs in
// fs2_conv_gemm; everything here is one pass over its operands.
#include "fs2_common.h"

#define DISPATCH_DTYPE(dtype, ...)                                   \
    if ((dtype) == FS2_F32) { typedef float T; __VA_ARGS__; }        \
    else if ((dtype) == FS2_BF16) { typedef bf16_t T; __VA_ARGS__; } \
    else { fs2_set_error("unsupported dtype %d", (int)(dtype)); return FS2_EDTYPE; }

// ------------------------------------------------------------------ (B, C, T) f32 -> rows [B*T][C]
// 32x32 LDS tile transpose: coalesced along T on the read, along C on the write.
template <typename T>
__global__ void chan_to_rows_kernel(const float* __restrict__ in, T* __restrict__ out, int C, int Tn) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z, t0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;           // 256 threads: ty in [0,8)
    for (int i = ty; i < 32; i += 8) {
        int c = c0 + i, t = t0 + tx;
        tile[i][tx] = (c < C && t < Tn) ? in[((size_t)b * C + c) * Tn + t] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        int t = t0 + i, c = c0 + tx;
        if (t < Tn && c < C) Elem<T>::st(out + ((size_t)b * Tn + t) * C + c, tile[tx][i]);
    }
}
extern "C" int fs2_chan_to_rows(const float* in, void* out, int B, int C, int T_, int dtype, hipStream_t stream) {
    FS2_CHECK_ARG(in && out, "chan_to_rows: null pointer");
    FS2_CHECK_ARG(B >= 0 && C > 0 && T_ >= 0, "chan_to_rows: bad shape");
    if (B == 0 || T_ == 0) return FS2_OK;
    dim3 grid(fs2_cdiv(T_, 32), fs2_cdiv(C, 32), B);
    DISPATCH_DTYPE(dtype, chan_to_rows_kernel<T><<<grid, 256, 0, stream>>>(in, (T*)out, C, T_));
    FS2_CHECK_LAUNCH("chan_to_rows");
    return FS2_OK;
}

// ------------------------------------------------------------------ conv_post: leaky_relu -> Conv1d(C,1,k) -> tanh -> PCM
// One output sample per thread: k rows x C channels (<= 7 x 32) read as 16-byte chunks; neighbouring threads share
// k-1 of their k rows, so HBM sees each row once (L1/L2 absorb the tap reuse).  wav f32 and/or int16 PCM out.
// PCM cast == numpy astype('int16') of float32 on x86: truncate toward zero to int32, keep the low 16 bits.
template <typename T>
__global__ void conv_post_kernel(const T* __restrict__ x, long ldx, const float* __restrict__ w, const float* __restrict__ bias,
                                 float in_slope, float* __restrict__ wav, int16_t* __restrict__ pcm, float max_wav, int M,
                                 int S, int C, int taps, int pad) {
    extern __shared__ float sw[];                       // [taps][C]
    for (int i = threadIdx.x; i < taps * C; i += blockDim.x) sw[i] = w[i];
    __syncthreads();
    int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    int t = m % S;
    float acc = bias ? bias[0] : 0.f;
    for (int j = 0; j < taps; ++j) {
        int ts = t + j - pad;
        if (ts < 0 || ts >= S) continue;
        const T* row = x + (size_t)(m + j - pad) * ldx;
        const float* wr = sw + j * C;
        for (int c = 0; c < C; c += 4) {
            float4 v = ld4<T>(row + c);
            v.x = v.x > 0.f ? v.x : v.x * in_slope;
            v.y = v.y > 0.f ? v.y : v.y * in_slope;
            v.z = v.z > 0.f ? v.z : v.z * in_slope;
            v.w = v.w > 0.f ? v.w : v.w * in_slope;
            acc = fmaf(v.x, wr[c], acc);
            acc = fmaf(v.y, wr[c + 1], acc);
            acc = fmaf(v.z, wr[c + 2], acc);
            acc = fmaf(v.w, wr[c + 3], acc);
        }
    }
    float y = tanhf(acc);
    if (wav) wav[m] = y;
    if (pcm) {
        float s = y * max_wav;
        int32_t i = (int32_t)s;                         // truncation toward zero
        pcm[m] = (int16_t)(i & 0xffff);                 // wrap like numpy (1.0 * 32768 -> -32768)
    }
}
extern "C" int fs2_conv_post_pcm(const void* x, long ldx, const float* w, const float* bias, float in_slope, float* wav,
                                 int16_t* pcm, float max_wav_value, int M, int S, int C, int taps, int pad, int dtype,
                                 hipStream_t stream) {
    FS2_CHECK_ARG(x && w && (wav || pcm), "conv_post_pcm: null pointer");
    FS2_CHECK_ARG(M >= 0 && S > 0 && C > 0 && C % 4 == 0 && taps > 0 && taps * C <= 8192 && ldx % 4 == 0,
                  "conv_post_pcm: bad shape M=%d S=%d C=%d taps=%d", M, S, C, taps);
    if (M == 0) return FS2_OK;
    DISPATCH_DTYPE(dtype, conv_post_kernel<T><<<fs2_cdiv(M, 256), 256, taps * C * sizeof(float), stream>>>(
                              (const T*)x, ldx, w, bias, in_slope, wav, pcm, max_wav_value, M, S, C, taps, pad));
    FS2_CHECK_LAUNCH("conv_post_pcm");
    return FS2_OK;
}

// ------------------------------------------------------------------ STFT framing: reflect pad into hop-wide rows
// xp[b][i] = y[b][reflect(i - P)] for i < min(row_len, N + 2P), 0 beyond; xp is [B][rows*hop] so that frame t of
// utterance b is rows t .. t + filter/hop - 1 of the [B*rows][hop] matrix (a `taps`-tap implicit GEMM).
__global__ void reflect_pad_kernel(const float* __restrict__ y, float* __restrict__ xp, int N, int P, long row_len) {
    int b = blockIdx.y;
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= row_len) return;
    float v = 0.f;
    if (i < (long)N + 2 * P) {
        long s = i - P;
        if (s < 0) s = -s;                              // F.pad(mode='reflect'): edge sample not repeated
        if (s >= N) s = 2L * (N - 1) - s;
        v = y[(size_t)b * N + s];
    }
    xp[(size_t)b * row_len + i] = v;
}
extern "C" int fs2_reflect_pad(const float* y, float* xp, int B, int N, int P, long row_len, hipStream_t stream) {
    FS2_CHECK_ARG(y && xp, "reflect_pad: null pointer");
    FS2_CHECK_ARG(B >= 0 && N > P && P >= 0 && row_len > 0, "reflect_pad: bad shape N=%d P=%d row_len=%ld", N, P, row_len);
    if (B == 0) return FS2_OK;
    dim3 grid(fs2_cdiv(row_len, 256), B);
    reflect_pad_kernel<<<grid, 256, 0, stream>>>(y, xp, N, P, row_len);
    FS2_CHECK_LAUNCH("reflect_pad");
    return FS2_OK;
}

// Ragged batch (corpus preprocessing): utterance b has its own length lens[b] (row stride ldy in y); each row is reflected at
// ITS end and zero-filled beyond, so frame t < lens[b]/hop + 1 of row b equals the frame of the utterance processed alone.
__global__ void reflect_pad_ragged_kernel(const float* __restrict__ y, long ldy, const int32_t* __restrict__ lens,
                                          float* __restrict__ xp, int P, long row_len) {
    int b = blockIdx.y;
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= row_len) return;
    const long N = lens[b];
    float v = 0.f;
    if (N > P && i < N + 2 * P) {
        long s = i - P;
        if (s < 0) s = -s;
        if (s >= N) s = 2L * (N - 1) - s;
        v = y[(size_t)b * ldy + s];
    }
    xp[(size_t)b * row_len + i] = v;
}
extern "C" int fs2_reflect_pad_ragged(const float* y, long ldy, const int32_t* lens, float* xp, int B, int P, long row_len,
                                      hipStream_t stream) {
    FS2_CHECK_ARG(y && xp && lens, "reflect_pad_ragged: null pointer");
    FS2_CHECK_ARG(B >= 0 && P >= 0 && row_len > 0 && ldy > 0, "reflect_pad_ragged: bad shape P=%d row_len=%ld ldy=%ld", P, row_len, ldy);
    if (B == 0) return FS2_OK;
    dim3 grid(fs2_cdiv(row_len, 256), B);
    reflect_pad_ragged_kernel<<<grid, 256, 0, stream>>>(y, ldy, lens, xp, P, row_len);
    FS2_CHECK_LAUNCH("reflect_pad_ragged");
    return FS2_OK;
}

// ------------------------------------------------------------------ |DFT| -> mel -> log ; energy
// ft rows [B*S][2*NF] = (Re[0..NF) | Im[0..NF)) from the framed-DFT GEMM; only rows t < frames of each utterance
// are frames.  A 256-thread block owns FR = 16 frames: magnitudes go to LDS once ([FR][NF] f32), energy is a wave
// reduction per frame, then thread (mel bin k, frame f) walks the non-zero span [lo_k, hi_k) of its triangular
// filter (Slaney filters are contiguous bands).  Outputs are channel-major like the reference: mel (B, n_mel, frames).
#define STFT_FR 16
__global__ void stft_mel_kernel(const float* __restrict__ ft, long ldft, const float* __restrict__ melb,
                                const int32_t* __restrict__ span, float* __restrict__ mel, float* __restrict__ energy,
                                int S, int frames, int NF, int n_mel, float clamp_min) {
    extern __shared__ float mag[];                      // [STFT_FR][NF]
    const int b = blockIdx.y, f0 = blockIdx.x * STFT_FR;
    const int nf = min(STFT_FR, frames - f0);
    for (int i = threadIdx.x; i < nf * NF; i += blockDim.x) {
        int f = i / NF, k = i - f * NF;
        const float* row = ft + ((size_t)b * S + f0 + f) * ldft;
        float re = row[k], im = row[NF + k];
        mag[f * NF + k] = sqrtf(re * re + im * im);
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int f = wave; f < nf; f += 4) {                // torch.norm(mag, dim=1)
        float s = 0.f;
        for (int k = lane; k < NF; k += 64) { float v = mag[f * NF + k]; s = fmaf(v, v, s); }
        s = wave_sum(s);
        if (lane == 0) energy[(size_t)b * frames + f0 + f] = sqrtf(s);
    }
    for (int i = threadIdx.x; i < n_mel * STFT_FR; i += blockDim.x) {
        int k = i / STFT_FR, f = i - k * STFT_FR;
        if (f >= nf) continue;
        int lo = span[2 * k], hi = span[2 * k + 1];
        const float* wrow = melb + (size_t)k * NF;
        const float* mrow = mag + f * NF;
        float acc = 0.f;
        for (int q = lo; q < hi; ++q) acc = fmaf(wrow[q], mrow[q], acc);
        mel[((size_t)b * n_mel + k) * frames + f0 + f] = logf(fmaxf(acc, clamp_min));
    }
}
extern "C" int fs2_stft_mel_epilogue(const float* ft, long ldft, const float* mel_basis, const int32_t* span, float* mel,
                                     float* energy, int B, int S, int frames, int NF, int n_mel, float clamp_min,
                                     hipStream_t stream) {
    FS2_CHECK_ARG(ft && mel_basis && span && mel && energy, "stft_mel_epilogue: null pointer");
    FS2_CHECK_ARG(B >= 0 && frames > 0 && frames <= S && NF > 0 && n_mel > 0 && ldft >= 2L * NF && NF <= 2048,
                  "stft_mel_epilogue: bad shape S=%d frames=%d NF=%d", S, frames, NF);
    if (B == 0) return FS2_OK;
    size_t lds = (size_t)STFT_FR * NF * sizeof(float);
    static Fs2DevOnce once;
    once.run([&] { (void)hipFuncSetAttribute((const void*)stft_mel_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 16 * 2048 * 4); });
    dim3 grid(fs2_cdiv(frames, STFT_FR), B);
    stft_mel_kernel<<<grid, 256, lds, stream>>>(ft, ldft, mel_basis, span, mel, energy, S, frames, NF, n_mel, clamp_min);
    FS2_CHECK_LAUNCH("stft_mel_epilogue");
    return FS2_OK;
}
