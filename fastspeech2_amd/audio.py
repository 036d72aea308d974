"""Mel extraction (reference audio/stft.py:130-178 TacotronSTFT, audio/tools.py:8-15 get_mel_from_wav) on the GPU.

`TacotronSTFT(filter_length, hop_length, win_length, n_mel_channels, sampling_rate, mel_fmin, mel_fmax)
 .mel_spectrogram(y (B, N) in [-1, 1]) -> (mel (B, n_mel, frames), energy (B, frames))`, frames = 1 + N // hop.

The reference computes the STFT as a strided conv1d with a windowed DFT basis (stft.py:66-72).  Here the
reflect-padded signal is viewed as rows of `hop` samples, which turns the framed DFT into a (filter/hop)-tap
implicit GEMM  ft[B*rows][2*cutoff] = sum_j X[m + j][:] . basis[:, j*hop:(j+1)*hop]^T  on the exact-fp32 MFMA
(`v_mfma_f32_32x32x2_f32`, same products and fp32 accumulation as the reference's dot products), followed by one
fused pass: magnitude -> mel filterbank (non-zero band of each filter only) -> log(clamp) and the per-frame
energy norm.

The mel filterbank is `librosa.filters.mel` (Slaney scale + area normalisation, librosa==0.7.2, the reference's
requirements.txt:3).  librosa is not available here; the formula is restated in `slaney_mel_filterbank` from its
published definition ("parity unpinned" at this one boundary — DESIGN.md §6).  A caller holding librosa can pass
`mel_basis=` explicitly.
"""
import numpy as np
import torch

from . import _lib, ops


def slaney_mel_filterbank(sr, n_fft, n_mels, fmin, fmax):
    """Slaney-style mel filterbank (htk=False, norm=1): linear below 1 kHz, log above, triangles normalised to unit area."""
    if fmax is None:
        fmax = sr / 2.0
    f_sp, min_log_hz = 200.0 / 3, 1000.0
    min_log_mel, logstep = min_log_hz / f_sp, np.log(6.4) / 27.0

    def to_mel(f):
        f = np.asarray(f, dtype=np.float64)
        return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-10) / min_log_hz) / logstep, f / f_sp)

    def to_hz(m):
        m = np.asarray(m, dtype=np.float64)
        return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)

    n_freq = 1 + n_fft // 2
    fft_f = np.linspace(0, sr / 2.0, n_freq)
    mel_f = to_hz(np.linspace(to_mel(fmin), to_mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fft_f[None, :]
    lower = -ramps[:-2] / fdiff[:-1, None]
    upper = ramps[2:] / fdiff[1:, None]
    # rounding order of the float32 table librosa returns: triangles -> float32, x float64 area norm -> float32
    w = np.maximum(0, np.minimum(lower, upper)).astype(np.float32)
    w = (w.astype(np.float64) * (2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels]))[:, None]).astype(np.float32)
    return w


def dft_basis(filter_length, win_length):
    """audio/stft.py:26-50: rows [Re(0..cutoff) ; Im(0..cutoff)] of the DFT matrix times a periodic hann window
    (scipy.signal.get_window('hann', win_length, fftbins=True), centre-padded to filter_length), float32."""
    from scipy.signal import get_window

    fb = np.fft.fft(np.eye(filter_length))
    cutoff = filter_length // 2 + 1
    fb = np.vstack([np.real(fb[:cutoff]), np.imag(fb[:cutoff])])
    win = get_window("hann", win_length, fftbins=True)
    if win_length < filter_length:
        lpad = (filter_length - win_length) // 2
        win = np.pad(win, (lpad, filter_length - win_length - lpad))
    return torch.FloatTensor(fb) * torch.from_numpy(win).float()          # (2*cutoff, filter_length)


class TacotronSTFT(torch.nn.Module):
    def __init__(self, filter_length, hop_length, win_length, n_mel_channels, sampling_rate, mel_fmin, mel_fmax,
                 mel_basis=None):
        super().__init__()
        assert filter_length % hop_length == 0, "framed-DFT GEMM needs hop | filter_length (every reference config)"
        assert filter_length >= win_length
        self.filter_length, self.hop_length, self.win_length = filter_length, hop_length, win_length
        self.n_mel_channels, self.sampling_rate = n_mel_channels, sampling_rate
        self.cutoff = filter_length // 2 + 1
        basis = dft_basis(filter_length, win_length)
        taps = filter_length // hop_length
        # packed for the implicit GEMM: W[n][tap][c] = basis[n][tap*hop + c]
        self.register_buffer("forward_basis", basis.view(2 * self.cutoff, taps, hop_length).contiguous())
        if mel_basis is None:
            mel_basis = slaney_mel_filterbank(sampling_rate, filter_length, n_mel_channels, mel_fmin, mel_fmax)
        mel_basis = torch.as_tensor(np.asarray(mel_basis), dtype=torch.float32)
        self.register_buffer("mel_basis", mel_basis.contiguous())
        nz = mel_basis != 0
        span = torch.zeros(n_mel_channels, 2, dtype=torch.int32)
        for k in range(n_mel_channels):
            idx = torch.nonzero(nz[k]).flatten()
            if idx.numel():
                span[k, 0], span[k, 1] = int(idx[0]), int(idx[-1]) + 1
        self.register_buffer("mel_span", span)

    def mel_spectrogram(self, y):
        """audio/stft.py:159-178."""
        assert torch.min(y.data) >= -1 and torch.max(y.data) <= 1          # stft.py:170-171
        if not y.is_cuda:
            raise RuntimeError("fastspeech2_amd.audio.TacotronSTFT runs on an AMD GPU only (no CPU fallback)")
        if self.forward_basis.device != y.device:
            self.to(y.device)
        y = y.contiguous().float()
        B, N = y.shape
        return self._framed_mel(y, B, N, None)

    def mel_spectrogram_ragged(self, y, lens):
        """Corpus form of `mel_spectrogram` (preprocessor/preprocessor.py:194 calls it once per utterance): y (B, Nmax) holds
        B utterances of `lens[b]` samples each (anything beyond is ignored).  Every row is reflected at its OWN end, so
        frames [0, lens[b] // hop + 1) of row b are bit-identical to the utterance processed alone; later frames are padding.
        Returns (mel (B, n_mel, Nmax // hop + 1), energy (B, Nmax // hop + 1), frames (B,) int64)."""
        if not y.is_cuda:
            raise RuntimeError("fastspeech2_amd.audio.TacotronSTFT runs on an AMD GPU only (no CPU fallback)")
        if self.forward_basis.device != y.device:
            self.to(y.device)
        y = y.contiguous().float()
        B, N = y.shape
        lens = torch.as_tensor(lens, device=y.device).to(torch.int32).contiguous()
        assert lens.numel() == B and int(lens.max()) <= N and int(lens.min()) > self.filter_length // 2, \
            "each utterance needs more than filter_length/2 samples (reflect padding) and must fit its row"
        mel, energy = self._framed_mel(y, B, N, lens)
        return mel, energy, lens.to(torch.int64) // self.hop_length + 1

    def _framed_mel(self, y, B, N, lens):
        hop, taps, P = self.hop_length, self.filter_length // self.hop_length, self.filter_length // 2
        frames = N // hop + 1
        S = frames + taps - 1                                               # rows of `hop` samples per utterance
        xp = torch.empty(B, S * hop, device=y.device, dtype=torch.float32)
        if lens is None:
            _lib.call("fs2_reflect_pad", y.data_ptr(), xp.data_ptr(), B, N, P, S * hop, ops._stream())
        else:
            _lib.call("fs2_reflect_pad_ragged", y.data_ptr(), N, lens.data_ptr(), xp.data_ptr(), B, P, S * hop, ops._stream())
        nft = 2 * self.cutoff
        ft = torch.empty(B * S, (nft + 3) // 4 * 4, device=y.device, dtype=torch.float32)[:, :nft]   # 16-B aligned rows
        ops.conv_gemm(xp.view(B * S, hop), self.forward_basis, None, S, taps=taps, pad=0, out=ft)
        mel = torch.empty(B, self.n_mel_channels, frames, device=y.device, dtype=torch.float32)
        energy = torch.empty(B, frames, device=y.device, dtype=torch.float32)
        _lib.call("fs2_stft_mel_epilogue", ft.data_ptr(), ft.stride(0), self.mel_basis.data_ptr(), self.mel_span.data_ptr(),
                  mel.data_ptr(), energy.data_ptr(), B, S, frames, self.cutoff, self.n_mel_channels, 1e-5, ops._stream())
        return mel, energy


def get_mel_from_wav(audio, _stft):
    """audio/tools.py:8-15: 1-D float array in [-1, 1] -> (mel (n_mel, frames), energy (frames,)) as float32 numpy."""
    dev = _stft.forward_basis.device if _stft.forward_basis.is_cuda else torch.device("cuda")
    a = torch.clip(torch.as_tensor(np.asarray(audio), dtype=torch.float32).unsqueeze(0), -1, 1).to(dev)
    mel, energy = _stft.mel_spectrogram(a)
    return mel[0].cpu().numpy().astype(np.float32), energy[0].cpu().numpy().astype(np.float32)
