"""Independent computation of the 80 x 513 Slaney mel filterbank (librosa == 0.7.2 `filters.mel(22050, 1024, 80, 0, 8000)`,
reference audio/stft.py:145-147) -> tests/golden/mel_filterbank.json (checksums + samples).

librosa is absent from the reference tree and from this image, so the product's table (fastspeech2_amd/audio.py) and the
oracle's (oracle/fs2_oracle.py) are restatements of its published definition.  This script is a THIRD implementation that
shares no code and no arithmetic with them: scalar loops in 60-digit decimal arithmetic (Decimal.ln / Decimal.exp), the
Slaney scale written out from its definition (linear 200/3 Hz per mel below 1 kHz, 27 log-spaced steps per factor 6.4 above),
then the two float32 roundings of librosa's float32 table (triangle -> float32; float32 x float64 area norm -> float32).
A float64 evaluation differs from the exact value by ~1e-16 relative, a float32 rounding boundary is hit with probability
~1e-9 per entry: the tables must agree BIT FOR BIT, and tests/test_host_logic_cpu.py checks exactly that (sha256 of the
float32 bytes)."""
import hashlib
import json
import os
import struct
from decimal import Decimal, getcontext

import numpy as np

getcontext().prec = 60
SR, N_FFT, N_MELS, FMIN, FMAX = 22050, 1024, 80, Decimal(0), Decimal(8000)
F_SP = Decimal(200) / Decimal(3)
MIN_LOG_HZ = Decimal(1000)
MIN_LOG_MEL = MIN_LOG_HZ / F_SP
LOGSTEP = Decimal("6.4").ln() / Decimal(27)


def hz_to_mel(f):
    return MIN_LOG_MEL + (f / MIN_LOG_HZ).ln() / LOGSTEP if f >= MIN_LOG_HZ else f / F_SP


def mel_to_hz(m):
    return MIN_LOG_HZ * (LOGSTEP * (m - MIN_LOG_MEL)).exp() if m >= MIN_LOG_MEL else F_SP * m


def f32(x):
    """nearest float32 of a Decimal (via the nearest float64: exact enough unless x sits within 1e-16 of a float32 tie)"""
    return np.float32(float(x))


def main():
    n_freq = 1 + N_FFT // 2
    fft_f = [Decimal(SR) / 2 * Decimal(i) / Decimal(n_freq - 1) for i in range(n_freq)]
    m_lo, m_hi = hz_to_mel(FMIN), hz_to_mel(FMAX)
    mel_f = [mel_to_hz(m_lo + (m_hi - m_lo) * Decimal(i) / Decimal(N_MELS + 1)) for i in range(N_MELS + 2)]
    table = np.zeros((N_MELS, n_freq), dtype=np.float32)
    for i in range(N_MELS):
        lo, ce, hi = mel_f[i], mel_f[i + 1], mel_f[i + 2]
        enorm = float(Decimal(2) / (hi - lo))                       # librosa holds enorm in float64
        for k in range(n_freq):
            f = fft_f[k]
            tri = min((f - lo) / (ce - lo), (hi - f) / (hi - ce))
            if tri <= 0:
                continue
            table[i, k] = np.float32(float(f32(tri)) * enorm)      # float32 triangle, float64 product, float32 result
    raw = (table + np.float32(0)).tobytes()                        # (-0.0 -> +0.0: numpy's maximum(0, -0.0) keeps the sign at [0, 0])
    out = {"what": "librosa==0.7.2 filters.mel(22050, 1024, 80, 0, 8000): independent 60-digit decimal computation",
           "shape": list(table.shape), "sha256_float32_le": hashlib.sha256(raw).hexdigest(),
           "sum": float(table.astype(np.float64).sum()), "nonzero": int((table != 0).sum()),
           "first_nonzero": [int(np.nonzero(r)[0][0]) for r in table], "last_nonzero": [int(np.nonzero(r)[0][-1]) for r in table],
           "samples": {f"{i},{k}": struct.unpack("<I", struct.pack("<f", float(table[i, k])))[0]
                       for i, k in ((0, 1), (0, 2), (10, 25), (40, 110), (79, 350), (79, 370))}}
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "mel_filterbank.json")
    json.dump(out, open(path, "w"), indent=1)
    print(out["sha256_float32_le"], out["sum"], out["nonzero"])


if __name__ == "__main__":
    main()
