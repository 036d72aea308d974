"""Throughput of the corpus mel extraction (device stage of fastspeech2_amd/preprocess.py): a ragged batch of LJSpeech-like
utterances through TacotronSTFT.mel_spectrogram_ragged, inputs resident in HBM; then the same including the pinned H2D / D2H
copies; the CPU oracle on a bounded sample beside it.  Prints one JSON line."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--utts", type=int, default=256)
    ap.add_argument("--steps", type=int, default=10)
    args = ap.parse_args()
    from fastspeech2_amd.audio import TacotronSTFT
    from fastspeech2_amd import ops
    dev = torch.device("cuda", 0)
    stft = TacotronSTFT(1024, 256, 1024, 80, 22050, 0, 8000).to(dev)
    rng = np.random.default_rng(0)
    lens = np.sort(rng.uniform(1.1, 10.1, size=args.utts) * 22050).astype(np.int64)[::-1].copy()   # LJSpeech clip lengths
    N = int(lens.max())
    host = torch.empty(args.utts, N, dtype=torch.float32).pin_memory()
    host.uniform_(-0.5, 0.5)
    y = host.to(dev)
    lens_t = torch.tensor(lens, dtype=torch.int32, device=dev)
    frames = int((lens // 256 + 1).sum())
    padded_frames = args.utts * (N // 256 + 1)

    def dev_step():
        return stft.mel_spectrogram_ragged(y, lens_t)

    def full_step():
        yy = host.to(dev, non_blocking=True)
        mel, en, _ = stft.mel_spectrogram_ragged(yy, lens_t)
        return mel.cpu(), en.cpu()

    out = {}
    for name, fn in (("resident", dev_step), ("with_copies", full_step)):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            fn()
        torch.cuda.synchronize()
        out[name] = (time.perf_counter() - t0) / args.steps
    # the framed-DFT GEMM alone (HIP events on the launch stream)
    ops.PROFILE = {}
    for _ in range(3):
        dev_step()
    torch.cuda.synchronize()
    prof, ops.PROFILE = ops.PROFILE, None
    rec = prof.get("conv_gemm", [])
    gemm_ms = sum(r[1].elapsed_time(r[2]) for r in rec) / max(len(rec), 1)
    gemm_flop = sum(r[0] for r in rec) / max(len(rec), 1)
    # CPU oracle on one 10 s utterance
    from oracle import fs2_oracle as O
    yc = host[:1, :220500].clone()
    O.mel_spectrogram(yc)
    t0 = time.perf_counter()
    O.mel_spectrogram(yc)
    cpu_s = time.perf_counter() - t0
    audio_s = float(lens.sum()) / 22050
    print(json.dumps({
        "workload": f"{args.utts} utterances, {audio_s:.0f} s of audio, {frames} valid frames ({padded_frames} padded), fp32",
        "resident_ms": round(out["resident"] * 1e3, 3), "valid_frames_per_s": round(frames / out["resident"], 1),
        "x_realtime": round(audio_s / out["resident"], 1),
        "with_copies_ms": round(out["with_copies"] * 1e3, 3), "with_copies_x_realtime": round(audio_s / out["with_copies"], 1),
        "dft_gemm_ms": round(gemm_ms, 3), "dft_gemm_tflops": round(gemm_flop / (gemm_ms * 1e-3) / 1e12, 1), "fp32_mfma_peak": 157.3,
        "cpu_oracle_frames_per_s": round((220500 // 256 + 1) / cpu_s, 1), "cpu_threads": torch.get_num_threads()}))


if __name__ == "__main__":
    main()
