#!/usr/bin/env python
"""Config 4 (LibriTTS-shaped, variable-length bucketed batches): what the sampler's sorting window costs in padding, measured.

The reference sorts inside windows of group_size = 4 batches (train.py:30-37, dataset.py:127-146); `data.BucketedBatchSampler`
keeps that as its default (window = group_size x world x batch items).  For every window size in --groups this script
  * computes the WHOLE epoch's valid-row fraction (phoneme rows and mel rows) from the sampler's own step lists, and
  * times the train step (fwd + loss + bwd + clip + Adam, bf16, side stream on) on --nsteps batches spread evenly over the epoch
    (a single batch says little: inside a window the steps run from its longest to its shortest utterances),
and prints valid mel-frames/s, padded mel-frames/s and the fraction of the bf16 MFMA peak over the sampled steps.  --world W deals
the steps as a W-rank job would (rank 0's batches: at W = 8 the window is 8 x wider at the same group_size).

    python tools/bench_libritts_sweep.py --groups 4,16,64 --nsteps 8 [--world 1]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fastspeech2_amd  # noqa: E402

import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--groups", default="4,16,64")
    ap.add_argument("--nsteps", type=int, default=0, help="0 = EVERY step of the epoch (the same utterances for every window size: "
                    "frames/s are comparable); n > 0 = n steps spread evenly over the epoch")
    ap.add_argument("--world", type=int, default=1)
    ap.add_argument("--reps", type=int, default=4)
    ap.add_argument("--batch", type=int, default=48)
    a = ap.parse_args()
    fastspeech2_amd.configure_hw_queues()
    import bench
    from fastspeech2_amd.data import BucketedBatchSampler
    from fastspeech2_amd.synthetic import synthetic_batch
    from fastspeech2_amd.utils import lens_to_device

    args = bench.parse(["--workload", "libritts", "--batch", str(a.batch)])
    device = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    torch.cuda.set_stream(torch.cuda.Stream(device=device, priority=-1))
    model, loss_fn, opt, b0, pcfg, mcfg = bench.build(args, device, 0, 1)
    g = torch.Generator().manual_seed(99)
    pool = torch.clamp(torch.exp(torch.randn(8192, generator=g) * 0.78 + 3.89), 5, 250).long()     # bench.build's pool
    frames_per_phoneme = 7.0                                                                        # dur_lo = 4 .. dur_hi = 10
    # a CORPUS: every utterance has its own per-phoneme durations (4 .. 10 frames, seeded by its index), shrunk on its own until it
    # fits max_seq_len = 1000 frames - so the epoch's valid frames are the same for every window size (the per-batch generator
    # shrinks a whole batch when its longest utterance overflows, which made the totals depend on how the utterances were dealt)
    utt_dur = []
    for i in range(len(pool)):
        d = torch.randint(4, 11, (int(pool[i]),), generator=torch.Generator().manual_seed(50000 + i))
        while int(d.sum()) > 1000:
            d = torch.clamp(d - 1, min=1)
        utt_dur.append(d)
    utt_frames = torch.tensor([int(d.sum()) for d in utt_dur])
    out = []
    for G in [int(x) for x in a.groups.split(",")]:
        steps = list(iter(BucketedBatchSampler(pool.numpy(), a.batch, world_size=a.world, rank=0, group_size=G, shuffle=True, seed=1234)))
        lens = [pool[s].numpy() for s in steps]
        src_valid = sum(int(l.sum()) for l in lens) / sum(int(l.max()) * len(l) for l in lens)
        # mel rows: T = sum of per-phoneme durations (4..10), capped by max_seq_len = 1000 - from the real batches below for the sample,
        # from the phoneme counts for the epoch (the same ratio up to the cap)
        mel_valid_epoch = sum(int(utt_frames[s].sum()) for s in steps) / sum(int(utt_frames[s].max()) * len(s) for s in steps)
        pick = [(2 * i + 1) * len(steps) // (2 * a.nsteps) for i in range(a.nsteps)] if a.nsteps > 0 else list(range(len(steps)))
        tot_valid = tot_padded = 0.0
        tot_t = 0.0
        per = []
        for si in pick:
            order = sorted(steps[si], key=lambda i: -int(pool[i]))                                 # the collate function sorts by text length
            b = synthetic_batch(1234 + si, 0, 0, n_speaker=2456, src_lens=pool[order].tolist(), sort=False,
                                utt_durations=[utt_dur[i] for i in order])
            b = {k: (lens_to_device(v, device) if k in ("src_lens", "mel_lens") else v.to(device) if isinstance(v, torch.Tensor) else v)
                 for k, v in b.items()}
            step, _ = bench.make_step(model, loss_fn, opt, b, None)
            for _ in range(2 if a.nsteps == 0 else 3):
                step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.reps):
                step()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / a.reps
            valid, padded = float(b["mel_lens"].sum()), float(a.batch * b["max_mel_len"])
            tot_valid += valid; tot_padded += padded; tot_t += dt
            per.append({"step": si, "L": int(b["max_src_len"]), "T": int(b["max_mel_len"]), "valid": round(valid / padded, 3), "ms": round(dt * 1e3, 3)})
        rec = {"group_size": G, "world": a.world, "window_items": G * a.world * a.batch, "epoch_steps": len(steps),
               "epoch_valid_phoneme_rows": round(src_valid, 4), "epoch_valid_mel_rows": round(mel_valid_epoch, 4),
               "sample_valid_mel_rows": round(tot_valid / tot_padded, 4),
               "valid_frames_per_s": round(tot_valid / tot_t, 1), "padded_frames_per_s": round(tot_padded / tot_t, 1),
               "ms_per_step_mean": round(tot_t / len(pick) * 1e3, 3),
               "step_frac_of_peak_valid_rows": round(tot_valid / tot_t * bench.train_flop_per_frame(args) / 2.5e15, 4),
               "step_frac_of_peak_padded_rows": round(tot_padded / tot_t * bench.train_flop_per_frame(args) / 2.5e15, 4),
               "steps": per if a.nsteps > 0 else per[::17]}
        out.append(rec)
        print(json.dumps(rec), flush=True)
    return out


if __name__ == "__main__":
    main()
