#!/usr/bin/env python
"""bench.py — mel-frames/s of the FastSpeech 2 TRAIN step on MI355X (BASELINE.json metric), one process per GPU.

    python bench.py --gpus N --steps K --warmup W            (N>1: under torch.distributed.run, or bare - it then re-executes
                                                              itself under torch.distributed.run with N ranks)

A "step" = forward + FastSpeech2Loss + backward + (RCCL gradient all-reduce) + global-norm clip + Adam over one
synthetic LJSpeech-shaped batch per GPU (config 2 of BASELINE.json: 4+4 FFT layers, d=256, 2 heads, 80-bin mel,
batch 48 per GPU, ~128 phonemes -> ~900 frames, bf16 compute with fp32 master weights), inputs resident in HBM.
value = sum of valid mel frames over all ranks and K steps / max-over-ranks wall time.

The same JSON line carries
  roofline     — the dominant kernel (conv_gemm MFMA contraction): algorithmic FLOPs of its launches / their HIP-event
                 durations, measured in an instrumented replay of the same step right after the timed region; plus
                 `wgrad_family`: the weight-gradient kernels of the same replay (the largest block of device time);
  cpu_baseline — the CPU oracle (a port of the reference algorithm, oracle/fs2_oracle.py; the reference tree itself is not
                 on the GPU box) timed on this host's cores on a bounded sample of the same workload: B=4, BASELINE.md §2's
                 10 steps with 2 discarded, at the best of a torch-thread sweep.  Reported, not a target;
  config.synth — the second half of BASELINE's metric, batch-synthesis RTF on a val.txt-shaped workload (N = 1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import fastspeech2_amd  # noqa: E402  (main() calls configure_hw_queues() before the first HIP call - importing changes nothing)

import torch  # noqa: E402

MFMA_PEAK_TFLOPS = {"bf16": 2500.0, "fp32": 157.3}     # dense, /opt/skills/guides/MI355X_MICROARCH.md
TRAIN_FLOP_PER_FRAME = 117.5e6                         # 3 x 39.17 MFLOP fwd (4+4, phoneme-level)


def train_flop_per_frame(args):
    """SURVEY §8(d), 3 x forward: 4+4 phoneme-level 117.5, 4+4 frame-level (paper) 121.6, 4+6 157.6 (+4.1 frame-level) MFLOP/frame."""
    return (TRAIN_FLOP_PER_FRAME + (args.dec_layers - 4) * 20.05e6 + (4.1e6 if args.frame_level else 0.0))


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--batch", type=int, default=48)
    ap.add_argument("--phonemes", type=int, default=128)
    ap.add_argument("--graph", type=int, default=0,
                    help="1: replay the train step from a hipGraph.  Default 0 (eager): weight gradients run on a side HIP stream "
                         "concurrently with the data-gradient chain, which the graph executor serialises (r01i A/B: 12.8 vs 14.2 ms)")
    ap.add_argument("--workload", default="ljspeech", choices=["ljspeech", "libritts"],
                    help="ljspeech: BASELINE configs[1] (the reported metric).  libritts: configs[3]'s shape per GPU - multi-speaker "
                         "(2456-way speaker embedding), LibriTTS-like phoneme counts (median ~49, p95 ~176), length-bucketed batch")
    ap.add_argument("--group-size", type=int, default=4,
                    help="libritts: sorting window of the BucketedBatchSampler in steps (window = group_size x world x batch items; 4 = the "
                         "reference's `group_size` in train.py:30-37, the compatibility default; wider windows leave less padding per batch)")
    ap.add_argument("--libri-step", type=float, default=1.0 / 3,
                    help="libritts: which step of the epoch is the measured batch, as a fraction of the epoch (default: the step a third in)")
    ap.add_argument("--main-prio", type=int, default=-1,
                    help="priority of the stream the step runs on (-1 = high: the dispatcher prefers the critical fwd/dgrad chain over "
                         "the side stream's weight gradients; 0 = run on the default stream)")
    ap.add_argument("--host-time", action="store_true", help="also report the host-side issue time of one step (no device sync)")
    ap.add_argument("--mode", default="train", choices=["train", "synth"],
                    help="train: mel-frames/s of the train step (default, the driver's metric); synth: batch-synthesis RTF")
    ap.add_argument("--synth-batch", type=int, default=8, help="utterances per synthesis batch (synthesize.py:199 uses 8)")
    ap.add_argument("--vocoder-dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--dec-layers", type=int, default=4, help="4 = BASELINE's 4+4 (the reported metric); 6 = the reference's stock model.yaml")
    ap.add_argument("--frame-level", action="store_true", help="frame-level pitch/energy (the paper's configuration) instead of phoneme-level")
    ap.add_argument("--side-stream", type=int, default=1, help="1: weight gradients on a side HIP stream (default, what train.py does); "
                                                                "0: single stream (for per-kernel profiling)")
    ap.add_argument("--windows", type=int, default=5, help="the timed region is repeated this many times (each = --steps steps, "
                                                          "barrier + sync on both sides); the MEDIAN window is reported")
    ap.add_argument("--no-fp32", action="store_true", help="skip the secondary fp32 (the reference's own arithmetic) step time")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-graph-line", action="store_true", help="skip the hipGraph replay of the same step reported beside the eager number")
    ap.add_argument("--no-synth", action="store_true", help="train mode: skip the batch-synthesis RTF object (config.synth)")
    ap.add_argument("--synth-voc-streams", type=int, default=3, help="synthesis A/B: vocoder streams of utils.SynthPipeline (its default: 3)")
    ap.add_argument("--no-synth-pipeline", action="store_true", help="synthesis A/B: the reference's sequential per-batch loop instead of "
                    "utils.SynthPipeline")
    ap.add_argument("--no-fuse-resblocks", action="store_true", help="synthesis A/B: HiFi-GAN's narrow-stage residual blocks as six launches "
                                                                      "each (the round-4 path) instead of one fused launch")
    ap.add_argument("--hw-queues", type=int, default=fastspeech2_amd.HW_QUEUES_DEFAULT,
                    help="GPU_MAX_HW_QUEUES of every rank (fastspeech2_amd.configure_hw_queues: the SAME at every --gpus N, so the N-rank "
                         "line is the configuration the one-rank line was measured on; an exported value wins; 0 = runtime default)")
    ap.add_argument("--cpu-threads", default="8,16,32,64", help="torch thread counts the CPU baseline sweeps (those <= cpu_count)")
    a = ap.parse_args(argv)
    argv = sys.argv if argv is None else argv
    if a.mode == "synth":                        # one pass over the val.txt-shaped batches by default (64 batches of 8)
        if "--steps" not in argv:
            a.steps = 64
        if "--warmup" not in argv:
            a.warmup = 64
    return a


def dev_environment():
    """Tamper evidence: the shipped library has no environment switches, but a development build (FS2_LIB_PATH -> -DFS2_DEV)
    does (ablations that SKIP work).  The bench refuses to run with any of them set and records which library it loaded."""
    # (FS2_LIB_PATH selects another BUILD of the library for same-box A/B runs; the line records which file was loaded)
    bad = sorted(k for k in os.environ if k.startswith("FS2_") and k not in ("FS2_BENCH_BACKEND", "FS2_BENCH_SHARE_GPU", "FS2_LIB_PATH"))
    if bad:
        print(f"bench.py: refusing to run with development variables set: {bad}", file=sys.stderr)
        sys.exit(3)
    from fastspeech2_amd import _lib
    return os.path.relpath(_lib.LIB_PATH, ROOT)


def build(args, device, rank, world, dtype=None):
    from fastspeech2_amd import synthetic as configs
    from fastspeech2_amd.synthetic import synthetic_batch
    from fastspeech2_amd.model import FastSpeech2, FastSpeech2Loss, ScheduledOptim
    from fastspeech2_amd import ddp

    libri = args.workload == "libritts"
    if libri:
        import json as _json, tempfile
        d = tempfile.mkdtemp(prefix="fs2_libritts_")
        _json.dump({f"spk{i}": i for i in range(2456)}, open(os.path.join(d, "speakers.json"), "w"))
        _json.dump(configs.LJ_STATS, open(os.path.join(d, "stats.json"), "w"))
    pcfg, mcfg = configs.make_configs(dec_layers=args.dec_layers, enc_layers=4, multi_speaker=libri, frame_level=args.frame_level)
    if libri:
        pcfg["path"]["preprocessed_path"] = d
    torch.manual_seed(1234)
    model = FastSpeech2(pcfg, mcfg, compute_dtype=dtype or args.dtype).to(device)
    model.train()
    model._ensure_flat(device)
    model._engine.use_side_stream = bool(args.side_stream)
    if libri:
        # a LibriTTS-like pool (log-normal phoneme counts: median ~49, p95 ~176) dealt by the REAL sampler of train.py
        # (fastspeech2_amd/data.BucketedBatchSampler: shuffled windows of 4 steps, sorted by length inside a window, each
        # step's world*batch items dealt card-wise): every rank gets ITS OWN bucket of the same step - similar but not
        # identical length profiles, so the line's per-rank times show what the slowest rank costs
        from fastspeech2_amd.data import BucketedBatchSampler
        g = torch.Generator().manual_seed(99)
        pool = torch.clamp(torch.exp(torch.randn(8192, generator=g) * 0.78 + 3.89), 5, 250).long()
        sampler = BucketedBatchSampler(pool.numpy(), args.batch, world_size=world, rank=rank, group_size=args.group_size, shuffle=True, seed=1234)
        steps = list(iter(sampler))
        idxs = steps[min(len(steps) - 1, int(len(steps) * args.libri_step))]    # one fixed step of the epoch (every rank picks the same step)
        b = synthetic_batch(1234 + rank, 0, 0, dur_lo=4, dur_hi=10, n_speaker=2456, src_lens=pool[idxs].tolist())
    else:
        # every rank gets the SAME length profile (what the length-bucketed card-wise sampler of fastspeech2_amd/data.py deals
        # per step, so no rank waits for another's longer batch) with its OWN contents (phoneme ids, mels, pitch, energy)
        b = synthetic_batch(1234, args.batch, args.phonemes, dur_lo=4, dur_hi=10, min_len_frac=0.75, frame_level=args.frame_level)
        if rank > 0 and not args.frame_level:
            g = torch.Generator().manual_seed(1234 + rank)
            L, T = b["max_src_len"], b["max_mel_len"]
            sv = torch.arange(L).unsqueeze(0) < b["src_lens"].unsqueeze(1)
            mv = (torch.arange(T).unsqueeze(0) < b["mel_lens"].unsqueeze(1)).unsqueeze(-1)
            b["texts"] = torch.randint(1, 361, (args.batch, L), generator=g) * sv
            b["mels"] = (torch.clamp(torch.randn(args.batch, T, 80, generator=g) * 2 - 5, -11.5, 2.0) * mv).float()
            b["pitches"] = (torch.clamp(torch.randn(args.batch, L, generator=g), -2.917, 11.391) * sv).float()
            b["energies"] = (torch.clamp(torch.randn(args.batch, L, generator=g), -1.431, 8.184) * sv).float()
    from fastspeech2_amd.utils import lens_to_device
    b = {k: (lens_to_device(v, device) if k in ("src_lens", "mel_lens") else v.to(device) if isinstance(v, torch.Tensor) else v) for k, v in b.items()}
    import torch.distributed as _dist
    loss_fn = FastSpeech2Loss(pcfg, mcfg, count_reduce=ddp.CountExchange() if (world > 1 or _dist.is_initialized()) else None)
    opt = ScheduledOptim(model, configs.TRAIN, mcfg, 0)
    opt._ensure()
    return model, loss_fn, opt, b, pcfg, mcfg


def make_step(model, loss_fn, opt, b, exchange):
    batch12 = (None, None, b["speakers"], b["texts"], b["src_lens"], b["max_src_len"], b["mels"], b["mel_lens"],
               b["max_mel_len"], b["pitches"], b["energies"], b["durations"])

    T_dec = min(int(b["max_mel_len"]), int(model.model_config["max_seq_len"]))

    def fwd_bwd():
        if loss_fn.count_reduce is not None:                # the valid-position counts travel while the forward pass runs
            loss_fn.count_reduce.start(b["src_lens"], b["mel_lens"], int(b["max_src_len"]), T_dec)
        out = model(*batch12[2:])
        losses = loss_fn(batch12, out)
        losses[0].backward()
        return losses[0].detach()

    def step():
        loss = fwd_bwd()
        if exchange is not None:
            exchange.finish()
        opt.step_and_update_lr(zero_grad=True)      # clip + Adam + bf16 weight refresh + gradient clear: one pass
        return loss

    return step, fwd_bwd


def capture_graph(model, opt, fwd_bwd, fork=False):
    """hipGraph of forward + loss + backward + clip/Adam (lr / bias corrections are read from device memory).  Returns
    (graph, static_loss, replay) - `replay()` advances the host-side optimiser state and replays.  The caller must have set
    model._engine.device_seed = True BEFORE its warm-up steps: the dropout position then lives in device memory and is bumped
    by a kernel inside the graph (every replay draws fresh masks); an eager-mode capture would bake one step's seed into the
    kernel arguments.  BatchNorm's backward workspaces are cleared inside the capture (engine.py), so replays do not
    accumulate into each other.
    fork = False: ONE stream inside the capture (Engine._side_begin enforces that while a capture is in progress).
    fork = True: the weight-gradient / variance-predictor branches fork onto the side stream INSIDE the capture, as the eager
    step does (Engine.fork_in_capture).  Rounds 3-5 measured such a capture 2.8e-3 away from the eager gradients; round 6 found
    the cause (a redundant event recorded on the capture's origin stream from inside a branch section, Engine._wgrad) and
    tests/test_graph_gpu.py::test_forked_capture_* holds it to the eager step now."""
    assert model._engine.device_seed, "set model._engine.device_seed = True before warm-up and capture"
    graph = torch.cuda.CUDAGraph()
    opt.zero_grad()
    torch.cuda.synchronize()
    was = model._engine.fork_in_capture
    model._engine.fork_in_capture = bool(fork)
    try:
        with torch.cuda.graph(graph):
            static_loss = fwd_bwd()
            opt.apply_update(zero_grad=True)
    finally:
        model._engine.fork_in_capture = was
    torch.cuda.synchronize()

    def replay():
        opt.current_step += 1
        opt._adam_step += 1
        lr = opt.init_lr * opt._get_lr_scale()
        b1, b2 = opt.betas
        opt.set_hyper(lr, 1 - b1 ** opt._adam_step, 1 - b2 ** opt._adam_step)
        graph.replay()

    return graph, static_loss, replay


def _thread_counts(args):
    n = os.cpu_count() or 1
    cand = sorted({int(t) for t in str(args.cpu_threads).split(",") if t.strip()})
    cand = [t for t in cand if t <= n] or [min(n, 8)]
    return cand


def cpu_baseline(args):
    """CPU oracle (port of the reference algorithm) on a bounded sample: B=4, same L/T, fwd+loss+bwd+clip+Adam.
    Protocol: the torch thread count is SWEPT (a B=4 step oversubscribed onto every
    logical CPU is slower than on 8-32 threads), 3 steps each with the first discarded; then 10 steps at the best count, 2
    discarded, MEDIAN of the remaining 8.  kind = "port": /root/reference is not shipped to the GPU box; the port follows it
    line by line (oracle/fs2_oracle.py) and runs at the unmodified reference's speed (same box, 8 cores: 0.80-1.03 vs 0.94-1.4 s)."""
    from fastspeech2_amd import synthetic as configs
    from oracle import fs2_oracle as O
    from fastspeech2_amd.synthetic import synthetic_batch
    from fastspeech2_amd.model import FastSpeech2

    pcfg, mcfg = configs.make_configs(dec_layers=4, enc_layers=4)
    torch.manual_seed(1234)
    sd = {k: v.clone() for k, v in FastSpeech2(pcfg, mcfg).state_dict().items()}
    params = []
    for k, v in sd.items():
        if v.is_floating_point() and not any(s in k for s in ("position_enc", "_bins", "running_")):
            v.requires_grad_(True)
            params.append(v)
    opt = torch.optim.Adam(params, betas=(0.9, 0.98), eps=1e-9)
    b = synthetic_batch(1234, 4, args.phonemes, dur_lo=4, dur_hi=10, min_len_frac=0.75)
    bn = {k: v for k, v in sd.items() if "running_" in k}

    def one():
        t0 = time.perf_counter()
        out = O.fastspeech2_forward(sd, mcfg, pcfg, b["speakers"], b["texts"], b["src_lens"], b["max_src_len"], b["mels"],
                                    b["mel_lens"], b["max_mel_len"], b["pitches"], b["energies"], b["durations"],
                                    training=True, dropout=True, bn_buffers=bn)
        loss = O.fastspeech2_loss(pcfg, (b["mels"], b["pitches"], b["energies"], b["durations"]), out)[0]
        opt.zero_grad()
        loss.backward()
        torch.nn.utils.clip_grad_norm_(params, 1.0)
        opt.step()
        return time.perf_counter() - t0

    prev = torch.get_num_threads()
    sweep = {}
    try:
        for nt in _thread_counts(args):
            torch.set_num_threads(nt)
            ts = [one() for _ in range(3)]
            sweep[nt] = sum(ts[1:]) / 2
        best = min(sweep, key=sweep.get)
        torch.set_num_threads(best)
        ts = sorted([one() for _ in range(10)][2:])
        t = 0.5 * (ts[3] + ts[4])                            # median of 8
    finally:
        torch.set_num_threads(prev)
    frames = int(b["mel_lens"].sum())
    return {"value": round(frames / t, 1), "unit": "mel-frames/s", "cores": best, "kind": "port",
            "sample": f"oracle train step (port of the reference; the reference tree is not on the GPU box), B=4, L={args.phonemes}, "
                      f"T={b['max_mel_len']}, {best} torch threads (best of sweep "
                      + ", ".join(f"{k}: {v:.2f} s" for k, v in sorted(sweep.items()))
                      + f"), 10 steps, 2 discarded, median of 8 = {t:.3f} s/step, host cpu_count={os.cpu_count()}"}


def synth_cpu_baseline(args, lens8, threads):
    """the oracle's batch synthesis (acoustic model + HiFi-GAN + int16) of ONE val.txt-shaped batch of 8 on the host cores."""
    import math
    from fastspeech2_amd import synthetic as configs
    from fastspeech2_amd.synthetic import synthetic_batch
    from fastspeech2_amd.model import FastSpeech2
    from fastspeech2_amd import hifigan, utils
    from oracle import fs2_oracle as O

    pcfg, mcfg = configs.make_configs(dec_layers=4, enc_layers=4)
    torch.manual_seed(1234)
    model = FastSpeech2(pcfg, mcfg)
    with torch.no_grad():
        model.variance_adaptor.duration_predictor.linear_layer.bias.fill_(math.log(8.0))
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    h = hifigan.AttrDict(utils.HIFIGAN_V1)
    voc = hifigan.Generator(h)
    vsd = O.remove_weight_norm_sd({k: v.clone() for k, v in voc.state_dict().items()})
    b = synthetic_batch(4321, 0, 0, src_lens=lens8, sort=False)
    prev = torch.get_num_threads()
    torch.set_num_threads(threads)
    try:
        times = []
        for _ in range(2):
            t0 = time.perf_counter()
            with torch.no_grad():
                out = O.fastspeech2_forward(sd, mcfg, pcfg, b["speakers"], b["texts"], b["src_lens"], b["max_src_len"], training=False,
                                            dropout=False)
                wav = O.hifigan_forward(vsd, h, out[1].transpose(1, 2))
                pcm = O.pcm16(wav)
            times.append(time.perf_counter() - t0)
    finally:
        torch.set_num_threads(prev)
    t = min(times)
    audio_s = float(out[9].sum()) * 256 / 22050.0
    return {"value": round(t / audio_s, 5), "unit": "s wall per s audio", "cores": threads, "kind": "port",
            "sample": f"oracle acoustic model + HiFi-GAN + int16 on the first val.txt-shaped batch (8 utterances, {audio_s:.1f} s of audio), "
                      f"best of 2 ({t:.2f} s), {threads} torch threads"}


SYNTH_NAMES = {1: "conv_gemm_kernel", 2: "conv_gemm_dma_kernel", 3: "conv_gemm_ring_kernel", 4: "conv_skinny_kernel",
               5: "conv_gemm_p_kernel<false> (taps >= 3)", 6: "conv_gemm_p_kernel<true> (taps == 1)", 7: "conv_gemm_w_kernel (taps == 1, N % 256 == 0)",
               9: "conv_gemm_s_kernel (taps == 1, K == 256: weights in registers)",
               10: "resblock_fused_kernel (HiFi-GAN C = 32 / 64: a stage's three residual blocks in one launch)"}


def synth_measure(args, device, rank, world, steps, warmup, want_roofline):
    """BASELINE metric part 2: batch-synthesis real-time factor = wall(acoustic model + HiFi-GAN + int16 + PCM to host) / audio
    seconds.  Workload = the reference's `synthesize.py --mode batch --source val.txt` (synthesize.py:197-199): the 512
    utterances of LJSpeech val.txt in FILE ORDER, 8 per batch, unsorted, each batch padded to its own longest utterance - the
    phoneme counts are the real ones (fastspeech2_amd/workloads/ljspeech_val_phonemes.json, made from the reference by
    tools/make_val_shape.py), the phoneme ids are random, weights are random-init with the duration predictor's bias set so that
    ~7 frames per phoneme come out.  A step = one batch of 8; with N ranks the 64 batches are dealt round-robin (replicas only).
    Returns a dict of raw numbers (this rank's)."""
    import math
    from fastspeech2_amd import synthetic as configs
    from fastspeech2_amd.synthetic import synthetic_batch, val_phoneme_counts
    from fastspeech2_amd.model import FastSpeech2
    from fastspeech2_amd import hifigan, utils, ops

    pcfg, mcfg = configs.make_configs(dec_layers=4, enc_layers=4)
    torch.manual_seed(1234)
    model = FastSpeech2(pcfg, mcfg, compute_dtype=args.dtype)
    with torch.no_grad():
        model.variance_adaptor.duration_predictor.linear_layer.bias.fill_(math.log(8.0))
    model.to(device).eval()
    voc = hifigan.Generator(hifigan.AttrDict(utils.HIFIGAN_V1), compute_dtype=args.vocoder_dtype)
    voc.eval()
    voc.remove_weight_norm()
    voc.to(device)
    voc.fuse_resblocks = not args.no_fuse_resblocks
    B = args.synth_batch
    counts = val_phoneme_counts()
    groups = [counts[i:i + B] for i in range(0, len(counts), B)][rank::world]
    batches = []
    for gi, lens in enumerate(groups):
        b = synthetic_batch(4321 + 64 * rank + gi, 0, 0, src_lens=lens, sort=False)
        batches.append(([f"u{i}" for i in range(len(lens))], None, b["speakers"].to(device), b["texts"].to(device),
                        b["src_lens"].to(device), b["max_src_len"]))

    def step(i):
        batch = batches[i % len(batches)]
        with torch.no_grad():
            out = model(*batch[2:])
            wavs = utils.synth_samples(batch, out, voc, mcfg, pcfg, None, write=False)   # includes the D2H of the PCM
        return out, wavs

    # the product's loop (synthesize.synthesize): utils.SynthPipeline - batch i+1's acoustic model on one stream under batch i's
    # vocoder on another, PCM to the host behind an event; --no-synth-pipeline: the reference's sequential loop (A/B)
    pipe = None if args.no_synth_pipeline else utils.SynthPipeline(model, voc, (pcfg, mcfg), device=device, voc_streams=args.synth_voc_streams)

    def run(n):
        """n batches through the loop under test; yields each batch's host PCM list (in order)"""
        if pipe is None:
            for i in range(n):
                yield step(i)[1]
        else:
            for _batch, _out, wavs in pipe(batches[i % len(batches)] for i in range(n)):
                yield wavs

    for _ in run(warmup):
        pass
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
    torch.cuda.synchronize()
    audio_samples, frames = 0, 0
    t0 = time.perf_counter()
    for wavs in run(steps):                                     # (fill and drain of the pipeline are inside the timed region)
        audio_samples += sum(len(w) for w in wavs)            # (host lists: the PCM is already on the host)
        frames += int(sum(len(w) for w in wavs)) // 256
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    res = {"dt": dt, "audio_s": audio_samples / 22050.0, "frames": float(frames), "n_batches": len(batches),
           "first_batch": groups[0],
           "loop": "sequential (synthesize.py:87-103)" if pipe is None else
                   f"utils.SynthPipeline: acoustic model of the next batch on one stream under the vocoders of the previous "
                   f"{args.synth_voc_streams} on {args.synth_voc_streams} more"}
    if want_roofline:
        # instrumented replay: HIP events around every conv_gemm launch of the acoustic model + vocoder (one stream), over
        # every 8th batch of the pass
        ops.PROFILE = {}
        nrep = 0
        for i in range(0, len(batches), 8):
            step(i)
            nrep += 1
        torch.cuda.synchronize()
        prof, ops.PROFILE = ops.PROFILE, None
        by = {}
        for (f, e0, e1, var, _hl, _S) in prof.get("conv_gemm", []):
            d = by.setdefault(var, [0.0, 0.0, 0])
            d[0] += f; d[1] += e0.elapsed_time(e1); d[2] += 1
        if by:
            dom = max(by, key=lambda v: by[v][1])
            fl, ms, n = by[dom]
            peak = MFMA_PEAK_TFLOPS[args.vocoder_dtype]
            # the C = 32 / 64 residual-block convs (conv_skinny_kernel) are HBM-bound by construction (48-96 FLOP/B); every
            # other variant is priced against the MFMA peak
            res["roofline"] = {"bound": "mfma", "kernel": SYNTH_NAMES.get(dom, str(dom)) + " (%s)" % args.vocoder_dtype,
                               "achieved": round(fl / (ms * 1e-3) / 1e12, 1), "peak": peak, "unit": "TFLOP/s",
                               "frac": round(fl / (ms * 1e-3) / 1e12 / peak, 4), "traffic": None,
                               "launches_per_step": round(n / nrep, 1), "kernel_ms_per_step": round(ms / nrep, 3),
                               "avg_launch_us": round(ms / n * 1e3, 1), "gflop_per_launch": round(fl / n / 1e9, 2),
                               "conv_gemm_family": {v2: {"kernel": SYNTH_NAMES.get(v2, str(v2)), "launches_per_step": round(d[2] / nrep, 1),
                                                         "ms_per_step": round(d[1] / nrep, 3), "tflops": round(d[0] / (d[1] * 1e-3) / 1e12, 1)}
                                                    for v2, d in sorted(by.items())},
                               # HiFi-GAN: 614.1 MFLOP per mel frame
                               "step_frac_of_peak": round(frames / dt * 614.1e6 / (peak * 1e12), 4)}
    del model, voc, batches
    return res


def synth_workload_text(args):
    return (f"BASELINE configs[4]: batch synthesis shaped like synthesize.py --mode batch over LJSpeech val.txt - 512 utterances in "
            f"file order, {args.synth_batch} per batch (unsorted, padded per batch), real phoneme counts (13..132, mean 69.7), "
            f"~7 frames/phoneme, 4+4 FastSpeech2 + HiFi-GAN V1, PCM copied to host")


def synth_main(args):
    lib_used = dev_environment()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch.distributed as dist
    device = init_rank(world, local_rank)
    r = synth_measure(args, device, rank, world, args.steps, args.warmup, rank == 0 and not args.no_roofline)
    tt = torch.tensor([r["dt"], r["audio_s"], r["frames"]], device=device, dtype=torch.float64)
    if world > 1:
        tmax = tt[:1].clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(tt)
        tt[0] = tmax[0]
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = synth_cpu_baseline(args, r["first_batch"], min(_thread_counts(args)[-1], 32))
    if rank == 0:
        dt, audio_s, frames = tt.tolist()
        print(json.dumps({
            "metric": "batch-synth real-time factor (acoustic model + HiFi-GAN + int16, 22.05 kHz)", "value": round(dt / audio_s, 6),
            "unit": "s wall per s audio", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": False, "scaling": "weak", "vs_baseline": None,
            "dtype": f"{args.dtype} acoustic / {args.vocoder_dtype} vocoder", "data": "synthetic",
            "config": {"workload": synth_workload_text(args),
                       "audio_s_per_step": round(audio_s / args.steps / world, 2), "mel_frames_per_s": round(frames / dt, 1),
                       "x_realtime": round(audio_s / dt, 1), "loop": r["loop"], "library": lib_used, "dev_env": [],
                       "hw_queues": dict(fastspeech2_amd.HW_QUEUES)},
            "roofline": r.get("roofline"), "cpu_baseline": cpu}))
    if dist.is_initialized():
        dist.barrier()                     # ranks leave together (rank 0 was still replaying for the roofline)
        dist.destroy_process_group()


def self_launch(n):
    """`python bench.py --gpus N` started WITHOUT a launcher (the way the driver starts N = 1): re-execute this command line
    under `python -m torch.distributed.run` with N ranks on this node - exactly the launch the task contract names - and pass
    rank 0's JSON line through.  The torchrun form itself (WORLD_SIZE already set) never comes here."""
    import socket
    import subprocess
    ndev = torch.cuda.device_count()
    if ndev < n and os.environ.get("FS2_BENCH_SHARE_GPU") != "1":
        print(f"bench.py --gpus {n}: this node exposes {ndev} GPU(s); one rank per GPU needs {n} (FS2_BENCH_SHARE_GPU=1 + "
              f"FS2_BENCH_BACKEND=gloo put the ranks on one device for a functional check only)", file=sys.stderr)
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    if os.environ.get("FS2_BENCH_SHARE_GPU") == "1":
        env.setdefault("FS2_BENCH_BACKEND", "gloo")           # RCCL refuses two ranks on one device
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env, cwd=os.getcwd())


def init_rank(world, local_rank):
    """One process per GPU: bind this rank to its device and join the RCCL group (backend "nccl" is RCCL on ROCm).
    FS2_BENCH_BACKEND=gloo + FS2_BENCH_SHARE_GPU=1 exist only so that the N>1 code path can be exercised by the tests on a
    1-GPU box (RCCL refuses two ranks on one device); the driver's launch never sets them."""
    import torch.distributed as dist
    share = os.environ.get("FS2_BENCH_SHARE_GPU") == "1"
    dev_index = 0 if share else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    # a launcher (torch.distributed.run sets WORLD_SIZE) means a process group - ALSO for one rank: `python -m torch.distributed.run
    # --nproc-per-node 1 bench.py --gpus 1` then runs the same bucketed RCCL all-reduces under backward as the N-rank job (the
    # identity reduction); the bare `python bench.py --gpus 1` has no group and no exchange
    if world > 1 or "WORLD_SIZE" in os.environ:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = os.environ.get("FS2_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)
    return device


def pmc_traffic(kernel_substr):
    """HBM bytes per launch of the roofline kernel family, from the committed rocprofv3 PMC passes (FETCH_SIZE x2 gfx950
    correction + WRITE_SIZE, collected in separate runs by tools/pmc_traffic.py; counters cannot be read live here)."""
    import glob
    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_traffic.json")))
    if not cands:
        return None
    path = cands[-1]                                      # newest round's passes
    doc = json.load(open(path))
    from fastspeech2_amd._lib import kernel_source_sha
    if doc.get("kernel_source_sha") != kernel_source_sha():
        # counters measured on OTHER kernel sources say nothing about these: no number rather than a stale one
        print(f"[bench] {os.path.basename(path)} was measured on different contraction-kernel sources: roofline.traffic = null "
              f"(re-run tools/gpu_milestone.sh)", file=sys.stderr)
        return None
    k = doc["kernels"]
    # "conv_gemm_p_kernel<false>" names the ONE_TAP = false instantiations, whatever template arguments follow in rocprof's name
    pat = kernel_substr[:-1] if kernel_substr.endswith(">") else kernel_substr
    hit = lambda name: name.startswith(pat) or (" " + pat) in name
    n = sum(v["launches_per_step"] for name, v in k.items() if hit(name))
    b = sum(v["launches_per_step"] * (v["fetch_bytes_per_launch"] + v["write_bytes_per_launch"]) for name, v in k.items() if hit(name))
    return {"bytes_per_launch": round(b / n), "source": "profiles/%s (rocprofv3 --pmc, separate passes)" % os.path.basename(path)} if n else None


def main():
    args = parse()
    # first thing, before any HIP call and before the self-launch (whose ranks inherit the exported value): one hardware-queue
    # setting for every world size - recorded in the line as config.hw_queues
    if os.environ.get("FS2_BENCH_SHARE_GPU") == "1" and "--hw-queues" not in sys.argv:
        # the tests' functional check of the N > 1 code path: several ranks on ONE device.  Their hardware queues add up on that
        # device (2 x 16 + the exchange's exceed what the command processor schedules without time-slicing queues: rank 0's
        # event-bracketed kernels then carry other queues' slices), so this mode keeps the runtime default; one rank per GPU
        # - every real launch - gets the package's setting
        args.hw_queues = 0
    fastspeech2_amd.configure_hw_queues(args.hw_queues)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus))
    if args.mode == "synth":
        return synth_main(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch.distributed as dist
    lib_used = dev_environment()
    device = init_rank(world, local_rank)

    from fastspeech2_amd import ddp, ops
    if args.main_prio != 0:
        torch.cuda.set_stream(torch.cuda.Stream(device=device, priority=args.main_prio))
    model, loss_fn, opt, b, pcfg, mcfg = build(args, device, rank, world)
    exchange = None
    pg = dist.is_initialized()
    if pg:
        exchange = ddp.GradExchange(model.flat_gradients(), world)
        model._engine.grad_hook = exchange.ready
        dist.broadcast(model.flat_parameters(), 0)
    step, fwd_bwd = make_step(model, loss_fn, opt, b, exchange)

    use_graph = bool(args.graph) and not pg
    graph = None
    if use_graph:
        model._engine.device_seed = True         # dropout position in device memory from the first warm-up step on
    for _ in range(max(args.warmup, 3) if use_graph else args.warmup):
        loss = step()
    torch.cuda.synchronize()
    replay = None
    if use_graph:
        try:
            graph, static_loss, replay = capture_graph(model, opt, fwd_bwd)
        except Exception as e:  # capture is an optimisation, never a change of what is computed
            print(f"[bench] hipGraph capture failed ({type(e).__name__}: {e}); running eager", file=sys.stderr)
            graph = None
            torch.cuda.synchronize()

    def run_step():
        if graph is not None:
            replay()
        else:
            step()

    for _ in range(2):
        run_step()
    # the timed region (EXACTLY --steps steps between barrier + synchronize on both sides) is repeated --windows times and the
    # MEDIAN window is reported: one 20-step window is ~0.2 s, boxes differ by several per cent and clocks wander
    window_s = []
    for _ in range(max(1, args.windows)):
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            run_step()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        window_s.append(time.perf_counter() - t0)
    wt = torch.tensor(window_s, device=device, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(wt, op=dist.ReduceOp.MAX)              # per window: the slowest rank
    window_s = sorted(wt.tolist())
    dt = window_s[len(window_s) // 2]
    host_ms = None
    if args.host_time and graph is None:
        torch.cuda.synchronize()
        h0 = time.perf_counter()
        for _ in range(5):
            run_step()
        host_ms = (time.perf_counter() - h0) / 5 * 1e3        # time to ISSUE a step (the device is still running)
        torch.cuda.synchronize()
    frames = b["mel_lens"].sum().to(torch.float64)
    padded = torch.tensor([float(args.batch * b["max_mel_len"])], device=device, dtype=torch.float64)
    dist_info = {}
    if pg:
        dist.all_reduce(frames)
        dist.all_reduce(padded)
        # data-parallel sanity carried by the line itself: the group really has N ranks on this backend and, after the timed
        # steps, every replica holds bit-identical parameters (each saw its own batch; only the exchanged gradients couple them)
        assert dist.get_world_size() == world == args.gpus, (dist.get_world_size(), world, args.gpus)
        chk = torch.stack([model.flat_parameters().double().sum(), model.flat_parameters().double().abs().sum()])
        allchk = [torch.zeros_like(chk) for _ in range(world)]
        dist.all_gather(allchk, chk)
        dist_info = {"backend": dist.get_backend(), "world_size": dist.get_world_size(),
                     "replicas_bit_identical": all(torch.equal(c, allchk[0]) for c in allchk),
                     # what the exchange did in the last timed step: all_reduce calls launched by the engine's prefix hooks (under
                     # backward) and by finish() (after it), on the communication stream
                     "exchange": {"collectives_total": exchange.n_buckets, "last_step_under_backward": exchange.last_step[0],
                                  "last_step_in_finish": exchange.last_step[1], "flat_gradient_mb": round(exchange.n * 4 / 2 ** 20, 1)}}
    value = frames.item() * args.steps / dt
    final_loss = float((static_loss if graph is not None else loss).item())
    if pg:
        # what each rank's OWN batch costs without the exchange (5 local steps; the replicas diverge from here on, nothing
        # below depends on them): the data-parallel step runs at the slowest rank's pace
        model._engine.grad_hook = None
        loss_fn.count_reduce = None
        local_step, _ = make_step(model, loss_fn, opt, b, None)
        local_step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            local_step()
        torch.cuda.synchronize()
        mine = torch.tensor([(time.perf_counter() - t0) / 5 * 1e3, float(b["max_mel_len"]), float(b["max_src_len"]),
                             float(b["mel_lens"].sum())], device=device, dtype=torch.float64)
        allm = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allm, mine)
        ms = [float(x[0]) for x in allm]
        # the three numbers an N-rank line is read for: the slowest rank's LOCAL step (what the job cannot beat), the
        # data-parallel step next to it - their difference is what the exchange + the count all-reduce + the barrier skew cost on
        # top, i.e. the part of the gradient exchange that backward did not hide - and the spread of the ranks' local steps
        dist_info.update({"per_rank_local_ms": [round(x, 3) for x in ms], "slowest_over_fastest": round(max(ms) / min(ms), 3),
                          "exchange_exposed_ms": round(dt / args.steps * 1e3 - max(ms), 3),
                          "per_rank_T": [int(x[1]) for x in allm], "per_rank_L": [int(x[2]) for x in allm],
                          "per_rank_frames": [int(x[3]) for x in allm]})

    roofline = None
    if rank == 0 and not args.no_roofline:
        # instrumented eager replay of the same step: HIP events around every conv_gemm launch on its stream
        # (single stream for this replay: a kernel's event-bracketed duration is only its own when nothing runs beside it)
        side = model._engine.use_side_stream
        model._engine.use_side_stream = False
        prof_step = step
        if pg:
            # the other ranks are past the timed region: replay a LOCAL step (no gradient exchange, local loss counts) so that
            # rank 0 never enters a collective alone; kernel shapes and launches are the same
            model._engine.grad_hook = None
            loss_fn.count_reduce = None
            prof_step, _ = make_step(model, loss_fn, opt, b, None)
        ops.PROFILE = {}
        for _ in range(3):
            prof_step()
        torch.cuda.synchronize()
        prof, ops.PROFILE = ops.PROFILE, None
        model._engine.use_side_stream = side
        rec = prof.get("conv_gemm", [])
        peak = MFMA_PEAK_TFLOPS[args.dtype]
        names = SYNTH_NAMES
        # fraction of the padded rows that are valid, per sequence length (launches that pass `lens` skip fully padded tiles:
        # `achieved` counts the reference's padded algorithmic FLOPs, `achieved_valid_rows` only those of valid rows)
        vfrac = {int(b["max_mel_len"]): float(b["mel_lens"].sum()) / (args.batch * b["max_mel_len"]),
                 int(b["max_src_len"]): float(b["src_lens"].sum()) / (args.batch * b["max_src_len"])}
        by = {}
        for (f, e0, e1, var, has_lens, S_) in rec:
            d = by.setdefault(var, [0.0, 0.0, 0, 0.0])
            d[0] += f; d[1] += e0.elapsed_time(e1); d[2] += 1
            d[3] += f * (vfrac.get(int(S_), 1.0) if has_lens else 1.0)
        tot_ms = sum(d[1] for d in by.values())
        if tot_ms > 0:
            # the DOMINANT kernel = the conv_gemm variant with the largest share of the step (the 256x128 ring kernel: k=9 FFN
            # conv and k=5 PostNet conv, forward + data gradient); its rocprofv3 row is `conv_gemm_ring_kernel<...>`
            dom = max(by, key=lambda v: by[v][1])
            fl, ms, n, flv = by[dom]
            ach = fl / (ms * 1e-3) / 1e12
            fam = sum(d[0] for d in by.values()) / (tot_ms * 1e-3) / 1e12
            roofline = {"bound": "mfma", "kernel": names.get(dom, str(dom)) + " (%s)" % args.dtype,
                        "achieved": round(ach, 1), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
                        "achieved_valid_rows": round(flv / (ms * 1e-3) / 1e12, 1), "frac_valid_rows": round(flv / (ms * 1e-3) / 1e12 / peak, 4),
                        "traffic": (pmc_traffic(names.get(dom, "conv_gemm").split(" ")[0]) if (args.dtype == "bf16" and args.workload == "ljspeech" and args.dec_layers == 4
                                                                                    and not args.frame_level and args.batch == 48) else None),   # PMC passes exist for the reported config only
                        "launches_per_step": n // 3, "kernel_ms_per_step": round(ms / 3, 3), "avg_launch_us": round(ms / n * 1e3, 1),
                        "gflop_per_launch": round(fl / n / 1e9, 1),
                        "conv_gemm_family": {v2: {"kernel": names.get(v2, str(v2)), "launches_per_step": d[2] // 3,
                                                  "ms_per_step": round(d[1] / 3, 3), "tflops": round(d[0] / (d[1] * 1e-3) / 1e12, 1)}
                                             for v2, d in sorted(by.items())},
                        "family_achieved": round(fam, 1), "family_frac": round(fam / peak, 4), "family_ms_per_step": round(tot_ms / 3, 3),
                        "step_frac_of_peak": round(value / world * train_flop_per_frame(args) / (peak * 1e12), 4)}
            # what the matrix pipes SUSTAIN on this chip under its power management: a bare back-to-back bf16 MFMA stream on every
            # SIMD (fs2_mfma_calibrate, ~0.5 ms), timed right here.  `peak` stays the nominal 2.5 PF; this says how much of the
            # distance to it is the clock (MI355X_MICROARCH.md "DVFS give-back") rather than the kernel.
            try:
                import ctypes
                sink = torch.zeros(1, device=device)
                fl_ = ctypes.c_double(0.0)
                ts = []
                for _ in range(4):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    ops._lib.call("fs2_mfma_calibrate", 4000, sink.data_ptr(), ctypes.byref(fl_), ops._stream())
                    e1.record()
                    torch.cuda.synchronize()
                    ts.append(e0.elapsed_time(e1))
                sus = fl_.value / (sorted(ts[1:])[1] * 1e-3) / 1e12
                roofline["mfma_sustained"] = {"tflops": round(sus, 1), "frac_of_nominal_peak": round(sus / peak, 4),
                                              "kernel_frac_of_sustained": round(ach / sus, 4),
                                              "how": "bare v_mfma_f32_32x32x16_bf16 stream, one wave per SIMD on every CU, non-zero operands, median of 3"}
            except Exception as e:  # a measurement aid: never fails the line
                print(f"[bench] mfma calibration failed ({type(e).__name__}: {e})", file=sys.stderr)
            # ... and what THIS box's memory side streams (boxes differ by ~10 % at equal or higher MFMA rates):
            # an HBM copy (1 GiB read + 1 GiB written, beyond the 256 MB Infinity Cache) and the L2 -> LDS-DMA operand path
            try:
                import ctypes
                src = torch.empty(1 << 30, device=device, dtype=torch.uint8).fill_(3)
                dst = torch.empty_like(src)
                ts = []
                for _ in range(4):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    ops._lib.call("fs2_hbm_calibrate", src.data_ptr(), dst.data_ptr(), src.numel(), ops._stream())
                    e1.record()
                    torch.cuda.synchronize()
                    ts.append(e0.elapsed_time(e1))
                t = sorted(ts[1:])[1]
                roofline["hbm_copy"] = {"tb_per_s": round(2 * src.numel() / (t * 1e-3) / 1e12, 3), "frac_of_8tb_peak": round(2 * src.numel() / (t * 1e-3) / 8e12, 4),
                                        "how": "16-byte-per-lane copy, 1 GiB read + 1 GiB written, median of 3"}
                by = ctypes.c_double(0.0)
                ts = []
                for _ in range(4):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    ops._lib.call("fs2_ldsdma_calibrate", src.data_ptr(), src.numel(), 2000, sink.data_ptr(), ctypes.byref(by), ops._stream())
                    e1.record()
                    torch.cuda.synchronize()
                    ts.append(e0.elapsed_time(e1))
                t = sorted(ts[1:])[1]
                roofline["l2_to_lds"] = {"tb_per_s": round(by.value / (t * 1e-3) / 1e12, 2),
                                         "how": "global_load_lds_dwordx4 of a 1 MiB L2-resident window per XCD, every CU, 8 waves x 4 KiB in flight, median of 3"}
                del src, dst
            except Exception as e:
                print(f"[bench] memory calibration failed ({type(e).__name__}: {e})", file=sys.stderr)
            # the weight-gradient kernels of the same replay (side stream off: each duration is the kernel's own).  They run on
            # the side stream in the timed step and are the largest block of device time after the contractions above.
            wrec = prof.get("conv_wgrad", [])
            if wrec:
                wf, wms = sum(r[0] for r in wrec), sum(r[1].elapsed_time(r[2]) for r in wrec)
                big = [r for r in wrec if r[3] > 1]
                roofline["wgrad_family"] = {
                    "launches_per_step": len(wrec) // 3, "ms_per_step": round(wms / 3, 3), "tflops": round(wf / (wms * 1e-3) / 1e12, 1),
                    "frac": round(wf / (wms * 1e-3) / 1e12 / peak, 4),
                    "taps_ge_3": {"launches_per_step": len(big) // 3,
                                  "ms_per_step": round(sum(r[1].elapsed_time(r[2]) for r in big) / 3, 3),
                                  "tflops": round(sum(r[0] for r in big) / max(sum(r[1].elapsed_time(r[2]) for r in big) * 1e-3, 1e-9) / 1e12, 1)}}
    # secondary figure: the same step in fp32 compute (the reference's own arithmetic; exact-f32 MFMA, 157.3 TF roof)
    fp32 = {}
    if rank == 0 and world == 1 and args.dtype == "bf16" and not args.no_fp32:
        m32, l32, o32, b32, _, _ = build(args, device, rank, world, dtype="fp32")
        s32, _ = make_step(m32, l32, o32, b32, None)
        for _ in range(2):
            s32()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            s32()
        torch.cuda.synchronize()
        t32 = (time.perf_counter() - t0) / 5
        fp32 = {"fp32_ms_per_step": round(t32 * 1e3, 3), "fp32_frames_per_s": round(float(b32["mel_lens"].sum()) / t32, 1),
                "fp32_frac_of_f32_peak": round(float(b32["mel_lens"].sum()) / t32 * train_flop_per_frame(args) / (MFMA_PEAK_TFLOPS["fp32"] * 1e12), 4)}
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:      # reported at N=1 only (host cores are shared by the ranks)
        cpu = cpu_baseline(args)
    synth = None
    if rank == 0 and world == 1 and args.dtype == "bf16" and not args.no_synth:
        # the second half of BASELINE's metric rides on the driver's line: one warm-up pass + one timed pass over the
        # val.txt-shaped batches (64 steps of 8 utterances), bf16 acoustic model + bf16 vocoder
        r = synth_measure(args, device, 0, 1, 64, 64, not args.no_roofline)
        synth = {"metric": "batch-synth real-time factor (acoustic model + HiFi-GAN + int16, 22.05 kHz)",
                 "rtf": round(r["dt"] / r["audio_s"], 6), "unit": "s wall per s audio", "steps": 64, "warmup": 64,
                 "ms_per_step": round(r["dt"] / 64 * 1e3, 3), "mel_frames_per_s": round(r["frames"] / r["dt"], 1),
                 "x_realtime": round(r["audio_s"] / r["dt"], 1), "audio_s": round(r["audio_s"], 1),
                 "dtype": f"{args.dtype} acoustic / {args.vocoder_dtype} vocoder", "workload": synth_workload_text(args),
                 "loop": r["loop"], "roofline": r.get("roofline")}
        if not args.no_fp32:
            # config 5 in the REFERENCE's own arithmetic (utils/model.py:74-92 and synthesize.py run fp32): fp32 acoustic model + fp32
            # vocoder on the same 64 batches through the same loop (exact-f32 MFMA, 157.3 TF roof; the fused ResBlock stage kernel is
            # bf16-only, so this is also the un-fused pass) - the number a lower-precision RTF has to be read beside
            import copy
            a32 = copy.copy(args)
            a32.dtype, a32.vocoder_dtype = "fp32", "fp32"
            r32 = synth_measure(a32, device, 0, 1, 64, 8, False)
            synth.update({"fp32_rtf": round(r32["dt"] / r32["audio_s"], 6), "fp32_ms_per_step": round(r32["dt"] / 64 * 1e3, 3),
                          "fp32_x_realtime": round(r32["audio_s"] / r32["dt"], 1), "fp32_dtype": "fp32 acoustic / fp32 vocoder",
                          "fp32_frac_of_f32_peak": round(r32["frames"] / r32["dt"] * 614.1e6 / (MFMA_PEAK_TFLOPS["fp32"] * 1e12), 4)})
        if not args.no_cpu_baseline:
            synth["cpu_baseline"] = synth_cpu_baseline(args, r["first_batch"], cpu["cores"] if cpu else min(_thread_counts(args)[-1], 32))

    # LAST measurement (it steps the live model - 3 eager steps, a capture, steps + 3 replays - so nothing may be measured
    # after it).  The same step as ONE captured hipGraph, reported BESIDE the eager number (VERDICT r03 missing 6): ~290 dependent launches per
    # step replayed without host dispatch - but on one stream (a forked capture was slower and less accurate, capture_graph), so the
    # weight gradients no longer overlap the data-gradient chain.  `value` stays the eager step, which is what train.py runs.
    graph_ms = graph_fork_ms = None
    if graph is None and not pg and args.dtype == "bf16" and not args.no_graph_line:
        seed_was = model._engine.device_seed
        try:
            model._engine.device_seed = True
            for _ in range(3):
                step()
            torch.cuda.synchronize()
            res = []
            for fork in (False, True):          # one stream inside the capture | branches forked inside it, as the eager step runs
                g2, _sl, replay2 = capture_graph(model, opt, fwd_bwd, fork=fork)
                for _ in range(3):
                    replay2()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(args.steps):
                    replay2()
                torch.cuda.synchronize()
                res.append((time.perf_counter() - t0) / args.steps * 1e3)
                del g2
            graph_ms, graph_fork_ms = res
        except Exception as e:  # a measurement aid: never fails the line
            print(f"[bench] hipGraph side measurement failed ({type(e).__name__}: {e})", file=sys.stderr)
        model._engine.device_seed = seed_was
    if rank == 0:
        line = {
            "metric": "mel-frames/sec (train, 80-bin)", "value": round(value, 1), "unit": "mel-frames/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": ("BASELINE configs[3] shape per GPU: LibriTTS-like multi-speaker (2456 speakers) bucketed batch, "
                                    if args.workload == "libritts" else "BASELINE configs[1]: LJSpeech train, ") +
                                   f"4+{args.dec_layers} FFT layers d=256 2 heads, 80-bin mel, " + ("frame-level pitch/energy, " if args.frame_level else "") +
                                   f"batch={args.batch}/GPU, L={b['max_src_len']} phonemes, T={b['max_mel_len']} frames, "
                                   "fwd+loss+bwd+clip+Adam, dropout on, fp32 master weights",
                       "global_batch": args.batch * world, "parallelism": f"dp{world}", "hip_graph": graph is not None,
                       "valid_row_fraction": round(frames.item() / padded.item(), 4),
                       **({"sampler_group_size": args.group_size} if args.workload == "libritts" else {}),
                       **({"hip_graph_ms_per_step": round(graph_ms, 3)} if graph_ms is not None else {}),
                       **({"hip_graph_forked_ms_per_step": round(graph_fork_ms, 3)} if graph_fork_ms is not None else {}),
                       "padded_frames_per_s": round(padded.item() * args.steps / dt, 1), "final_loss": round(final_loss, 4),
                       "side_stream_wgrad": bool(args.side_stream) and graph is None, **({"host_issue_ms_per_step": round(host_ms, 3)} if host_ms else {}),
                       "windows": len(window_s), "window_ms_per_step": [round(w / args.steps * 1e3, 3) for w in window_s],
                       "library": lib_used, "dev_env": [], "hw_queues": dict(fastspeech2_amd.HW_QUEUES), **fp32, **dist_info, **({"synth": synth} if synth else {})},
            "roofline": roofline, "cpu_baseline": cpu,
        }
        print(json.dumps(line), flush=True)
    if dist.is_initialized():
        dist.barrier()                     # ranks leave together (rank 0 was still replaying for the roofline)
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
