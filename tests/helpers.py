"""Shared helpers for parity tests: build seeded weights / batches and run the CPU oracle."""
import numpy as np
import torch

from oracle import fs2_oracle as O
from oracle.weights import seeded_state_dict, synthetic_batch
from tests.golden import configs


def load_golden(tag):
    import os
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", tag + ".npz"), allow_pickle=False)


def make_model(pcfg, mcfg, compute_dtype="fp32"):
    from fastspeech2_amd.model import FastSpeech2
    return FastSpeech2(pcfg, mcfg, compute_dtype=compute_dtype)


def oracle_train_case(pcfg, mcfg, sd, b, dtype=torch.float32):
    """train-mode (dropout neutralised) forward + loss + backward through the oracle; returns outputs, losses, grads."""
    sdr = {k: (v.to(dtype) if v.is_floating_point() else v).clone() for k, v in sd.items()}
    leaves = {}
    for k, v in sdr.items():
        if v.is_floating_point() and not any(s in k for s in ("position_enc", "_bins", "running_")):
            v.requires_grad_(True)
            leaves[k] = v
    bn_buffers = {k: v.clone() for k, v in sdr.items() if "running_" in k}
    out = O.fastspeech2_forward(sdr, mcfg, pcfg, b["speakers"], b["texts"], b["src_lens"], b["max_src_len"],
                                b["mels"].to(dtype), b["mel_lens"], b["max_mel_len"], b["pitches"].to(dtype),
                                b["energies"].to(dtype), b["durations"], training=True, dropout=False, bn_buffers=bn_buffers)
    losses = O.fastspeech2_loss(pcfg, (b["mels"].to(dtype), b["pitches"].to(dtype), b["energies"].to(dtype), b["durations"]), out)
    losses[0].backward()
    grads = {k: v.grad for k, v in leaves.items() if v.grad is not None}
    return out, losses, grads, bn_buffers


def oracle_gathered_case(pcfg, mcfg, sd, batches, dtype=torch.float64):
    """What the reference's nn.DataParallel step computes (train.py:42, 82-86): every replica runs the module on ITS chunk of the
    batch (train-mode BatchNorm statistics are therefore per replica), the outputs are gathered, ONE loss is taken over the gathered
    batch (masked means over ALL valid positions), and backward sums the replicas' parameter gradients.  `batches` = one oracle
    batch dict per replica; outputs are padded to the longest replica before the concatenation (masked positions never enter the
    loss).  Returns (losses, grads of the gathered-batch loss)."""
    sdr = {k: (v.to(dtype) if v.is_floating_point() else v).clone() for k, v in sd.items()}
    leaves = {}
    for k, v in sdr.items():
        if v.is_floating_point() and not any(s in k for s in ("position_enc", "_bins", "running_")):
            v.requires_grad_(True)
            leaves[k] = v
    outs = []
    for b in batches:
        bn_buffers = {k: v.clone() for k, v in sdr.items() if "running_" in k}
        outs.append(O.fastspeech2_forward(sdr, mcfg, pcfg, b["speakers"], b["texts"], b["src_lens"], b["max_src_len"],
                                          b["mels"].to(dtype), b["mel_lens"], b["max_mel_len"], b["pitches"].to(dtype),
                                          b["energies"].to(dtype), b["durations"], training=True, dropout=False, bn_buffers=bn_buffers))
    L = max(o[6].shape[1] for o in outs)
    T = max(o[7].shape[1] for o in outs)

    def pad(x, n, value=0):
        """pad dim 1 to n"""
        if x.shape[1] == n:
            return x
        shp = list(x.shape)
        shp[1] = n - x.shape[1]
        return torch.cat([x, torch.full(shp, value, dtype=x.dtype)], 1)

    frame_p = pcfg["preprocessing"]["pitch"]["feature"] != "phoneme_level"
    frame_e = pcfg["preprocessing"]["energy"]["feature"] != "phoneme_level"
    gathered = (torch.cat([pad(o[0], T) for o in outs]), torch.cat([pad(o[1], T) for o in outs]),
                torch.cat([pad(o[2], T if frame_p else L) for o in outs]), torch.cat([pad(o[3], T if frame_e else L) for o in outs]),
                torch.cat([pad(o[4], L) for o in outs]), None,
                torch.cat([pad(o[6], L, True) for o in outs]), torch.cat([pad(o[7], T, True) for o in outs]), None, None)
    targets = (torch.cat([pad(b["mels"].to(dtype), T) for b in batches]),
               torch.cat([pad(b["pitches"].to(dtype), T if frame_p else L) for b in batches]),
               torch.cat([pad(b["energies"].to(dtype), T if frame_e else L) for b in batches]),
               torch.cat([pad(b["durations"], L) for b in batches]))
    losses = O.fastspeech2_loss(pcfg, targets, gathered)
    losses[0].backward()
    return losses, {k: v.grad for k, v in leaves.items() if v.grad is not None}


def bf16_matrix(name, v):
    """the parameters the bf16 engine holds a bf16 copy of and multiplies in bf16: the weights of its MFMA contractions (Linear /
    Conv1d of the FFT blocks, variance predictors' convs, mel_linear, PostNet convs: Engine._conv_list).  Embedding tables, the
    variance predictors' 256 -> 1 output rows, biases and normalisation parameters are read in fp32."""
    if not (v.is_floating_point() and v.dim() >= 2):
        return False
    if any(s in name for s in ("position_enc", "emb", "linear_layer")):
        return False
    return True


def grad_stats(g):
    g = g.double()
    return np.array([g.sum().item(), g.abs().sum().item(), g.norm().item()])


_PHONES = ("AA1 AE0 AH0 AO1 B CH D DH EH1 ER0 F G HH IH1 IY0 JH K L M N NG OW1 P R S SH T TH UW1 V W Y Z sp".split())


def make_preprocessed_dir(root, seed=77, n_train=11, n_val=5, n_mel=80, speakers=("LJSpeech",), lo=5, hi=24):
    """A tiny synthetic dataset in the reference's on-disk format (preprocessor/preprocessor.py:120-161 writes exactly
    these files): {train,val}.txt lines `name|speaker|{phones}|raw`, speakers.json, stats.json and per-utterance
    mel/pitch/energy/duration .npy files (phoneme-level pitch/energy).  Deterministic in (seed, sizes)."""
    import json
    import os

    rng = np.random.default_rng(seed)
    for k in ("mel", "pitch", "energy", "duration"):
        os.makedirs(os.path.join(root, k), exist_ok=True)
    with open(os.path.join(root, "speakers.json"), "w") as f:
        json.dump({s: i for i, s in enumerate(speakers)}, f)
    with open(os.path.join(root, "stats.json"), "w") as f:
        json.dump(configs.LJ_STATS, f)
    for fname, n in (("train.txt", n_train), ("val.txt", n_val)):
        lines = []
        for i in range(n):
            name = f"{fname[:2]}{i:03d}"
            spk = speakers[int(rng.integers(len(speakers)))]
            L = int(rng.integers(lo, hi + 1))
            ph = [_PHONES[int(j)] for j in rng.integers(0, len(_PHONES), L)]
            dur = rng.integers(1, 7, L).astype(np.int64)
            T = int(dur.sum())
            np.save(os.path.join(root, "mel", f"{spk}-mel-{name}.npy"), rng.normal(-5, 2, (T, n_mel)).astype(np.float32))
            np.save(os.path.join(root, "pitch", f"{spk}-pitch-{name}.npy"), rng.normal(0, 1, L))
            np.save(os.path.join(root, "energy", f"{spk}-energy-{name}.npy"), rng.normal(0, 1, L).astype(np.float32))
            np.save(os.path.join(root, "duration", f"{spk}-duration-{name}.npy"), dur)
            lines.append(f"{name}|{spk}|{{{' '.join(ph)}}}|utterance number {i}")
        with open(os.path.join(root, fname), "w", encoding="utf-8") as f:
            f.write("\n".join(lines) + "\n")
    return root


# ---------------------------------------------------------------------------------------------- raw corpus (preprocessor tests)
def fake_pitch(wav, sampling_rate, hop_length):
    """Deterministic stand-in for pyworld dio + stonemask (absent from this image): one value per frame (len // hop + 1 frames,
    as DIO yields), 0 = unvoiced where the frame's RMS is below 0.01, else a smooth contour.  Used identically by the golden
    generator (stubbed into the reference as `pyworld`) and by the tests (passed as `pitch_fn`)."""
    wav = np.asarray(wav, dtype=np.float64)
    n = len(wav) // hop_length + 1
    pad = np.concatenate([wav, np.zeros(n * hop_length - len(wav))])
    rms = np.sqrt((pad.reshape(n, hop_length) ** 2).mean(axis=1))
    i = np.arange(n)
    f0 = 140.0 + 45.0 * np.sin(i / 5.0) + 20.0 * np.cos(i / 1.7) + 300.0 * rms
    return np.where(rms >= 0.01, f0, 0.0)


def make_raw_corpus(root, seed=5, sr=22050, hop=256):
    """A tiny raw corpus + MFA-style alignments in the layout preprocessor/preprocessor.py:66-79,155-160 walks:
    `{root}/raw/{speaker}/{name}.wav|.lab`, `{root}/pre/TextGrid/{speaker}/{name}.TextGrid`.
    Cases: leading / trailing / inner silences, a zero-frame phone, a wav without TextGrid (stale-statistics quirk),
    an all-silence alignment (start >= end -> skipped), a near-silent recording (<= 1 voiced frame -> skipped), a short-format
    TextGrid with an empty interval.  Returns (preprocess config dict, {(speaker, name): [(start, end, phone), ...]})."""
    import os
    from scipy.io import wavfile

    rng = np.random.RandomState(seed)
    raw, pre = os.path.join(root, "raw"), os.path.join(root, "pre")
    plan = {
        ("spkA", "a1"): ["sil", "DH", "AH0", "sp", "K", "AE1", "T", "sp"],
        ("spkA", "a2"): ["HH", "AH0", "L", "OW1", "sp", "W", "ER1", "L", "D", "sil"],
        ("spkA", "a3"): None,                                   # wav + lab but no TextGrid
        ("spkA", "a4"): ["sil", "sp", "sil"],                   # nothing but silence
        ("spkB", "b1"): ["sp", "S", "IH1", "K", "S", "spn", "T", "IY1", "N"],
        ("spkB", "b2"): ["T", "EH1", "S", "T"],                 # near-silent audio: no voiced frames
        ("spkB", "b3"): ["sil", "M", "AO1", "R", "", "F", "AY1", "V", "sil"],    # short format, "" interval dropped by the reader
    }
    tables = {}
    for (spk, name), phones in plan.items():
        os.makedirs(os.path.join(raw, spk), exist_ok=True)
        os.makedirs(os.path.join(pre, "TextGrid", spk), exist_ok=True)
        n_ph = len(phones) if phones else 6
        lens = rng.uniform(0.03, 0.22, size=n_ph)
        if name == "a2":
            lens[2] = 0.0009                                    # rounds to a zero-frame phone
        bounds = np.round(np.concatenate([[0.0], np.cumsum(lens)]), 4)
        total = float(bounds[-1]) + 0.05
        t = np.arange(int(total * sr)) / sr
        wav = 0.001 * rng.randn(len(t))
        for k in range(n_ph):
            p = phones[k] if phones else "AH0"
            if p in ("sil", "sp", "spn", ""):
                continue
            m = (t >= bounds[k]) & (t < bounds[k + 1])
            f = rng.uniform(100, 320)
            amp = 0.002 if name == "b2" else rng.uniform(0.15, 0.45)
            wav[m] += amp * (np.sin(2 * np.pi * f * t[m]) + 0.4 * np.sin(2 * np.pi * 3.1 * f * t[m]) + 0.1 * rng.randn(m.sum()))
        wavfile.write(os.path.join(raw, spk, name + ".wav"), sr, (np.clip(wav, -1, 1) * 32767).astype(np.int16))
        with open(os.path.join(raw, spk, name + ".lab"), "w") as f:
            f.write("raw text of %s\n" % name)
        if phones is None:
            continue
        iv = [(float(bounds[k]), float(bounds[k + 1]), phones[k]) for k in range(n_ph)]
        tables[(spk, name)] = [x for x in iv if x[2] != ""]
        words = [(0.0, float(bounds[-1]), "w")]
        path = os.path.join(pre, "TextGrid", spk, name + ".TextGrid")
        with open(path, "w") as f:
            if name == "b3":                                    # Praat "short" text format
                f.write('File type = "ooTextFile"\nObject class = "TextGrid"\n\n0\n%r\n<exists>\n2\n' % float(bounds[-1]))
                for tier, items in (("words", words), ("phones", iv)):
                    f.write('"IntervalTier"\n"%s"\n0\n%r\n%d\n' % (tier, float(bounds[-1]), len(items)))
                    for s, e, p in items:
                        f.write('%r\n%r\n"%s"\n' % (s, e, p))
            else:
                f.write('File type = "ooTextFile"\nObject class = "TextGrid"\n\nxmin = 0 \nxmax = %r \ntiers? <exists> \nsize = 2 \nitem []: \n'
                        % float(bounds[-1]))
                for ti, (tier, items) in enumerate((("words", words), ("phones", iv))):
                    f.write('    item [%d]:\n        class = "IntervalTier" \n        name = "%s" \n        xmin = 0 \n        xmax = %r \n'
                            '        intervals: size = %d \n' % (ti + 1, tier, float(bounds[-1]), len(items)))
                    for k, (s, e, p) in enumerate(items):
                        f.write('        intervals [%d]:\n            xmin = %r \n            xmax = %r \n            text = "%s" \n'
                                % (k + 1, s, e, p))
    cfg = {"dataset": "Synth", "path": {"raw_path": raw, "preprocessed_path": pre},
           "preprocessing": {"val_size": 2, "text": {"text_cleaners": ["english_cleaners"], "language": "en"},
                             "audio": {"sampling_rate": sr, "max_wav_value": 32768.0},
                             "stft": {"filter_length": 1024, "hop_length": hop, "win_length": 1024},
                             "mel": {"n_mel_channels": 80, "mel_fmin": 0, "mel_fmax": 8000},
                             "pitch": {"feature": "phoneme_level", "normalization": True},
                             "energy": {"feature": "phoneme_level", "normalization": True}}}
    return cfg, tables


# ---------------------------------------------------------------------------------------------- checkpoints (resume tests)
CKPT_SEED, CKPT_B, CKPT_L = 606, 3, 14
CKPT_CFG = dict(dec_layers=2, enc_layers=2)


def oracle_written_checkpoint(path, steps=2):
    """The file the reference's train loop writes after `steps` optimiser steps (train.py:82-97,152-161), rebuilt WITHOUT the
    reference: oracle forward + FastSpeech2Loss restatement, torch.nn.utils.clip_grad_norm_, torch.optim.Adam over the parameters
    in the reference's parameters() order (pinned by tests/golden/state_schema.json) with the reference's LR schedule
    (model/optimizer.py:33-51).  tests/test_checkpoint_cpu.py proves it equal to the file the live reference writes;
    tests/golden/ckpt_resume.npz carries the reference-written file's per-tensor checksums for boxes without the reference.
    Returns (pcfg, mcfg, batch, ckpt dict)."""
    pcfg, mcfg = configs.make(dropout=False, **CKPT_CFG)
    from fastspeech2_amd.model import FastSpeech2
    template = FastSpeech2(pcfg, mcfg)
    names = [n for n, _ in template.named_parameters()]
    seeded = seeded_state_dict(template.state_dict(), CKPT_SEED)
    sd = {k: seeded[k].clone() for k in template.state_dict()}          # state_dict() order, as torch.save sees it
    params = []
    for n, p in template.named_parameters():
        sd[n].requires_grad_(p.requires_grad)
        params.append(sd[n])
    tc = configs.TRAIN["optimizer"]
    opt = torch.optim.Adam(params, betas=tc["betas"], eps=tc["eps"], weight_decay=tc["weight_decay"])
    init_lr = np.power(mcfg["transformer"]["encoder_hidden"], -0.5)
    b = synthetic_batch(CKPT_SEED + 1, CKPT_B, CKPT_L)
    bn = {k: v for k, v in sd.items() if "running_" in k}
    for step in range(1, steps + 1):
        out = O.fastspeech2_forward(sd, mcfg, pcfg, b["speakers"], b["texts"], b["src_lens"], b["max_src_len"], b["mels"], b["mel_lens"],
                                    b["max_mel_len"], b["pitches"], b["energies"], b["durations"], training=True, dropout=False,
                                    bn_buffers=bn)
        loss = O.fastspeech2_loss(pcfg, (b["mels"], b["pitches"], b["energies"], b["durations"]), out)[0]
        loss.backward()
        torch.nn.utils.clip_grad_norm_(params, tc["grad_clip_thresh"])
        lr = init_lr * np.min([np.power(step, -0.5), np.power(tc["warm_up_step"], -1.5) * step])
        for g in opt.param_groups:
            g["lr"] = lr                                     # a numpy scalar, as in the reference (optimizer.py:19,50)
        opt.step()
        opt.zero_grad()
        for k in sd:
            if k.endswith("num_batches_tracked"):
                sd[k] += 1                                   # nn.BatchNorm1d counts training forwards
    ckpt = {"model": {k: v.detach().clone() for k, v in sd.items()}, "optimizer": opt.state_dict()}
    torch.save(ckpt, path)
    return pcfg, mcfg, b, ckpt, names
