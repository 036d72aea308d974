"""Corpus feature extraction around the HIP STFT (reference preprocessor/preprocessor.py:15-310, preprocess.py).

`Preprocessor(config).build_from_path()` reads `{raw_path}/{speaker}/{basename}.wav|.lab` and the MFA alignments under
`{preprocessed_path}/TextGrid/{speaker}/{basename}.TextGrid`, and writes what `dataset.py` reads:
`{mel,pitch,energy,duration}/{speaker}-{kind}-{basename}.npy`, `stats.json`, `speakers.json`, `train.txt`, `val.txt`.

The reference runs one utterance at a time through a CPU conv1d STFT (`preprocessor.py:194`).  Here the corpus is
processed in two stages so that the GPU sees few, large launches:

  host stage    per utterance: TextGrid -> phones / frame durations / trim window (`get_alignment`), wav read + trim,
                F0 (pyworld DIO + StoneMask when importable, or a caller-supplied `pitch_fn`); thread pool.
  device stage  utterances are packed, longest first, into ragged batches of up to `batch_seconds` of audio
                (one pinned staging buffer -> one H2D copy -> reflect pad per row -> framed-DFT GEMM -> fused
                |.| / mel / log / energy epilogue -> one D2H copy), `audio.TacotronSTFT.mel_spectrogram_ragged`.

Everything after the device stage (trim to sum(duration), phoneme-level averaging, outlier removal for the statistics,
normalisation, file formats, metadata lines, the train/val split) follows the reference line by line, including
its quirk that a wav WITHOUT a TextGrid re-feeds the previous utterance's pitch/energy values into the running
statistics (`preprocessor.py:70-89`: `pitch`, `energy`, `n` keep their last values).

Third-party pieces of the reference that are absent from this image and what stands in for them:
  tgt==1.4.4 `io.read_textgrid`  -> `read_textgrid` below (long and short Praat text formats; empty-text intervals
                                    are dropped, as tgt does by default with include_empty_intervals=False)
  librosa==0.7.2 `load`          -> `load_wav`: scipy.io.wavfile, mono, float32 in [-1, 1); polyphase resampling to
                                    22050 Hz only if the file's rate differs (librosa.load's default sr; the files
                                    `prepare_align.py` writes are already at that rate)
  pyworld==0.2.10 dio/stonemask  -> imported when present; otherwise `pitch_fn` must be given (no silent substitute)
  sklearn StandardScaler         -> `RunningMoments` (the same incremental mean/variance update, Chan et al.)
"""
import json
import os
import random
import re
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

from . import audio as Audio

SIL_PHONES = ("sil", "sp", "spn")


# ------------------------------------------------------------------------------------------------ TextGrid
_TOKEN = re.compile(r'"((?:[^"]|"")*)"|(-?\d+(?:\.\d+)?(?:[eE][-+]?\d+)?)|(<exists>|<absent>)')


def read_textgrid(path, include_empty_intervals=False):
    """Praat TextGrid (long "key = value" or short value-only text format) -> {tier name: [(start, end, text), ...]}.
    Point tiers yield (time, time, mark).  Replaces tgt.io.read_textgrid for `preprocessor.py:162-165`."""
    raw = open(path, "rb").read()
    for enc in ("utf-8-sig", "utf-16"):
        try:
            src = raw.decode(enc)
            break
        except UnicodeError:
            continue
    else:
        raise ValueError(f"{path}: not a UTF-8/UTF-16 TextGrid")
    lines = []
    for line in src.splitlines():
        s = line.strip()
        if s.startswith("!"):                                   # Praat comment line
            continue
        # long format: drop everything up to '=' (key names, "[n]" indices) unless the '=' sits inside a quoted string
        if "=" in s:
            q = s.find('"')
            e = s.find("=")
            if q < 0 or e < q:
                s = s[e + 1:]
        elif re.match(r"^(item|intervals|points)\s*\[\d*\]\s*:?$", s) or s in ("item []:", "item []"):
            continue
        lines.append(s)
    toks = []
    for m in _TOKEN.finditer("\n".join(lines)):
        if m.group(1) is not None:
            toks.append(m.group(1).replace('""', '"'))
        elif m.group(2) is not None:
            toks.append(float(m.group(2)))
        else:
            toks.append(m.group(3) == "<exists>")
    if len(toks) < 6 or toks[0] != "ooTextFile" or toks[1] != "TextGrid":
        raise ValueError(f"{path}: not a Praat TextGrid text file")
    pos = 4                                                     # file type, object class, xmin, xmax
    if toks[pos] is not True:
        return {}
    n_tiers = int(toks[pos + 1])
    pos += 2
    tiers = {}
    for _ in range(n_tiers):
        cls, name = toks[pos], toks[pos + 1]
        n = int(toks[pos + 4])
        pos += 5
        items = []
        if cls == "IntervalTier":
            for _ in range(n):
                s, e, text = float(toks[pos]), float(toks[pos + 1]), toks[pos + 2]
                pos += 3
                if include_empty_intervals or text.strip() != "":
                    items.append((s, e, text))
        elif cls == "TextTier":
            for _ in range(n):
                t, text = float(toks[pos]), toks[pos + 1]
                pos += 2
                items.append((t, t, text))
        else:
            raise ValueError(f"{path}: unknown tier class {cls!r}")
        tiers.setdefault(name, items)
    return tiers


# ------------------------------------------------------------------------------------------------ small host pieces
def load_wav(path, target_sr=22050):
    """float32 mono waveform at `target_sr` (what `librosa.load(path)` returns for the reference, preprocessor.py:172)."""
    from scipy.io import wavfile

    sr, w = wavfile.read(path)
    if w.dtype.kind == "i":
        w = w.astype(np.float32) / float(1 << (8 * w.dtype.itemsize - 1))
    elif w.dtype.kind == "u":                                   # 8-bit PCM
        w = (w.astype(np.float32) - 128.0) / 128.0
    else:
        w = w.astype(np.float32)
    if w.ndim == 2:
        w = w.mean(axis=1)
    if sr != target_sr:
        from math import gcd
        from scipy.signal import resample_poly
        g = gcd(int(sr), int(target_sr))
        w = resample_poly(w, target_sr // g, sr // g).astype(np.float32)
    return w


class RunningMoments:
    """Incremental mean / population variance (sklearn StandardScaler.partial_fit's update, preprocessor.py:61-62,86-89)."""

    def __init__(self):
        self.n, self.mean, self.m2 = 0, 0.0, 0.0

    def partial_fit(self, x):
        x = np.asarray(x, dtype=np.float64).reshape(-1)
        if x.size == 0:
            return self
        new_sum = x.sum()
        n_new, n_old = x.size, self.n
        n_tot = n_old + n_new
        new_m2 = x.var() * n_new
        if n_old == 0:
            m2 = new_m2
        else:
            last_sum = self.mean * n_old
            ratio = n_old / n_new
            m2 = self.m2 + new_m2 + ratio / n_tot * (last_sum / ratio - new_sum) ** 2
        self.mean = (self.mean * n_old + new_sum) / n_tot
        self.m2, self.n = m2, n_tot
        return self

    @property
    def scale(self):
        s = float(np.sqrt(self.m2 / max(self.n, 1)))
        return s if s != 0.0 else 1.0                           # sklearn's _handle_zeros_in_scale


def remove_outlier(values):
    """preprocessor.py:283-291: Tukey fence - keep what lies strictly inside (Q1 - 1.5 IQR, Q3 + 1.5 IQR)."""
    v = np.asarray(values)
    q1, q3 = np.percentile(v, [25, 75])
    fence = 1.5 * (q3 - q1)
    return v[(v > q1 - fence) & (v < q3 + fence)]


def phoneme_average(values, durations):
    """preprocessor.py:210-218,222-229: mean over each phoneme's frames (0 for zero-length phonemes); returns len(durations)
    values.  Slices past the end of `values` behave as numpy's do (mean of an empty slice is NaN), like the reference."""
    out = np.array(values, copy=True)
    pos = 0
    for i, d in enumerate(durations):
        out[i] = np.mean(out[pos:pos + d]) if d > 0 else 0
        pos += d
    return out[:len(durations)]


def _pyworld_pitch():
    try:
        import pyworld as pw
    except ImportError:
        return None

    def pitch_fn(wav, sampling_rate, hop_length):
        x = wav.astype(np.float64)
        f0, t = pw.dio(x, sampling_rate, frame_period=hop_length / sampling_rate * 1000)
        return pw.stonemask(x, f0, t, sampling_rate)
    return pitch_fn


# ------------------------------------------------------------------------------------------------ the preprocessor
class Preprocessor:
    def __init__(self, config, device="cuda", pitch_fn=None, batch_seconds=1800.0, num_workers=8, seed=None):
        """`config` = preprocess.yaml (preprocessor.py:16-51).  `pitch_fn(wav float32, sampling_rate, hop_length) -> f0 per
        frame (0 = unvoiced)` replaces pyworld when that package is absent; `seed` fixes the train/val shuffle
        (the reference uses the unseeded global `random`)."""
        self.config = config
        self.in_dir = config["path"]["raw_path"]
        self.out_dir = config["path"]["preprocessed_path"]
        self.val_size = config["preprocessing"]["val_size"]
        self.sampling_rate = config["preprocessing"]["audio"]["sampling_rate"]
        self.hop_length = config["preprocessing"]["stft"]["hop_length"]
        assert config["preprocessing"]["pitch"]["feature"] in ["phoneme_level", "frame_level"]
        assert config["preprocessing"]["energy"]["feature"] in ["phoneme_level", "frame_level"]
        self.pitch_phoneme_averaging = config["preprocessing"]["pitch"]["feature"] == "phoneme_level"
        self.energy_phoneme_averaging = config["preprocessing"]["energy"]["feature"] == "phoneme_level"
        self.pitch_normalization = config["preprocessing"]["pitch"]["normalization"]
        self.energy_normalization = config["preprocessing"]["energy"]["normalization"]
        self.STFT = Audio.TacotronSTFT(
            config["preprocessing"]["stft"]["filter_length"], config["preprocessing"]["stft"]["hop_length"],
            config["preprocessing"]["stft"]["win_length"], config["preprocessing"]["mel"]["n_mel_channels"],
            config["preprocessing"]["audio"]["sampling_rate"], config["preprocessing"]["mel"]["mel_fmin"],
            config["preprocessing"]["mel"]["mel_fmax"])
        self.device = torch.device(device)
        self.pitch_fn = pitch_fn
        self.batch_samples = int(batch_seconds * self.sampling_rate)
        self.num_workers = num_workers
        self.host_chunk = max(64, 8 * num_workers)          # utterances handed to the host thread pool at a time
        self.seed = seed
        self._staging = None

    # ---------------------------------------------------------------- host stage
    def get_alignment(self, intervals):
        """preprocessor.py:243-281 on [(start, end, phone), ...]: trim leading / trailing silences, durations in frames as the
        difference of the ROUNDED boundary positions (so they telescope), trim window in seconds."""
        phones, durations = [], []
        start_time = end_time = 0
        end_idx = 0
        for s, e, p in intervals:
            if phones == []:
                if p in SIL_PHONES:
                    continue
                start_time = s
            phones.append(p)
            if p not in SIL_PHONES:
                end_time = e
                end_idx = len(phones)
            durations.append(int(np.round(e * self.sampling_rate / self.hop_length)
                                 - np.round(s * self.sampling_rate / self.hop_length)))
        return phones[:end_idx], durations[:end_idx], start_time, end_time

    def _tg_path(self, speaker, basename):
        return os.path.join(self.out_dir, "TextGrid", speaker, "{}.TextGrid".format(basename))

    def _host_stage(self, speaker, basename):
        """preprocessor.py:154-191 up to (not including) the STFT.  Returns None where the reference returns None."""
        tiers = read_textgrid(self._tg_path(speaker, basename))
        if "phones" not in tiers:
            raise KeyError("{}: no tier named 'phones'".format(self._tg_path(speaker, basename)))
        phone, duration, start, end = self.get_alignment(tiers["phones"])
        if start >= end:
            return None
        wav = load_wav(os.path.join(self.in_dir, speaker, "{}.wav".format(basename)))
        wav = wav[int(self.sampling_rate * start):int(self.sampling_rate * end)].astype(np.float32)
        with open(os.path.join(self.in_dir, speaker, "{}.lab".format(basename)), "r") as f:
            raw_text = f.readline().strip("\n")
        pitch = np.asarray(self.pitch_fn(wav, self.sampling_rate, self.hop_length), dtype=np.float64)[:sum(duration)]
        if np.sum(pitch != 0) <= 1:
            return None
        return {"speaker": speaker, "basename": basename, "text": "{" + " ".join(phone) + "}", "raw_text": raw_text,
                "duration": duration, "wav": wav, "pitch": pitch}

    # ---------------------------------------------------------------- device stage
    def _batches(self, items):
        """Longest first, then greedy packing under `batch_samples` of PADDED audio (rows x longest row)."""
        order = sorted(range(len(items)), key=lambda i: -len(items[i]["wav"]))
        batch, longest = [], 0
        for i in order:
            n = len(items[i]["wav"])
            if batch and (len(batch) + 1) * max(longest, n) > self.batch_samples:
                yield batch
                batch, longest = [], 0
            batch.append(i)
            longest = max(longest, n)
        if batch:
            yield batch

    def _extract_mels(self, wavs):
        """[float32 1-D] -> [(mel (n_mel, frames), energy (frames,))] float32 numpy; one ragged launch set on the GPU
        (audio/tools.py:8-15 `get_mel_from_wav` per utterance in the reference: clip to [-1, 1], STFT, squeeze)."""
        if self.device.type != "cuda":
            raise RuntimeError("fastspeech2_amd.preprocess runs its STFT on an AMD GPU only (no CPU fallback)")
        lens = [len(w) for w in wavs]
        B, N = len(wavs), max(lens)
        if self._staging is None or self._staging.numel() < B * N:
            self._staging = torch.empty(B * N, dtype=torch.float32).pin_memory()
        host = self._staging[:B * N].view(B, N)
        hv = host.numpy()
        for b, w in enumerate(wavs):
            np.clip(w, -1.0, 1.0, out=hv[b, :lens[b]])
            hv[b, lens[b]:] = 0.0
        y = host.to(self.device, non_blocking=True)
        mel, energy, frames = self.STFT.mel_spectrogram_ragged(y, torch.tensor(lens, dtype=torch.int32))
        mel, energy = mel.cpu().numpy(), energy.cpu().numpy()               # one D2H each; the copy orders after the kernels
        return [(mel[b, :, :f].astype(np.float32), energy[b, :f].astype(np.float32))
                for b, f in enumerate(frames.tolist())]

    # ---------------------------------------------------------------- per utterance, after the STFT
    def _finish_utterance(self, it, mel_spectrogram, energy):
        """preprocessor.py:194-241."""
        duration, pitch = it["duration"], it["pitch"]
        speaker, basename = it["speaker"], it["basename"]
        mel_spectrogram = mel_spectrogram[:, :sum(duration)]
        energy = energy[:sum(duration)]
        if self.pitch_phoneme_averaging:
            from scipy.interpolate import interp1d
            nonzero_ids = np.where(pitch != 0)[0]
            interp_fn = interp1d(nonzero_ids, pitch[nonzero_ids],
                                 fill_value=(pitch[nonzero_ids[0]], pitch[nonzero_ids[-1]]), bounds_error=False)
            pitch = phoneme_average(interp_fn(np.arange(0, len(pitch))), duration)
        if self.energy_phoneme_averaging:
            energy = phoneme_average(energy, duration)
        np.save(os.path.join(self.out_dir, "duration", "{}-duration-{}.npy".format(speaker, basename)), duration)
        np.save(os.path.join(self.out_dir, "pitch", "{}-pitch-{}.npy".format(speaker, basename)), pitch)
        np.save(os.path.join(self.out_dir, "energy", "{}-energy-{}.npy".format(speaker, basename)), energy)
        np.save(os.path.join(self.out_dir, "mel", "{}-mel-{}.npy".format(speaker, basename)), mel_spectrogram.T)
        return ("|".join([basename, speaker, it["text"], it["raw_text"]]), remove_outlier(pitch), remove_outlier(energy),
                mel_spectrogram.shape[1])

    def normalize(self, in_dir, mean, std):
        """preprocessor.py:293-305: standardise every .npy of `in_dir` in place; returns the (min, max) over the whole corpus
        (float64 extremes when the directory is empty, as the reference's running min / max start there)."""
        lo, hi = np.finfo(np.float64).max, np.finfo(np.float64).min
        for name in os.listdir(in_dir):
            path = os.path.join(in_dir, name)
            z = (np.load(path) - mean) / std
            np.save(path, z)
            if z.size:
                lo, hi = min(lo, z.min()), max(hi, z.max())
        return lo, hi

    # ---------------------------------------------------------------- the corpus pass
    def build_from_path(self):
        if self.pitch_fn is None:
            self.pitch_fn = _pyworld_pitch()
        if self.pitch_fn is None:
            raise RuntimeError("pyworld is not installed: pass pitch_fn=(wav, sampling_rate, hop_length) -> f0 per frame")
        for d in ("mel", "pitch", "energy", "duration"):
            os.makedirs(os.path.join(self.out_dir, d), exist_ok=True)
        print("Processing Data ...")

        # corpus walk in the reference's order (preprocessor.py:66-73); entries without a TextGrid are kept as markers
        speakers, entries = {}, []
        for i, speaker in enumerate(os.listdir(self.in_dir)):
            speakers[speaker] = i
            for wav_name in os.listdir(os.path.join(self.in_dir, speaker)):
                if ".wav" not in wav_name:
                    continue
                basename = wav_name.split(".")[0]
                entries.append((speaker, basename, os.path.exists(self._tg_path(speaker, basename))))

        # the corpus streams through in windows of a few device batches of audio, so host memory holds one window of waveforms
        # (not the corpus: LibriTTS is ~185 GB of float32 samples); per-utterance results kept for the statistics are tiny
        todo = [k for k, e in enumerate(entries) if e[2]]
        results = {}                                                        # entry index -> _finish_utterance tuple, or None
        window, window_samples = [], 0

        def flush():
            items = [it for _, it in window]
            for batch in self._batches(items):
                for i, (mel, energy) in zip(batch, self._extract_mels([items[i]["wav"] for i in batch])):
                    items[i]["wav"] = None                                  # release the audio once its features exist
                    results[window[i][0]] = self._finish_utterance(items[i], mel, energy)
            window.clear()

        with ThreadPoolExecutor(max_workers=max(1, self.num_workers)) as pool:
            chunk = self.host_chunk
            for c0 in range(0, len(todo), chunk):
                ks = todo[c0:c0 + chunk]
                for k, it in zip(ks, pool.map(lambda k: self._host_stage(entries[k][0], entries[k][1]), ks)):
                    if it is None:
                        results[k] = None
                        continue
                    window.append((k, it))
                    window_samples += len(it["wav"])
                if window_samples >= 4 * self.batch_samples:
                    flush()
                    window_samples = 0
            flush()

        # running statistics in corpus order, with the reference's stale-value behaviour for wavs that have no TextGrid
        out, n_frames = [], 0
        pitch_scaler, energy_scaler = RunningMoments(), RunningMoments()
        last = None
        for k, (speaker, basename, has_tg) in enumerate(entries):
            if has_tg:
                if results[k] is None:
                    continue                                                # preprocessor.py:80-81
                last = results[k]
                out.append(last[0])
            if last is None:
                raise NameError("first wav of the corpus has no TextGrid (the reference fails here too: "
                                "preprocessor.py:86 reads `pitch` before assignment)")
            _, pitch, energy, n = last
            if len(pitch) > 0:
                pitch_scaler.partial_fit(pitch)
            if len(energy) > 0:
                energy_scaler.partial_fit(energy)
            n_frames += n

        print("Computing statistic quantities ...")
        pitch_mean, pitch_std = (pitch_scaler.mean, pitch_scaler.scale) if self.pitch_normalization else (0, 1)
        energy_mean, energy_std = (energy_scaler.mean, energy_scaler.scale) if self.energy_normalization else (0, 1)
        pitch_min, pitch_max = self.normalize(os.path.join(self.out_dir, "pitch"), pitch_mean, pitch_std)
        energy_min, energy_max = self.normalize(os.path.join(self.out_dir, "energy"), energy_mean, energy_std)

        with open(os.path.join(self.out_dir, "speakers.json"), "w") as f:
            f.write(json.dumps(speakers))
        with open(os.path.join(self.out_dir, "stats.json"), "w") as f:
            f.write(json.dumps({"pitch": [float(pitch_min), float(pitch_max), float(pitch_mean), float(pitch_std)],
                                "energy": [float(energy_min), float(energy_max), float(energy_mean), float(energy_std)]}))
        print("Total time: {} hours".format(n_frames * self.hop_length / self.sampling_rate / 3600))

        (random.Random(self.seed) if self.seed is not None else random).shuffle(out)
        out = [r for r in out if r is not None]
        with open(os.path.join(self.out_dir, "train.txt"), "w", encoding="utf-8") as f:
            for m in out[self.val_size:]:
                f.write(m + "\n")
        with open(os.path.join(self.out_dir, "val.txt"), "w", encoding="utf-8") as f:
            for m in out[:self.val_size]:
                f.write(m + "\n")
        return out
