"""Data formats on the input side of the hot path: the reference's `Dataset` / `TextDataset`
surface (dataset.py:12-198) plus what an MI355X needs so the loader does not starve a ~17 ms train step:

* `Dataset`, `TextDataset` — same constructors, same sample dicts, same `collate_fn` result (a list of the 12-tuples /
  one 6-tuple that `to_device` and `model(*batch[2:])` consume), same metadata line format `name|speaker|{phones}|raw`.
* `FeaturePack` — the reference opens four `.npy` files per utterance per epoch (dataset.py:39-62).  `pack_features`
  concatenates every utterance's mel / pitch / energy / duration into four flat arrays + an offset table in ONE `.npz`
  (memory-mappable), `Dataset` uses it when present: one sequential read instead of 4·N small opens.
* `BucketedBatchSampler` — length-sorted batches dealt round-robin to the ranks (one process per GPU), so all ranks
  run similar T in the same step and padding waste stays low; deterministic per (seed, epoch), no communication.
* `DevicePrefetcher` — a host thread collates batch i+1.. into PINNED staging buffers and issues the H2D copies on a
  side HIP stream while step i runs; the consumer gets device tensors plus an event to wait on.

Everything here is host-side numpy/python; the only torch use is pinned memory, streams and the H2D copy.
"""
import json
import os
import queue
import threading

import numpy as np
import torch

from .text import text_to_sequence
from .utils import pad_1D, pad_2D

_KINDS = ("mel", "pitch", "energy", "duration")


def read_metadata(path):
    """`name|speaker|{phones}|raw text` lines -> four parallel lists (dataset.py:81-96,176-188)."""
    name, speaker, text, raw = [], [], [], []
    with open(path, "r", encoding="utf-8") as f:
        for line in f.readlines():
            n, s, t, r = line.strip("\n").split("|")
            name.append(n); speaker.append(s); text.append(t); raw.append(r)
    return name, speaker, text, raw


# ---------------------------------------------------------------------------------------------------- packed features
class FeaturePack:
    """Flat feature arrays + per-utterance offsets.  File layout (`{preprocessed_path}/{filename}.pack.npz`):
    keys[N] (str "speaker-basename"), mel [sum T, n_mel] f32, pitch / energy [sum Lp] f32, duration [sum L] i64,
    off_{kind} [N+1] i64."""

    def __init__(self, path):
        z = np.load(path, allow_pickle=False)
        self.index = {k: i for i, k in enumerate(z["keys"].tolist())}
        self.arr = {k: z[k] for k in _KINDS}
        self.off = {k: z["off_" + k] for k in _KINDS}

    def get(self, key):
        i = self.index[key]
        return {k: self.arr[k][self.off[k][i]:self.off[k][i + 1]] for k in _KINDS}

    @staticmethod
    def path_for(preprocessed_path, filename):
        return os.path.join(preprocessed_path, filename + ".pack.npz")


def pack_features(preprocessed_path, filename):
    """Build the packed feature file for one metadata list from the per-utterance `.npy` files."""
    names, speakers, _, _ = read_metadata(os.path.join(preprocessed_path, filename))
    cols = {k: [] for k in _KINDS}
    keys = []
    for n, s in zip(names, speakers):
        keys.append(f"{s}-{n}")
        for k in _KINDS:
            cols[k].append(np.load(os.path.join(preprocessed_path, k, f"{s}-{k}-{n}.npy")))
    out = {"keys": np.array(keys)}
    for k in _KINDS:
        lens = np.array([0] + [len(a) for a in cols[k]], dtype=np.int64)
        out["off_" + k] = np.cumsum(lens)
        out[k] = np.concatenate(cols[k], axis=0) if cols[k] else np.zeros((0,))
    path = FeaturePack.path_for(preprocessed_path, filename)
    np.savez(path, **out)
    return path


# ---------------------------------------------------------------------------------------------------- datasets
class Dataset(torch.utils.data.Dataset):
    """reference dataset.py:12-146."""

    def __init__(self, filename, preprocess_config, train_config, sort=False, drop_last=False):
        self.dataset_name = preprocess_config["dataset"]
        self.preprocessed_path = preprocess_config["path"]["preprocessed_path"]
        self.cleaners = preprocess_config["preprocessing"]["text"]["text_cleaners"]
        self.batch_size = train_config["optimizer"]["batch_size"]
        self.basename, self.speaker, self.text, self.raw_text = read_metadata(os.path.join(self.preprocessed_path, filename))
        with open(os.path.join(self.preprocessed_path, "speakers.json")) as f:
            self.speaker_map = json.load(f)
        self.sort = sort
        self.drop_last = drop_last
        pack = FeaturePack.path_for(self.preprocessed_path, filename)
        self.pack = FeaturePack(pack) if os.path.exists(pack) else None
        self._phones = [None] * len(self.text)          # id sequences are cached: the string parse is the per-item CPU cost

    def __len__(self):
        return len(self.text)

    def phones(self, idx):
        if self._phones[idx] is None:
            self._phones[idx] = np.array(text_to_sequence(self.text[idx], self.cleaners))
        return self._phones[idx]

    def _features(self, speaker, basename):
        if self.pack is not None:
            return self.pack.get(f"{speaker}-{basename}")
        return {k: np.load(os.path.join(self.preprocessed_path, k, f"{speaker}-{k}-{basename}.npy")) for k in _KINDS}

    def __getitem__(self, idx):
        basename, speaker = self.basename[idx], self.speaker[idx]
        f = self._features(speaker, basename)
        return {"id": basename, "speaker": self.speaker_map[speaker], "text": self.phones(idx), "raw_text": self.raw_text[idx],
                "mel": f["mel"], "pitch": f["pitch"], "energy": f["energy"], "duration": f["duration"]}

    def length(self, idx):
        """phoneme count of item idx without touching the feature files (used by the bucketed sampler)."""
        return len(self.phones(idx))

    @staticmethod
    def reprocess(data, idxs):
        """dataset.py:98-125 -> the train 12-tuple."""
        pick = lambda k: [data[i][k] for i in idxs]      # noqa: E731
        texts, mels = pick("text"), pick("mel")
        text_lens = np.array([t.shape[0] for t in texts])
        mel_lens = np.array([m.shape[0] for m in mels])
        return (pick("id"), pick("raw_text"), np.array(pick("speaker")), pad_1D(texts), text_lens, max(text_lens),
                pad_2D(mels), mel_lens, max(mel_lens), pad_1D(pick("pitch")), pad_1D(pick("energy")), pad_1D(pick("duration")))

    def collate_fn(self, data):
        """dataset.py:127-146: (optionally length-sorted) group of samples -> list of batch tuples of `batch_size`."""
        n = len(data)
        idx = np.argsort(-np.array([d["text"].shape[0] for d in data])) if self.sort else np.arange(n)
        rem = n % self.batch_size
        tail, idx = idx[n - rem:], idx[:n - rem]
        groups = idx.reshape((-1, self.batch_size)).tolist()
        if not self.drop_last and len(tail) > 0:
            groups += [tail.tolist()]
        return [self.reprocess(data, g) for g in groups]


class TextDataset(torch.utils.data.Dataset):
    """reference dataset.py:149-198 (batch synthesis input)."""

    def __init__(self, filepath, preprocess_config):
        self.cleaners = preprocess_config["preprocessing"]["text"]["text_cleaners"]
        self.basename, self.speaker, self.text, self.raw_text = read_metadata(filepath)
        with open(os.path.join(preprocess_config["path"]["preprocessed_path"], "speakers.json")) as f:
            self.speaker_map = json.load(f)

    def __len__(self):
        return len(self.text)

    def __getitem__(self, idx):
        phone = np.array(text_to_sequence(self.text[idx], self.cleaners))
        return (self.basename[idx], self.speaker_map[self.speaker[idx]], phone, self.raw_text[idx])

    def collate_fn(self, data):
        texts = [d[2] for d in data]
        text_lens = np.array([t.shape[0] for t in texts])
        return [d[0] for d in data], [d[3] for d in data], np.array([d[1] for d in data]), pad_1D(texts), text_lens, max(text_lens)


# ---------------------------------------------------------------------------------------------------- sampling
class BucketedBatchSampler:
    """Per-rank list of index lists for one epoch.  The global index list is shuffled (seed + epoch), cut into
    windows of `window` = group_size * world * batch_size items (the reference sorts inside windows of 4 batches,
    train.py:30-37), each window is sorted by length, cut into steps of world*batch_size items, and each step is
    dealt to the ranks card-wise: every rank sees the same length profile in the same step.  Incomplete last windows are dropped (drop_last)."""

    def __init__(self, lengths, batch_size, world_size=1, rank=0, group_size=4, shuffle=True, seed=1234):
        self.lengths = np.asarray(lengths)
        self.batch_size, self.world, self.rank = batch_size, world_size, rank
        self.group_size, self.shuffle, self.seed = group_size, shuffle, seed
        self.epoch = 0

    def set_epoch(self, epoch):
        self.epoch = epoch

    def __len__(self):
        per_step = self.world * self.batch_size
        return len(self.lengths) // per_step

    def __iter__(self):
        n = len(self.lengths)
        order = np.random.default_rng(self.seed + self.epoch).permutation(n) if self.shuffle else np.arange(n)
        per_step = self.world * self.batch_size
        window = self.group_size * per_step
        for w0 in range(0, n - per_step + 1, window):
            w = order[w0:w0 + window]
            w = w[: len(w) // per_step * per_step]
            w = w[np.argsort(-self.lengths[w], kind="stable")]
            for s0 in range(0, len(w), per_step):
                # strided deal: rank r takes items r, r+world, ... of the step's sorted slice -> every rank sees the same
                # length profile (and nearly the same longest item, which is what sets its padded T)
                yield w[s0 + self.rank: s0 + per_step: self.world].tolist()


# ---------------------------------------------------------------------------------------------------- H2D prefetch
_TRAIN_DTYPES = {2: np.int64, 3: np.int64, 4: np.int64, 6: np.float32, 7: np.int64, 9: np.float32, 10: np.float32, 11: np.int64}
_SYNTH_DTYPES = {2: np.int64, 3: np.int64, 4: np.int64}


class DevicePrefetcher:
    """Iterate device-resident batch tuples.  `batches` yields numpy batch tuples (12 = train, 6 = synth, the
    `to_device` convention of utils/tools.py:18-66).  A worker thread stages each array in pinned host memory (a ring
    of `depth` slots, re-used) and enqueues the copies on `copy_stream`; `__next__` makes the CURRENT stream wait for
    that batch's copy event, so step i's kernels overlap batch i+1's collate + PCIe transfer."""

    def __init__(self, batches, device, depth=3):
        self.it = iter(batches)
        self.device = torch.device(device)
        self.depth = depth
        self.q = queue.Queue(maxsize=depth)
        self.copy_stream = torch.cuda.Stream(device=self.device) if self.device.type == "cuda" else None
        self._pinned = [dict() for _ in range(depth + 2)]
        self._slot_ev = [None] * (depth + 2)            # last copy event of each staging slot (waited for before reuse)
        self._slot = 0
        self._err = None
        self.thread = threading.Thread(target=self._work, daemon=True)
        self.thread.start()

    def _stage(self, slot, key, arr):
        """numpy -> (pinned) host tensor, reusing the slot's buffer when it is large enough."""
        arr = np.ascontiguousarray(arr)
        if self.copy_stream is None:
            return torch.from_numpy(arr)
        buf = self._pinned[slot].get(key)
        if buf is None or buf.numel() < arr.size or buf.dtype != torch.from_numpy(arr[:0]).dtype:
            buf = torch.empty(max(arr.size, 1), dtype=torch.from_numpy(arr[:0]).dtype).pin_memory()
            self._pinned[slot][key] = buf
        view = buf[:arr.size].view(arr.shape)
        view.numpy()[...] = arr
        return view

    def _work(self):
        try:
            for b in self.it:
                dts = _TRAIN_DTYPES if len(b) == 12 else _SYNTH_DTYPES if len(b) == 6 else None
                if dts is None:
                    raise ValueError(f"DevicePrefetcher: batch of length {len(b)} (expected 12 or 6)")
                slot = self._slot
                self._slot = (self._slot + 1) % len(self._pinned)
                if self._slot_ev[slot] is not None:
                    self._slot_ev[slot].synchronize()   # the DMA that last read this slot's pinned buffers has finished
                out = list(b)
                ev = None
                if self.copy_stream is not None:
                    with torch.cuda.stream(self.copy_stream):
                        for i, dt_ in dts.items():
                            out[i] = self._stage(slot, i, np.asarray(b[i]).astype(dt_, copy=False)).to(self.device, non_blocking=True)
                        ev = torch.cuda.Event()
                        ev.record(self.copy_stream)
                        self._slot_ev[slot] = ev
                else:
                    for i, dt_ in dts.items():
                        out[i] = self._stage(slot, i, np.asarray(b[i]).astype(dt_, copy=False))
                for i in ((4, 7) if len(b) == 12 else (4,)):    # lengths vectors carry their host copy (utils.lens_to_device's contract:
                    out[i]._fs2_host = np.asarray(b[i])            # the engine reads the padding fraction from it without a device round trip)
                self.q.put((tuple(out), ev))
        except BaseException as e:  # surfaced on the consumer side
            self._err = e
        self.q.put(None)

    def __iter__(self):
        return self

    def __next__(self):
        item = self.q.get()
        if item is None:
            if self._err is not None:
                raise self._err
            raise StopIteration
        batch, ev = item
        if ev is not None:
            cur = torch.cuda.current_stream(self.device)
            cur.wait_event(ev)
            # the tensors were allocated on the copy stream: tell the caching allocator that the CONSUMER stream uses them, so
            # that their blocks are not handed back to the copy stream (and overwritten by the next batch's H2D) while step
            # i's loss / embedding-gradient kernels are still queued (no host sync separates steps: gradient accumulation,
            # graph replay)
            for t in batch:
                if isinstance(t, torch.Tensor) and t.is_cuda:
                    t.record_stream(cur)
        return batch


def train_batches(dataset, sampler):
    """numpy train 12-tuples for one epoch of `sampler` (each yielded index list is one batch)."""
    for idxs in sampler:
        data = [dataset[i] for i in idxs]
        yield Dataset.reprocess(data, list(range(len(data))))
