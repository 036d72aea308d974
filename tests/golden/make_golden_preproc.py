"""Golden outputs of the reference's OWN preprocessor (preprocessor/preprocessor.py) on the tiny synthetic corpus of
tests/helpers.make_raw_corpus.  Run in the build container (needs /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_preproc.py

Absent third-party packages are stubbed, each as thinly as possible, so that every line of preprocessor.py itself runs:
  tgt      io.read_textgrid returns tier objects built from the interval tables the corpus generator returns (NOT from our
           TextGrid reader — that one is tested against the same tables separately)
  librosa  load = scipy wavfile -> float32 / 32768 (the corpus is already 22050 Hz mono); util / filters as in make_golden.py
           (filters.mel serves the oracle's restated Slaney filterbank: "parity unpinned" at that one boundary)
  pyworld  dio / stonemask = tests.helpers.fake_pitch (the tests pass the same function as `pitch_fn`)
os.listdir is made sorted (directory order is filesystem-dependent and decides speaker ids and the stale-statistics quirk).
Two configurations: phoneme-level + normalisation (stock), frame-level without normalisation.
"""
import copy
import json
import os
import sys
import tempfile
import types

import numpy as np
import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)

from oracle import fs2_oracle as O  # noqa: E402
from tests.helpers import fake_pitch, make_raw_corpus  # noqa: E402

TABLES = {}


class _Iv:
    def __init__(self, s, e, p):
        self.start_time, self.end_time, self.text = s, e, p


class _Tier:
    def __init__(self, items):
        self._objects = [_Iv(*x) for x in items]


class _TG:
    def __init__(self, items):
        self.items = items

    def get_tier_by_name(self, name):
        assert name == "phones"
        return _Tier(self.items)


def _read_textgrid(path):
    spk, name = path.split(os.sep)[-2], os.path.basename(path)[:-len(".TextGrid")]
    return _TG(TABLES[(spk, name)])


def install_stubs():
    from scipy.io import wavfile
    tgt = types.ModuleType("tgt")
    tgt.io = types.ModuleType("tgt.io")
    tgt.io.read_textgrid = _read_textgrid
    lib = types.ModuleType("librosa")
    util, filt = types.ModuleType("librosa.util"), types.ModuleType("librosa.filters")

    def pad_center(data, size, axis=-1):
        n = data.shape[axis]
        lpad = (size - n) // 2
        return np.pad(data, (lpad, size - n - lpad))

    def load(path):
        sr, w = wavfile.read(path)
        assert sr == 22050 and w.dtype == np.int16
        return w.astype(np.float32) / 32768.0, sr
    util.pad_center, util.tiny, util.normalize = pad_center, (lambda x: np.finfo(np.float32).tiny), (lambda x, norm=None: x)
    filt.mel = lambda sr, n_fft, n_mels, fmin, fmax: O.slaney_mel_filterbank(sr, n_fft, n_mels, fmin, fmax)
    lib.util, lib.filters, lib.load = util, filt, load
    pw = types.ModuleType("pyworld")
    state = {}

    def dio(x, fs, frame_period):
        hop = int(round(frame_period * fs / 1000))
        f0 = fake_pitch(x, fs, hop)
        state["f0"] = f0
        return f0, np.arange(len(f0)) * frame_period / 1000
    pw.dio = dio
    pw.stonemask = lambda x, f0, t, fs: f0
    sys.modules.update({"tgt": tgt, "tgt.io": tgt.io, "librosa": lib, "librosa.util": util, "librosa.filters": filt, "pyworld": pw})
    for name in ("unidecode", "inflect"):
        sys.modules.setdefault(name, types.ModuleType(name))


def run(tag, mutate):
    with tempfile.TemporaryDirectory() as tmp:
        cfg, tables = make_raw_corpus(tmp)
        TABLES.clear()
        TABLES.update(tables)
        cfg = copy.deepcopy(cfg)
        mutate(cfg)
        from preprocessor.preprocessor import Preprocessor
        import random
        random.seed(3)
        out = Preprocessor(cfg).build_from_path()
        pre = cfg["path"]["preprocessed_path"]
        arrays = {}
        for kind in ("mel", "pitch", "energy", "duration"):
            for fn in sorted(os.listdir(os.path.join(pre, kind))):
                arrays[kind + "/" + fn[:-4]] = np.load(os.path.join(pre, kind, fn))
        meta = {"stats": json.load(open(os.path.join(pre, "stats.json"))),
                "speakers": json.load(open(os.path.join(pre, "speakers.json"))),
                "lines": sorted(out),
                "n_train": len(open(os.path.join(pre, "train.txt")).read().splitlines()),
                "n_val": len(open(os.path.join(pre, "val.txt")).read().splitlines())}
        np.savez_compressed(os.path.join(HERE, f"preproc_{tag}.npz"), meta=json.dumps(meta), **arrays)
        print(tag, meta["stats"], meta["lines"], {k: v.shape for k, v in arrays.items() if k.startswith("mel/")})


if __name__ == "__main__":
    os.chdir(REF)
    install_stubs()
    real_listdir = os.listdir
    os.listdir = lambda p: sorted(real_listdir(p))
    real_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self            # audio/stft.py:68-69 call .cuda() unconditionally
    try:
        run("phoneme", lambda c: None)

        def frame_level(c):
            for k in ("pitch", "energy"):
                c["preprocessing"][k]["feature"] = "frame_level"
                c["preprocessing"][k]["normalization"] = False
        run("frame", frame_level)
    finally:
        torch.Tensor.cuda = real_cuda
        os.listdir = real_listdir
