"""Corpus preprocessor host logic (fastspeech2_amd/preprocess.py) against the reference's OWN preprocessor run on the same
synthetic corpus (tests/golden/preproc_*.npz, made by tests/golden/make_golden_preproc.py).  The STFT itself is the GPU
test's business (tests/test_preprocess_gpu.py); here `_extract_mels` is overridden with the CPU oracle so that everything
around it — TextGrid reader, alignment, trimming, phoneme averaging, outlier removal, running statistics incl. the
stale-value quirk, normalisation, file formats, metadata — is pinned without a GPU."""
import json
import os

import numpy as np
import pytest
import torch

from fastspeech2_amd import preprocess as P
from oracle import fs2_oracle as O
from tests.helpers import fake_pitch, load_golden, make_raw_corpus


class OraclePreprocessor(P.Preprocessor):
    def _extract_mels(self, wavs):
        out = []
        for w in wavs:
            y = torch.clip(torch.from_numpy(np.asarray(w, dtype=np.float32)).unsqueeze(0), -1, 1)
            mel, energy = O.mel_spectrogram(y)
            out.append((mel[0].numpy().astype(np.float32), energy[0].numpy().astype(np.float32)))
        return out


def _run(tmp_path, monkeypatch, cls, frame_level, **kw):
    cfg, tables = make_raw_corpus(str(tmp_path))
    if frame_level:
        for k in ("pitch", "energy"):
            cfg["preprocessing"][k]["feature"] = "frame_level"
            cfg["preprocessing"][k]["normalization"] = False
    real = os.listdir
    monkeypatch.setattr(os, "listdir", lambda p: sorted(real(p)))
    out = cls(cfg, pitch_fn=fake_pitch, seed=3, **kw).build_from_path()
    return cfg, tables, out


def check_against_golden(cfg, out, tag, mel_atol):
    g = load_golden("preproc_" + tag)
    meta = json.loads(str(g["meta"]))
    pre = cfg["path"]["preprocessed_path"]
    assert sorted(out) == meta["lines"]
    assert json.load(open(os.path.join(pre, "speakers.json"))) == meta["speakers"]
    n_train = len(open(os.path.join(pre, "train.txt")).read().splitlines())
    n_val = len(open(os.path.join(pre, "val.txt")).read().splitlines())
    assert (n_train, n_val) == (meta["n_train"], meta["n_val"])
    stats = json.load(open(os.path.join(pre, "stats.json")))
    for k in ("pitch", "energy"):
        np.testing.assert_allclose(stats[k], meta["stats"][k], rtol=2e-5, atol=2e-5)
    names = [k for k in g.files if k != "meta"]
    for kind in ("mel", "pitch", "energy", "duration"):
        assert sorted(os.listdir(os.path.join(pre, kind))) == sorted(k.split("/")[1] + ".npy" for k in names if k.startswith(kind + "/"))
    for k in names:
        got = np.load(os.path.join(pre, k + ".npy"))
        ref = g[k]
        assert got.shape == ref.shape and got.dtype == ref.dtype, (k, got.shape, ref.shape, got.dtype, ref.dtype)
        if k.startswith("duration/"):
            assert np.array_equal(got, ref), k                               # integer frame counts: bit-exact
        elif k.startswith("mel/"):
            assert np.abs(got - ref).max() <= mel_atol, (k, np.abs(got - ref).max())
        else:
            np.testing.assert_allclose(got, ref, rtol=1e-4, atol=1e-4, err_msg=k)


@pytest.mark.parametrize("tag", ["phoneme", "frame"])
def test_preprocessor_matches_reference_golden(tmp_path, monkeypatch, tag):
    cfg, _, out = _run(tmp_path, monkeypatch, OraclePreprocessor, tag == "frame", device="cpu", num_workers=2)
    check_against_golden(cfg, out, tag, mel_atol=2e-4)


def test_corpus_streams_through_windows(tmp_path, monkeypatch):
    """host memory holds one window of waveforms: with a tiny window the corpus takes several flushes and the outputs do not change."""
    flushes = []

    class Windowed(OraclePreprocessor):
        def _extract_mels(self, wavs):
            flushes.append(len(wavs))
            return super()._extract_mels(wavs)

    cfg, tables = make_raw_corpus(str(tmp_path))
    real = os.listdir
    monkeypatch.setattr(os, "listdir", lambda p: sorted(real(p)))
    pp = Windowed(cfg, pitch_fn=fake_pitch, seed=3, device="cpu", num_workers=2, batch_seconds=0.2)
    pp.host_chunk = 2
    out = pp.build_from_path()
    assert sum(flushes) == 4 and len(flushes) >= 3, flushes
    check_against_golden(cfg, out, "phoneme", mel_atol=2e-4)


def test_textgrid_reader_long_and_short_formats(tmp_path):
    cfg, tables = make_raw_corpus(str(tmp_path))
    pre = cfg["path"]["preprocessed_path"]
    for (spk, name), iv in tables.items():
        tiers = P.read_textgrid(os.path.join(pre, "TextGrid", spk, name + ".TextGrid"))
        assert set(tiers) == {"words", "phones"}
        assert tiers["phones"] == iv, (spk, name)                           # exact floats, exact order, "" intervals dropped
        assert len(tiers["words"]) == 1
    full = P.read_textgrid(os.path.join(pre, "TextGrid", "spkB", "b3.TextGrid"), include_empty_intervals=True)
    assert len(full["phones"]) == len(tables[("spkB", "b3")]) + 1
    bad = tmp_path / "bad.TextGrid"
    bad.write_text("not a textgrid\n")
    with pytest.raises(ValueError):
        P.read_textgrid(str(bad))


def test_get_alignment_rules(tmp_path):
    cfg, _ = make_raw_corpus(str(tmp_path))
    pp = OraclePreprocessor(cfg, device="cpu", pitch_fn=fake_pitch)
    f = 256 / 22050
    iv = [(0.0, 3 * f, "sil"), (3 * f, 5.4 * f, "AH0"), (5.4 * f, 5.6 * f, "sp"), (5.6 * f, 9 * f, "T"), (9 * f, 12 * f, "sp"),
          (12 * f, 13 * f, "sil")]
    phones, dur, s, e = pp.get_alignment(iv)
    assert phones == ["AH0", "sp", "T"] and dur == [2, 1, 3] and s == 3 * f and e == 9 * f   # rounded-boundary differences telescope
    assert pp.get_alignment([(0.0, 1.0, "sil"), (1.0, 2.0, "sp")]) == ([], [], 0, 0)
    assert pp.get_alignment([]) == ([], [], 0, 0)


def test_running_moments_equals_sklearn_partial_fit():
    from sklearn.preprocessing import StandardScaler
    rng = np.random.default_rng(0)
    sk, rm = StandardScaler(), P.RunningMoments()
    for n in (1, 7, 300, 2, 64):
        x = rng.normal(3.0, 40.0, size=n)
        sk.partial_fit(x.reshape(-1, 1))
        rm.partial_fit(x)
        assert abs(rm.mean - sk.mean_[0]) <= 1e-12 * abs(sk.mean_[0]) + 1e-12
        assert abs(rm.scale - sk.scale_[0]) <= 1e-12 * sk.scale_[0] + 1e-12
    rm.partial_fit(np.array([]))
    assert rm.n == 374
    one = P.RunningMoments().partial_fit([5.0, 5.0])
    assert one.scale == 1.0                                                  # zero variance -> scale 1, as sklearn


def test_phoneme_average_and_outliers():
    v = np.arange(10, dtype=np.float32)
    assert np.allclose(P.phoneme_average(v, [2, 0, 3, 5]), [0.5, 0.0, 3.0, 7.0])
    # zero-length phones first: the reference averages IN PLACE, so later means see the zeros it already wrote
    assert np.allclose(P.phoneme_average(np.array([4.0, 6.0, 8.0]), [0, 0, 3]), [0.0, 0.0, (0 + 0 + 8.0) / 3])
    x = np.array([1.0, 1.1, 0.9, 1.05, 50.0, -40.0, 1.0])
    assert set(P.remove_outlier(x)) == {1.0, 1.1, 0.9, 1.05}


def test_missing_pitch_backend_and_cpu_device_fail_loudly(tmp_path, monkeypatch):
    cfg, _ = make_raw_corpus(str(tmp_path))
    monkeypatch.setattr(P, "_pyworld_pitch", lambda: None)
    with pytest.raises(RuntimeError, match="pyworld"):
        P.Preprocessor(cfg, device="cpu").build_from_path()
    real = os.listdir
    monkeypatch.setattr(os, "listdir", lambda p: sorted(real(p)))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        P.Preprocessor(cfg, device="cpu", pitch_fn=fake_pitch, num_workers=1).build_from_path()


def test_load_wav_formats(tmp_path):
    from scipy.io import wavfile
    t = np.arange(4410) / 44100.0
    stereo = np.stack([np.sin(2 * np.pi * 200 * t), np.sin(2 * np.pi * 200 * t)], axis=1)
    wavfile.write(str(tmp_path / "s.wav"), 44100, (stereo * 20000).astype(np.int16))
    w = P.load_wav(str(tmp_path / "s.wav"))
    assert w.dtype == np.float32 and w.ndim == 1 and abs(len(w) - 2205) <= 1 and abs(np.abs(w).max() - 20000 / 32768) < 0.02
    wavfile.write(str(tmp_path / "f.wav"), 22050, np.linspace(-0.5, 0.5, 100).astype(np.float32))
    assert np.allclose(P.load_wav(str(tmp_path / "f.wav")), np.linspace(-0.5, 0.5, 100), atol=1e-7)
