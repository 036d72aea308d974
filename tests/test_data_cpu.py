"""Data formats on the input side of the hot path (CPU): symbol inventory / phoneme ids, Dataset / TextDataset collation
against golden vectors produced by the live reference (tests/golden/make_golden_data.py), packed feature files,
length-bucketed sharding and the prefetcher's host path."""
import json
import os

import numpy as np
import pytest

from tests.golden import configs
from tests.helpers import make_preprocessed_dir

HERE = os.path.dirname(os.path.abspath(__file__))


def test_symbols_and_sequences_match_reference():
    from fastspeech2_amd import text
    from fastspeech2_amd.model import N_SYMBOLS
    g = json.load(open(os.path.join(HERE, "golden", "symbols.json"), encoding="utf-8"))
    assert text.symbols == g["symbols"]
    assert len(text.symbols) == N_SYMBOLS
    for c in g["cases"]:
        assert text.text_to_sequence(c["phones"], ["english_cleaners"]) == c["ids"], c["dataset"]
    ids = g["cases"][0]["ids"]
    assert text.text_to_sequence(text.sequence_to_text(ids)) == ids


def _cfg(d, batch=4):
    pcfg, _ = configs.make()
    pcfg["path"]["preprocessed_path"] = d
    return pcfg, {"optimizer": {"batch_size": batch}}


def test_collate_matches_reference(tmp_path):
    from fastspeech2_amd.data import Dataset, TextDataset
    d = make_preprocessed_dir(str(tmp_path), seed=77, n_train=11, n_val=5)
    pcfg, tcfg = _cfg(d)
    g = np.load(os.path.join(HERE, "golden", "collate.npz"), allow_pickle=False)
    ds = Dataset("train.txt", pcfg, tcfg, sort=True, drop_last=True)
    batches = ds.collate_fn([ds[i] for i in range(len(ds))])
    assert len(batches) == int(g["n_batches"]) == 2
    names = ("speakers", "texts", "text_lens", "max_text_len", "mels", "mel_lens", "max_mel_len", "pitches", "energies", "durations")
    for bi, b in enumerate(batches):
        assert len(b) == 12
        assert list(b[0]) == [str(s) for s in g[f"b{bi}_ids"]]
        for j, n in zip(range(2, 12), names):
            ref = g[f"b{bi}_{n}"]
            got = np.asarray(b[j])
            assert got.shape == ref.shape and np.array_equal(got, ref), (bi, n)
            if got.ndim:
                assert got.dtype == ref.dtype, (bi, n, got.dtype, ref.dtype)
    # not dropping the tail keeps the 3 left-over utterances as a short batch (dataset.py:137-143)
    ds2 = Dataset("train.txt", pcfg, tcfg, sort=True, drop_last=False)
    assert [len(b[0]) for b in ds2.collate_fn([ds2[i] for i in range(len(ds2))])] == [4, 4, 3]
    tds = TextDataset(os.path.join(d, "val.txt"), pcfg)
    tb = tds.collate_fn([tds[i] for i in range(len(tds))])
    assert list(tb[0]) == [str(s) for s in g["t_ids"]]
    assert np.array_equal(tb[2], g["t_speakers"]) and np.array_equal(tb[3], g["t_texts"]) and np.array_equal(tb[4], g["t_lens"])
    assert int(tb[5]) == int(g["t_max"])


def test_feature_pack_equals_npy_files(tmp_path):
    from fastspeech2_amd.data import Dataset, FeaturePack, pack_features
    d = make_preprocessed_dir(str(tmp_path), seed=5, n_train=9, n_val=3, speakers=("a", "b"))
    pcfg, tcfg = _cfg(d)
    plain = Dataset("train.txt", pcfg, tcfg)
    items = [plain[i] for i in range(len(plain))]
    path = pack_features(d, "train.txt")
    assert path == FeaturePack.path_for(d, "train.txt") and os.path.exists(path)
    packed = Dataset("train.txt", pcfg, tcfg)
    assert packed.pack is not None
    for a, b in zip(items, (packed[i] for i in range(len(packed)))):
        assert a["id"] == b["id"] and a["speaker"] == b["speaker"]
        for k in ("text", "mel", "pitch", "energy", "duration"):
            assert a[k].dtype == b[k].dtype and np.array_equal(a[k], b[k]), k


def test_bucketed_sampler_partitions_and_balances():
    from fastspeech2_amd.data import BucketedBatchSampler
    rng = np.random.default_rng(0)
    lengths = rng.integers(20, 200, 1000)
    world, bs = 4, 8
    per_rank = [list(BucketedBatchSampler(lengths, bs, world, r, group_size=4, seed=3)) for r in range(world)]
    n_steps = len(per_rank[0])
    assert n_steps == 1000 // (world * bs) == len(BucketedBatchSampler(lengths, bs, world, 0))
    assert all(len(p) == n_steps for p in per_rank)
    seen = [i for p in per_rank for b in p for i in b]
    assert len(seen) == len(set(seen)) == n_steps * world * bs           # disjoint across ranks and steps
    # same step -> similar lengths on every rank (spread across ranks much smaller than the global spread)
    spread = [max(np.mean(lengths[per_rank[r][s]]) for r in range(world)) - min(np.mean(lengths[per_rank[r][s]]) for r in range(world))
              for s in range(n_steps)]
    assert np.mean(spread) < 0.35 * lengths.std()
    # deterministic per (seed, epoch); a new epoch reshuffles
    a = list(BucketedBatchSampler(lengths, bs, world, 1, seed=3))
    s2 = BucketedBatchSampler(lengths, bs, world, 1, seed=3); s2.set_epoch(1)
    assert a == per_rank[1] and list(s2) != a


def test_prefetcher_host_path(tmp_path):
    """On a CPU device the prefetcher is a plain background collate thread: order, dtypes and values are preserved."""
    import torch
    from fastspeech2_amd.data import BucketedBatchSampler, Dataset, DevicePrefetcher, train_batches
    d = make_preprocessed_dir(str(tmp_path), seed=9, n_train=16, n_val=2)
    pcfg, tcfg = _cfg(d)
    ds = Dataset("train.txt", pcfg, tcfg)
    lengths = [ds.length(i) for i in range(len(ds))]
    sampler = BucketedBatchSampler(lengths, 4, shuffle=False)
    ref = list(train_batches(ds, sampler))
    got = list(DevicePrefetcher(train_batches(ds, sampler), "cpu", depth=2))
    assert len(got) == len(ref) == 4
    for r, g in zip(ref, got):
        assert g[0] == r[0] and g[5] == r[5] and g[8] == r[8]
        for j in (2, 3, 4, 6, 7, 9, 10, 11):
            assert isinstance(g[j], torch.Tensor)
            assert np.array_equal(g[j].numpy(), np.asarray(r[j]).astype(g[j].numpy().dtype))
        assert g[3].dtype == torch.int64 and g[6].dtype == torch.float32 and g[11].dtype == torch.int64
        # the two lengths vectors carry their host copy (what Engine._lens_pays reads instead of a device round trip)
        assert np.array_equal(g[4]._fs2_host, np.asarray(r[4])) and np.array_equal(g[7]._fs2_host, np.asarray(r[7]))
    with pytest.raises(ValueError):
        list(DevicePrefetcher(iter([(1, 2, 3)]), "cpu"))
