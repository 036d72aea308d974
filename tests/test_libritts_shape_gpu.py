"""BASELINE configs[3] at ITS OWN shape, against the oracle: the batch `bench.py --workload libritts` measures (2456-way speaker
embedding, LibriTTS-like phoneme counts dealt by the real BucketedBatchSampler, B = 48, L ~ 250, T ~ 990, ~46 % valid rows).

Why a file of its own: the FFT-block contractions run in two modes (engine.py `_lens_pays`): lens-free (< 10 %
of the 256-row tiles wholly padded - every golden and both LJSpeech-shaped full-size fixtures) and lens + tile map (buckets like
this one: padded tiles are skipped, padded rows of the others zeroed).  The tests here
  * assert that the engine really CHOOSES the lens + tile-map mode on this batch (and the lens-free mode on the LJSpeech batch),
  * compare the fp32 eval forward of the whole batch with the oracle (north_star bar: mel L1 < 1e-4, lengths / masks bit-equal),
  * compare one fp32 train step's gradients, tensor by tensor, with the fp64 oracle - on a B = 16 slice of the bucket (every third
    utterance: same length profile, same skippable-tile fraction; the fp64 oracle of all 48 would take minutes), in the mode the
    engine chooses and with the mode forced both ways,
  * the same with the mode forced both ways on the LJSpeech-shaped full-size case (the mode it never takes by itself there),
  * bf16 (the persistent kernels, where the tile map actually removes tiles): the lens + tile-map step is as close to the fp64
    oracle of the bf16-rounded network as the lens-free step of the same batch, tensor by tensor.
Reference: model/fastspeech2.py:43-110 (forward incl. speaker embedding :68-71), transformer/Layers.py:21-30 (masked_fill after
each sub-layer), model/loss.py:19-92."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import fs2_oracle as O
from oracle.weights import seeded_state_dict
from tests.helpers import bf16_matrix, make_model, oracle_train_case
from tests.test_model_gpu import to_dev

pytestmark = pytest.mark.gpu
N_SPEAKER = 2456


def libritts_bucket(batch=48, rank=0, world=1, group_size=4):
    """the batch bench.py's `--workload libritts` builds (bench.py build()): the same pool, sampler, step and generator seeds"""
    from fastspeech2_amd.data import BucketedBatchSampler
    from fastspeech2_amd.synthetic import synthetic_batch
    g = torch.Generator().manual_seed(99)
    pool = torch.clamp(torch.exp(torch.randn(8192, generator=g) * 0.78 + 3.89), 5, 250).long()
    sampler = BucketedBatchSampler(pool.numpy(), batch, world_size=world, rank=rank, group_size=group_size, shuffle=True, seed=1234)
    steps = list(iter(sampler))
    idxs = steps[len(steps) // 3]
    return synthetic_batch(1234 + rank, 0, 0, dur_lo=4, dur_hi=10, n_speaker=N_SPEAKER, src_lens=pool[idxs].tolist())


def rows_of(b, rows):
    """a sub-batch (row indices) padded to ITS OWN maxima, as the collate function would make it"""
    rows = torch.as_tensor(rows)
    L, T = int(b["src_lens"][rows].max()), int(b["mel_lens"][rows].max())
    return dict(speakers=b["speakers"][rows], texts=b["texts"][rows][:, :L], src_lens=b["src_lens"][rows], max_src_len=L,
                mels=b["mels"][rows][:, :T], mel_lens=b["mel_lens"][rows], max_mel_len=T, pitches=b["pitches"][rows][:, :L],
                energies=b["energies"][rows][:, :L], durations=b["durations"][rows][:, :L])


@pytest.fixture(scope="module")
def libri(tmp_path_factory):
    from fastspeech2_amd.synthetic import LJ_STATS, make_configs
    d = str(tmp_path_factory.mktemp("libritts_cfg"))
    json.dump({f"spk{i}": i for i in range(N_SPEAKER)}, open(os.path.join(d, "speakers.json"), "w"))
    json.dump(LJ_STATS, open(os.path.join(d, "stats.json"), "w"))
    pcfg, mcfg = make_configs(dec_layers=4, enc_layers=4, multi_speaker=True, dropout=False, data_dir=d)
    b = libritts_bucket()
    assert b["max_src_len"] >= 150 and b["max_mel_len"] >= 900, (b["max_src_len"], b["max_mel_len"])
    valid = float(b["mel_lens"].sum()) / (48 * b["max_mel_len"])
    assert 0.35 < valid < 0.6, valid                                      # the bench line's "46 % valid rows"
    model = make_model(pcfg, mcfg, "fp32")
    assert model.speaker_emb.weight.shape == (N_SPEAKER, 256)            # model/fastspeech2.py:31-40
    sd = seeded_state_dict(model.state_dict(), 2026)
    sub = rows_of(b, list(range(0, 48, 3)))                               # 16 utterances incl. the longest: the same length profile
    assert sub["max_mel_len"] == b["max_mel_len"] and sub["max_src_len"] == b["max_src_len"]
    return pcfg, mcfg, sd, b, sub


@pytest.fixture(scope="module")
def libri_oracle(libri):
    pcfg, mcfg, sd, b, sub = libri
    oout, olosses, ograds, _ = oracle_train_case(pcfg, mcfg, sd, sub, dtype=torch.float64)
    sdr = {k: (v.to(torch.bfloat16).to(v.dtype) if bf16_matrix(k, v) else v) for k, v in sd.items()}
    wout, wlosses, wgrads, _ = oracle_train_case(pcfg, mcfg, sdr, sub, dtype=torch.float64)
    return (oout, olosses, ograds), (wout, wlosses, wgrads)


class ModeRecorder:
    """wraps ops.conv_gemm as the engine sees it: which FFT-block contractions carried lens / a tile map, by row count per sequence"""

    def __init__(self, monkeypatch):
        from fastspeech2_amd import engine
        self.calls = []
        real = engine.ops.conv_gemm

        def spy(x, wf, b, S, *a, **kw):
            self.calls.append((int(S), kw.get("lens") is not None, kw.get("tmap") is not None, bool(kw.get("ragged"))))
            return real(x, wf, b, S, *a, **kw)

        monkeypatch.setattr(engine.ops, "conv_gemm", spy)

    def ragged(self, S):
        """(calls that COULD take lens, calls that did, calls that also had a tile map) among the launches over S rows per sequence"""
        r = [c for c in self.calls if c[0] == S and c[3]]
        return len(r), sum(c[1] for c in r), sum(c[1] and c[2] for c in r)


def loader_batch(b, dev):
    """the batch on the device the way train.py's loader hands it over: the two length vectors carry their host copies
    (utils.lens_to_device / data.DevicePrefetcher) - that is what the engine's per-batch mode decision reads"""
    from fastspeech2_amd.utils import lens_to_device
    d = to_dev(b, dev)
    d["src_lens"], d["mel_lens"] = lens_to_device(b["src_lens"], dev), lens_to_device(b["mel_lens"], dev)
    return d


def product_step(dev, pcfg, mcfg, sd, b, cdt, lens_mode=None):
    from fastspeech2_amd.model import FastSpeech2Loss
    model = make_model(pcfg, mcfg, cdt)
    model.load_state_dict(sd)
    model.to(dev).train()
    model.disable_dropout = True
    model._ensure_flat(dev)
    model._engine.gemm_lens_fwd = model._engine.gemm_lens_bwd = lens_mode
    d = loader_batch(b, dev)
    batch12 = (None, None, d["speakers"], d["texts"], d["src_lens"], d["max_src_len"], d["mels"], d["mel_lens"], d["max_mel_len"],
               d["pitches"], d["energies"], d["durations"])
    out = model(*batch12[2:])
    losses = FastSpeech2Loss(pcfg, mcfg)(batch12, out)
    losses[0].backward()
    grads = {n: p.grad.detach().cpu().double() for n, p in model.named_parameters() if p.grad is not None}
    return out, losses, grads


def assert_step_matches_fp64(out, losses, grads, oout, olosses, ograds, tag):
    """the full-size fp32 bar of tests/test_a_prodshape_gpu.py: outputs at the north_star bar, every gradient tensor elementwise"""
    assert torch.equal(out[9].cpu(), oout[9]) and torch.equal(out[7].cpu(), oout[7]) and torch.equal(out[6].cpu(), oout[6]), tag
    for i in (0, 1):
        l1 = (out[i].detach().float().cpu().double() - oout[i].detach()).abs().mean().item()
        assert l1 < 1e-4, (tag, i, l1)
    for a, o in zip(losses, olosses):
        assert abs(a.item() - o.item()) <= 1e-5 * max(1.0, abs(o.item())), (tag, a.item(), o.item())
    assert sorted(grads) == sorted(ograds), tag
    gmax = max(g.abs().max().item() for g in ograds.values())
    for n, og in ograds.items():
        scale = og.abs().max().item()
        err = (grads[n] - og).abs().max().item()
        if scale < 1e-9 * gmax:                                   # true gradient zero (w_ks.bias, conv biases in front of BatchNorm)
            assert err <= 1e-6 * gmax, (tag, n, err, gmax)
            continue
        assert err <= 2e-3 * scale, (tag, n, err, scale)
        assert ((grads[n] - og).norm() / og.norm()).item() <= 1e-3, (tag, n)


def test_engine_chooses_the_lens_tile_map_mode_on_this_bucket_and_not_on_ljspeech(dev, libri, monkeypatch):
    from fastspeech2_amd.synthetic import synthetic_batch
    pcfg, mcfg, sd, b, sub = libri
    model = make_model(pcfg, mcfg, "bf16")
    model.load_state_dict(sd)
    model.to(dev).train()
    model.disable_dropout = True
    eng_min = None
    for batch, want in ((b, True), (sub, True), (synthetic_batch(1234, 48, 128, dur_lo=4, dur_hi=10, min_len_frac=0.75), False)):
        rec = ModeRecorder(monkeypatch)
        d = loader_batch(batch, dev)
        out = model(d["speakers"], d["texts"], d["src_lens"], d["max_src_len"], d["mels"], d["mel_lens"], d["max_mel_len"],
                    d["pitches"], d["energies"], d["durations"])
        eng = model._engine
        eng_min = eng.lens_skip_min
        T = min(int(batch["max_mel_len"]), mcfg["max_seq_len"])
        skip = eng._skip_fraction(batch["mel_lens"].numpy(), T)
        n, with_lens, with_map = rec.ragged(T)
        assert n >= 4 * 4, (n, rec.calls[:5])                     # 4 decoder layers x (QKV, fc, w_1, w_2)
        if want:
            assert skip >= eng_min, skip
            assert with_lens == n and with_map == n, (n, with_lens, with_map)       # every decoder contraction: lens AND tile map
        else:
            assert skip < eng_min, skip
            assert with_lens == 0, (n, with_lens)
        out[1].float().sum().backward()                          # (backward takes the same decision; exercised, values checked below)
        monkeypatch.undo()
    assert eng_min == 0.10


def test_fp32_eval_forward_of_the_whole_bucket_matches_oracle(dev, libri):
    pcfg, mcfg, sd, b, sub = libri
    model = make_model(pcfg, mcfg, "fp32")
    model.load_state_dict(sd)
    model.to(dev).eval()
    d = to_dev(b, dev)
    with torch.no_grad():
        out = model(d["speakers"], d["texts"], d["src_lens"], d["max_src_len"], d["mels"], d["mel_lens"], d["max_mel_len"],
                    d["pitches"], d["energies"], d["durations"])
        ref = O.fastspeech2_forward(sd, mcfg, pcfg, b["speakers"], b["texts"], b["src_lens"], b["max_src_len"], b["mels"], b["mel_lens"],
                                    b["max_mel_len"], b["pitches"], b["energies"], b["durations"], training=False)
    assert torch.equal(out[9].cpu(), ref[9]) and torch.equal(out[7].cpu(), ref[7]) and torch.equal(out[6].cpu(), ref[6])
    assert torch.equal(out[5].cpu(), ref[5])
    for i in (0, 1):
        l1 = (out[i].float().cpu() - ref[i]).abs().mean().item()
        assert l1 < 1e-4, (i, l1)                                 # north_star bar (fp32)
    for i in (2, 3, 4):
        assert torch.allclose(out[i].float().cpu(), ref[i], atol=2e-4, rtol=1e-4), i


@pytest.mark.parametrize("mode", [None, True, False], ids=["engine-chosen", "lens-forced-on", "lens-forced-off"])
def test_fp32_train_step_on_the_bucket_matches_fp64_oracle_elementwise(dev, libri, libri_oracle, mode):
    pcfg, mcfg, sd, b, sub = libri
    (oout, olosses, ograds), _ = libri_oracle
    assert "speaker_emb.weight" in ograds
    out, losses, grads = product_step(dev, pcfg, mcfg, sd, sub, "fp32", mode)
    assert_step_matches_fp64(out, losses, grads, oout, olosses, ograds, mode)
    # only the speakers of the batch receive a gradient (an embedding row per utterance, model/fastspeech2.py:68-71)
    rows = grads["speaker_emb.weight"].abs().sum(1) > 0
    assert set(torch.nonzero(rows).flatten().tolist()) <= set(sub["speakers"].tolist())


@pytest.mark.parametrize("mode", [True, False], ids=["lens-forced-on", "lens-forced-off"])
def test_fp32_ljspeech_full_size_step_with_the_mode_forced_matches_fp64_oracle(dev, full_case, mode):
    """the LJSpeech-shaped full-size case takes the lens-free mode by itself (tests/test_a_prodshape_gpu.py compares THAT with the
    oracle); here both modes are forced on it"""
    pcfg, mcfg, sd, b, oout, olosses, ograds = full_case
    out, losses, grads = product_step(dev, pcfg, mcfg, sd, b, "fp32", mode)
    assert_step_matches_fp64(out, losses, grads, oout, olosses, ograds, mode)


def test_bf16_lens_tile_map_step_is_as_close_to_the_oracle_as_the_lens_free_step(dev, libri, libri_oracle):
    """bf16 runs the persistent / wide / streaming kernels - the ones that DROP padded tiles from their tile lists.  Against the
    fp64 oracle of the network the bf16 engine differentiates (matrices rounded to bf16), per tensor: the lens + tile-map step's
    relative Frobenius distance <= 1.5 x the lens-free step's + 1e-3 (the lens-free arithmetic is what the frozen bf16 bars of
    tests/test_z_bf16_budget_gpu.py pin at the LJSpeech shape; on valid rows the two modes multiply the same numbers), and both
    inside an absolute sanity bound."""
    pcfg, mcfg, sd, b, sub = libri
    _, (wout, wlosses, wgrads) = libri_oracle
    res = {}
    for mode in (True, False):
        out, losses, grads = product_step(dev, pcfg, mcfg, sd, sub, "bf16", mode)
        assert torch.equal(out[9].cpu(), wout[9]) and torch.equal(out[7].cpu(), wout[7])
        valid = (~wout[7]).unsqueeze(-1)
        l1 = [((out[i].detach().float().cpu().double() - wout[i].detach()).abs() * valid).sum().item() / (valid.sum().item() * 80) for i in (0, 1)]
        lrel = [abs(a.item() - o.item()) / max(1.0, abs(o.item())) for a, o in zip(losses, wlosses)]
        res[mode] = (l1, lrel, grads)
    gmax = max(g.abs().max().item() for g in wgrads.values())
    worst = []
    for n, og in wgrads.items():
        if og.abs().max().item() < 1e-9 * gmax:
            for mode in (True, False):
                assert res[mode][2][n].abs().max().item() <= 1e-3 * gmax, (n, mode)
            continue
        f_on = ((res[True][2][n] - og).norm() / og.norm()).item()
        f_off = ((res[False][2][n] - og).norm() / og.norm()).item()
        worst.append((f_on / max(f_off, 1e-12), f_on, f_off, n))
        assert f_on <= 1.5 * f_off + 1e-3, (n, f_on, f_off)
        assert f_on <= 0.12 and f_off <= 0.12, (n, f_on, f_off)
    for i in (0, 1):
        assert res[True][0][i] <= 1.5 * res[False][0][i] + 1e-3, (i, res[True][0], res[False][0])
        assert res[True][0][i] < 0.05, res[True][0]
    for a, o in zip(res[True][1], res[False][1]):
        assert a <= 1.5 * o + 2e-3, (res[True][1], res[False][1])
    worst.sort(reverse=True)
    print(f"bf16 LibriTTS bucket (B = 16 slice, T = {sub['max_mel_len']}): valid-frame mel L1 lens-on {res[True][0]} lens-off {res[False][0]}; "
          f"largest per-tensor (lens-on / lens-off) distance ratios: {[(f'{r:.2f}', f'{a:.2e}', f'{o:.2e}', n) for r, a, o, n in worst[:5]]}")
