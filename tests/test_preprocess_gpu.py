"""Corpus preprocessor on the GPU (fastspeech2_amd/preprocess.py + audio.TacotronSTFT.mel_spectrogram_ragged): the same
synthetic corpus as tests/test_preprocess_cpu.py, mel extraction through the HIP STFT in ragged batches, against the outputs
of the reference's own preprocessor (tests/golden/preproc_*.npz)."""
import numpy as np
import pytest
import torch

from fastspeech2_amd import preprocess as P
from tests.test_preprocess_cpu import _run, check_against_golden

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("tag,batch_seconds", [("phoneme", 1800.0), ("frame", 2.5)])
def test_preprocessor_gpu_matches_reference_golden(dev, tmp_path, monkeypatch, tag, batch_seconds):
    """batch_seconds=2.5 forces several ragged batches (the corpus holds ~4.5 s of trimmed audio)."""
    calls = []
    real = P.Preprocessor._extract_mels

    def spy(self, wavs):
        calls.append(len(wavs))
        return real(self, wavs)
    monkeypatch.setattr(P.Preprocessor, "_extract_mels", spy)
    cfg, _, out = _run(tmp_path, monkeypatch, P.Preprocessor, tag == "frame", device=dev, batch_seconds=batch_seconds)
    assert sum(calls) == 4 and (len(calls) == 1 if batch_seconds > 100 else len(calls) > 1), calls
    check_against_golden(cfg, out, tag, mel_atol=2e-4)


def test_ragged_batch_equals_single_utterances(dev):
    from fastspeech2_amd.audio import TacotronSTFT
    stft = TacotronSTFT(1024, 256, 1024, 80, 22050, 0, 8000).to(dev)
    g = torch.Generator().manual_seed(9)
    lens = [22050, 513, 256 * 30, 9001, 4000]
    N = max(lens)
    y = torch.zeros(len(lens), N)
    for b, n in enumerate(lens):
        t = torch.arange(n, dtype=torch.float32) / 22050.0
        y[b, :n] = torch.clamp(0.4 * torch.sin(2 * np.pi * (150.0 + 70 * b) * t) + 0.1 * torch.randn(n, generator=g), -1, 1)
        y[b, n:] = 0.77                                                     # garbage beyond the length must not leak in
    mel, energy, frames = stft.mel_spectrogram_ragged(y.to(dev), lens)
    assert frames.tolist() == [n // 256 + 1 for n in lens] and mel.shape == (len(lens), 80, N // 256 + 1)
    for b, n in enumerate(lens):
        m1, e1 = stft.mel_spectrogram(y[b:b + 1, :n].to(dev))
        f = n // 256 + 1
        # same products and fp32 accumulation; bit-identical unless the GEMM picks a different tile variant for the two shapes
        assert (mel[b, :, :f] - m1[0]).abs().max().item() <= 2e-5, b
        assert torch.allclose(energy[b, :f], e1[0], rtol=1e-6, atol=1e-5), b
    with pytest.raises(AssertionError):
        stft.mel_spectrogram_ragged(y.to(dev), [22050, 512, 100, 9001, 4000])   # reflect padding needs > filter_length / 2 samples
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        stft.mel_spectrogram_ragged(y, lens)


def test_raw_corpus_to_train_step(dev, tmp_path, monkeypatch):
    """raw wavs + TextGrids -> GPU preprocessor -> the reference's on-disk format -> Dataset / collate -> one training step."""
    import copy
    from fastspeech2_amd.data import Dataset
    from fastspeech2_amd.model import FastSpeech2, FastSpeech2Loss, ScheduledOptim
    from fastspeech2_amd.utils import to_device
    from tests.golden import configs

    cfg, _, out = _run(tmp_path, monkeypatch, P.Preprocessor, False, device=dev)
    cfg["preprocessing"]["val_size"] = 2
    pcfg, mcfg = configs.make(dec_layers=1, enc_layers=1)
    pcfg = copy.deepcopy(pcfg)
    pcfg["path"] = dict(pcfg["path"], preprocessed_path=cfg["path"]["preprocessed_path"])
    pcfg["dataset"] = cfg["dataset"]
    tcfg = copy.deepcopy(configs.TRAIN)
    tcfg["optimizer"]["batch_size"] = 2
    ds = Dataset("train.txt", pcfg, tcfg, sort=True, drop_last=True)
    assert len(ds) == 2
    batch = ds.collate_fn([ds[i] for i in range(len(ds))])[0]
    assert batch[6].shape[2] == 80 and batch[11].sum(1).tolist() == batch[7].tolist()    # durations sum to the mel lengths
    model = FastSpeech2(pcfg, mcfg, compute_dtype="fp32").to(dev).train()
    opt = ScheduledOptim(model, tcfg, mcfg, 0)
    loss_fn = FastSpeech2Loss(pcfg, mcfg)
    b = to_device(batch, dev)
    losses = loss_fn(b, model(*b[2:]))
    losses[0].backward()
    opt.step_and_update_lr()
    assert all(torch.isfinite(l).item() for l in losses)
