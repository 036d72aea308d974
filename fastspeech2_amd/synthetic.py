"""Synthetic LJSpeech-/LibriTTS-shaped workloads and the reference's YAML configuration as dicts.

Used by bench.py (the measured workload), the tools/ scripts and the parity tests; there is no network for datasets or
checkpoints, so batches are generated here with the statistics of the real corpus (reference
preprocessed_data/LJSpeech/stats.json, config/LJSpeech/{model,preprocess,train}.yaml).
"""
import copy
import json
import os
import tempfile

import torch

# reference preprocessed_data/LJSpeech/stats.json (pitch/energy [min, max, mean, std])
LJ_STATS = {"pitch": [-2.917079304729967, 11.391254536985784, 207.6309860026605, 46.77559025098988],
            "energy": [-1.431044578552246, 8.184337615966797, 37.32621679053821, 26.044180782835863]}

MODEL = {
    "transformer": {"encoder_layer": 4, "encoder_head": 2, "encoder_hidden": 256, "decoder_layer": 4, "decoder_head": 2,
                    "decoder_hidden": 256, "conv_filter_size": 1024, "conv_kernel_size": [9, 1], "encoder_dropout": 0.2,
                    "decoder_dropout": 0.2},
    "variance_predictor": {"filter_size": 256, "kernel_size": 3, "dropout": 0.5},
    "variance_embedding": {"pitch_quantization": "linear", "energy_quantization": "linear", "n_bins": 256},
    "multi_speaker": False,
    "max_seq_len": 1000,
    "vocoder": {"model": "HiFi-GAN", "speaker": "LJSpeech"},
}

PREPROCESS = {
    "dataset": "LJSpeech",
    "path": {"preprocessed_path": None},
    "preprocessing": {
        "val_size": 512,
        "text": {"text_cleaners": ["english_cleaners"], "language": "en"},
        "audio": {"sampling_rate": 22050, "max_wav_value": 32768.0},
        "stft": {"filter_length": 1024, "hop_length": 256, "win_length": 1024},
        "mel": {"n_mel_channels": 80, "mel_fmin": 0, "mel_fmax": 8000},
        "pitch": {"feature": "phoneme_level", "normalization": True},
        "energy": {"feature": "phoneme_level", "normalization": True},
    },
}

def val_phoneme_counts():
    """phoneme count of every line of the reference's preprocessed_data/LJSpeech/val.txt, in file order (512 utterances;
    table made by tools/make_val_shape.py from the reference, committed under fastspeech2_amd/workloads/)."""
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "workloads", "ljspeech_val_phonemes.json")) as f:
        return json.load(f)["counts"]


TRAIN = {
    "path": {"ckpt_path": "./output/ckpt/LJSpeech", "log_path": "./output/log/LJSpeech", "result_path": "./output/result/LJSpeech"},
    "optimizer": {"batch_size": 48, "betas": [0.9, 0.98], "eps": 1e-9, "weight_decay": 0.0, "grad_clip_thresh": 1.0,
                  "grad_acc_step": 1, "warm_up_step": 4000, "anneal_steps": [300000, 400000, 500000], "anneal_rate": 0.3},
    "step": {"total_step": 900000, "log_step": 100, "synth_step": 1000, "val_step": 1000, "save_step": 100000},
}

_DATA_DIR = None


def ensure_data(path=None, n_speaker=4):
    """A directory holding the two files the model constructor reads (stats.json, speakers.json); a fresh temp dir by default."""
    global _DATA_DIR
    if path is None:
        if _DATA_DIR is None:
            _DATA_DIR = tempfile.mkdtemp(prefix="fs2_synth_")
        path = _DATA_DIR
    os.makedirs(path, exist_ok=True)
    p = os.path.join(path, "stats.json")
    if not os.path.exists(p):
        with open(p, "w") as f:
            json.dump(LJ_STATS, f)
    p = os.path.join(path, "speakers.json")
    if not os.path.exists(p):
        with open(p, "w") as f:
            json.dump({f"spk{i}": i for i in range(n_speaker)}, f)
    return path


def make_configs(dec_layers=4, multi_speaker=False, frame_level=False, dropout=True, enc_layers=4, max_seq_len=1000, data_dir=None):
    """(preprocess_config, model_config) in the reference's YAML structure."""
    m = copy.deepcopy(MODEL)
    p = copy.deepcopy(PREPROCESS)
    p["path"]["preprocessed_path"] = ensure_data(data_dir)
    m["transformer"]["decoder_layer"] = dec_layers
    m["transformer"]["encoder_layer"] = enc_layers
    m["multi_speaker"] = multi_speaker
    m["max_seq_len"] = max_seq_len
    if frame_level:
        p["preprocessing"]["pitch"]["feature"] = "frame_level"
        p["preprocessing"]["energy"]["feature"] = "frame_level"
    if not dropout:
        m["transformer"]["encoder_dropout"] = 0.0
        m["transformer"]["decoder_dropout"] = 0.0
        m["variance_predictor"]["dropout"] = 0.0
    return p, m


def synthetic_batch(seed, B, L, dur_lo=2, dur_hi=8, n_mel=80, n_vocab=361, n_speaker=1, min_len_frac=0.6,
                    max_seq_len=1000, frame_level=False, src_lens=None, sort=True, utt_durations=None):
    """LJSpeech-shaped synthetic batch: the reference's 12-tuple minus ids/raw_texts.
    Returns dict of CPU tensors + python ints.  src_lens (optional): the utterances' phoneme counts (then B = len(src_lens),
    L = their maximum; `sort` = False keeps their order, as a TextDataset batch does).  utt_durations (optional, with src_lens):
    one 1-D integer tensor of per-phoneme durations PER UTTERANCE - an utterance then has the same frames whatever batch it is
    dealt into (a corpus: tools/bench_libritts_sweep.py); without it the durations are drawn per batch and the whole batch is
    shrunk until its longest utterance fits max_seq_len."""
    g = torch.Generator().manual_seed(seed)
    if src_lens is not None:
        src_lens = torch.as_tensor(src_lens, dtype=torch.int64).clone()
        B, L = int(src_lens.numel()), int(src_lens.max())
    else:
        lo = max(1, int(L * min_len_frac))
        src_lens = torch.randint(lo, L + 1, (B,), generator=g)
        src_lens[0] = L
    if sort:
        src_lens, _ = torch.sort(src_lens, descending=True)
    texts = torch.randint(1, n_vocab, (B, L), generator=g)
    durations = torch.randint(dur_lo, dur_hi + 1, (B, L), generator=g)
    if utt_durations is not None:
        assert not sort and len(utt_durations) == B
        durations = torch.zeros(B, L, dtype=torch.int64)
        for i, d in enumerate(utt_durations):
            assert d.numel() == int(src_lens[i]) and int(d.sum()) <= max_seq_len
            durations[i, :d.numel()] = d
    valid = torch.arange(L).unsqueeze(0) < src_lens.unsqueeze(1)
    texts = texts * valid
    durations = durations * valid
    mel_lens = durations.sum(1)
    while int(mel_lens.max()) > max_seq_len:          # keep max mel_len <= max_seq_len as the survey prescribes
        durations = torch.clamp(durations - 1, min=0) * valid
        mel_lens = durations.sum(1)
    T = int(mel_lens.max())
    mels = torch.clamp(torch.randn(B, T, n_mel, generator=g) * 2 - 5, -11.5, 2.0)
    mels = mels * (torch.arange(T).unsqueeze(0) < mel_lens.unsqueeze(1)).unsqueeze(-1)
    n_var = T if frame_level else L
    var_valid = (torch.arange(T).unsqueeze(0) < mel_lens.unsqueeze(1)) if frame_level else valid
    pitches = torch.clamp(torch.randn(B, n_var, generator=g), -2.917, 11.391) * var_valid
    energies = torch.clamp(torch.randn(B, n_var, generator=g), -1.431, 8.184) * var_valid
    speakers = torch.randint(0, n_speaker, (B,), generator=g)
    return dict(speakers=speakers, texts=texts, src_lens=src_lens, max_src_len=L, mels=mels.float(), mel_lens=mel_lens,
                max_mel_len=T, pitches=pitches.float(), energies=energies.float(), durations=durations)
