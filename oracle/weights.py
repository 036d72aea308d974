"""Deterministic, reference-independent weight / batch generators for parity tests (TEST INFRASTRUCTURE).

`seeded_state_dict` fills a state_dict schema (name -> shape/dtype) from a torch.Generator so that the live
reference (in tests/golden/make_golden.py), the oracle and the HIP engine can all be loaded with bit-identical
parameters without shipping 100 MB fixtures.  torch's CPU generator is bit-reproducible for a fixed version,
and the golden fixtures record the torch version they were made with.
"""
import math

import torch


def seeded_state_dict(template_sd, seed, keep=("position_enc", "pitch_bins", "energy_bins", "num_batches_tracked")):
    """template_sd: an existing state_dict (gives names, shapes, dtypes, and the values of `keep` entries)."""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for name in sorted(template_sd.keys()):
        t = template_sd[name]
        if any(k in name for k in keep):
            out[name] = t.clone()
            continue
        shape = tuple(t.shape)
        if name.endswith("running_var"):
            v = torch.rand(shape, generator=g) + 0.5
        elif name.endswith("running_mean"):
            v = torch.randn(shape, generator=g) * 0.1
        elif "layer_norm" in name or (".1." in name and "postnet" in name):      # LayerNorm / BatchNorm affine
            if name.endswith("weight"):
                v = 1.0 + 0.1 * torch.randn(shape, generator=g)
            else:
                v = 0.1 * torch.randn(shape, generator=g)
        elif "embedding" in name or "emb" in name:
            v = torch.randn(shape, generator=g) * 0.5
            if "src_word_emb" in name:
                v[0].zero_()
        elif len(shape) >= 2:
            fan_in = 1
            for s in shape[1:]:
                fan_in *= s
            v = torch.randn(shape, generator=g) * (1.0 / math.sqrt(fan_in))
        else:
            v = torch.randn(shape, generator=g) * 0.1
        out[name] = v.to(t.dtype)
    return out


def synthetic_batch(seed, B, L, dur_lo=2, dur_hi=8, n_mel=80, n_vocab=361, n_speaker=1, min_len_frac=0.6,
                    max_seq_len=1000, frame_level=False):
    """LJSpeech-shaped synthetic batch: the reference's 12-tuple minus ids/raw_texts.
    Returns dict of CPU tensors + python ints."""
    g = torch.Generator().manual_seed(seed)
    lo = max(1, int(L * min_len_frac))
    src_lens = torch.randint(lo, L + 1, (B,), generator=g)
    src_lens[0] = L
    src_lens, _ = torch.sort(src_lens, descending=True)
    texts = torch.randint(1, n_vocab, (B, L), generator=g)
    durations = torch.randint(dur_lo, dur_hi + 1, (B, L), generator=g)
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
