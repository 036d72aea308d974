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


from fastspeech2_amd.synthetic import synthetic_batch  # noqa: E402,F401  (the workload generator is product code; re-exported for the tests)
