"""Shared helpers for parity tests: build seeded weights / batches and run the CPU oracle."""
import numpy as np
import torch

from oracle import fs2_oracle as O
from oracle.weights import seeded_state_dict, synthetic_batch
from tests.golden import configs


def load_golden(tag):
    import os
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", tag + ".npz"), allow_pickle=False)


def make_model(pcfg, mcfg, compute_dtype="fp32"):
    from fastspeech2_amd.model import FastSpeech2
    return FastSpeech2(pcfg, mcfg, compute_dtype=compute_dtype)


def oracle_train_case(pcfg, mcfg, sd, b, dtype=torch.float32):
    """train-mode (dropout neutralised) forward + loss + backward through the oracle; returns outputs, losses, grads."""
    sdr = {k: (v.to(dtype) if v.is_floating_point() else v).clone() for k, v in sd.items()}
    leaves = {}
    for k, v in sdr.items():
        if v.is_floating_point() and not any(s in k for s in ("position_enc", "_bins", "running_")):
            v.requires_grad_(True)
            leaves[k] = v
    bn_buffers = {k: v.clone() for k, v in sdr.items() if "running_" in k}
    out = O.fastspeech2_forward(sdr, mcfg, pcfg, b["speakers"], b["texts"], b["src_lens"], b["max_src_len"],
                                b["mels"].to(dtype), b["mel_lens"], b["max_mel_len"], b["pitches"].to(dtype),
                                b["energies"].to(dtype), b["durations"], training=True, dropout=False, bn_buffers=bn_buffers)
    losses = O.fastspeech2_loss(pcfg, (b["mels"].to(dtype), b["pitches"].to(dtype), b["energies"].to(dtype), b["durations"]), out)
    losses[0].backward()
    grads = {k: v.grad for k, v in leaves.items() if v.grad is not None}
    return out, losses, grads, bn_buffers


def grad_stats(g):
    g = g.double()
    return np.array([g.sum().item(), g.abs().sum().item(), g.norm().item()])
