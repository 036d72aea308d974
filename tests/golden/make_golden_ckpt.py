"""Checkpoint / resume golden from the LIVE reference (run in the build container; needs /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_ckpt.py

`reference_two_steps(dir)` runs the unmodified reference modules (FastSpeech2, FastSpeech2Loss, ScheduledOptim = torch.optim.Adam
+ the LR schedule) for TWO optimiser steps exactly as train.py:82-97 does (forward, loss, backward, clip_grad_norm_, 
step_and_update_lr, zero_grad), writes `{dir}/2.pth.tar` = {"model", "optimizer"} exactly as train.py:152-161 does, then takes
step 3 and returns its losses / learning rate / updated parameters.  A full checkpoint is ~70 MB (the PostNet's width is not
configurable), so it is NOT committed: ckpt_resume.npz holds per-tensor checksums of the reference-written file (model, exp_avg,
exp_avg_sq), the step-3 results and a few full tensors; tests/test_checkpoint_cpu.py (which runs where /root/reference exists)
re-runs this function and loads the real file into the product, and the GPU test rebuilds the same file from the oracle +
torch.optim.Adam, proves it against these checksums and resumes the product from it.
"""
import os
import sys

import numpy as np

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402  (installs the unidecode / inflect stubs, sets sys.path)
import torch  # noqa: E402
from tests.golden import configs  # noqa: E402
from oracle.weights import seeded_state_dict, synthetic_batch  # noqa: E402

SEED, B, L = 606, 3, 14
CFG = dict(dec_layers=2, enc_layers=2)
FULL = ("mel_linear.bias", "decoder.layer_stack.1.pos_ffn.layer_norm.weight", "encoder.layer_stack.0.slf_attn.fc.bias",
        "variance_adaptor.pitch_predictor.linear_layer.weight", "postnet.convolutions.4.1.bias")


def tensor_stats(t):
    t = t.double()
    return [t.sum().item(), t.abs().sum().item(), t.norm().item()]


def checkpoint_stats(ckpt, param_names):
    """per-tensor [sum, abs-sum, norm] of model / exp_avg / exp_avg_sq (Adam state is indexed by parameters() order)."""
    out = {"model:" + k: tensor_stats(v) for k, v in ckpt["model"].items() if v.is_floating_point()}
    for i, st in ckpt["optimizer"]["state"].items():
        out["exp_avg:" + param_names[int(i)]] = tensor_stats(st["exp_avg"])
        out["exp_avg_sq:" + param_names[int(i)]] = tensor_stats(st["exp_avg_sq"])
    return out


def reference_two_steps(out_dir):
    import torch.nn as nn
    import torch.nn.functional as F
    from model import FastSpeech2Loss, ScheduledOptim
    cwd = os.getcwd()
    os.chdir(MG.REF)
    real_dropout = F.dropout
    F.dropout = lambda x, p=0.5, training=True, inplace=False: x       # dropout off (PostNet hard-codes F.dropout(0.5))
    try:
        pcfg, mcfg = configs.make(dropout=False, **CFG)
        model = MG.build_reference(pcfg, mcfg)
        model.load_state_dict(seeded_state_dict(model.state_dict(), SEED))
        model.train()
        b = synthetic_batch(SEED + 1, B, L)
        batch12 = (None, None, b["speakers"], b["texts"], b["src_lens"], b["max_src_len"], b["mels"], b["mel_lens"],
                   b["max_mel_len"], b["pitches"], b["energies"], b["durations"])
        loss_fn = FastSpeech2Loss(pcfg, mcfg)
        opt = ScheduledOptim(model, configs.TRAIN, mcfg, 0)
        names = [n for n, _ in model.named_parameters()]
        res = {}
        for step in (1, 2, 3):
            out = model(*batch12[2:])
            losses = loss_fn(batch12, out)
            losses[0].backward()
            nn.utils.clip_grad_norm_(model.parameters(), configs.TRAIN["optimizer"]["grad_clip_thresh"])
            if step == 3:
                before = {n: dict(model.named_parameters())[n].detach().clone() for n in FULL}
            opt.step_and_update_lr()
            opt.zero_grad()
            if step == 2:
                path = os.path.join(out_dir, "2.pth.tar")
                torch.save({"model": model.state_dict(), "optimizer": opt._optimizer.state_dict()}, path)     # train.py:152-161
                res["path"] = path
            if step == 3:
                res["losses3"] = np.array([float(l) for l in losses])
                res["lr3"] = float(opt._optimizer.param_groups[0]["lr"])
                res["after3"] = {n: dict(model.named_parameters())[n].detach().clone() for n in FULL}
                res["before3"] = before
        res["param_names"] = names
        return res
    finally:
        F.dropout = real_dropout
        os.chdir(cwd)


if __name__ == "__main__":
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        r = reference_two_steps(d)
        ckpt = torch.load(r["path"], map_location="cpu", weights_only=False)
        stats = checkpoint_stats(ckpt, r["param_names"])
        keys = sorted(stats)
        extra = {}
        for n in FULL:
            extra["p2:" + n] = r["before3"][n].numpy()
            extra["p3:" + n] = r["after3"][n].numpy()
        st0 = next(iter(ckpt["optimizer"]["state"].values()))
        np.savez_compressed(os.path.join(HERE, "ckpt_resume.npz"), keys=np.array(keys), stats=np.array([stats[k] for k in keys]),
                            losses3=r["losses3"], lr3=r["lr3"], adam_step=float(st0["step"]),
                            lr_in_ckpt=float(ckpt["optimizer"]["param_groups"][0]["lr"]), torch_version=torch.__version__, **extra)
        print("ckpt", os.path.getsize(r["path"]) >> 20, "MB;", len(keys), "tensors; losses3", r["losses3"], "lr3", r["lr3"])
