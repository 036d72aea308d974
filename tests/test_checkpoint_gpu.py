"""(f)3 on the device: the product resumes from a checkpoint in the reference's format and its step 3 equals the reference's.

The 200 MB file the live reference wrote is not committed; `oracle_written_checkpoint` rebuilds it (oracle + torch.optim.Adam),
tests/test_checkpoint_cpu.py proves that stand-in equal to the reference-written file where the reference exists and against the
committed per-tensor checksums everywhere, and tests/golden/ckpt_resume.npz holds the LIVE reference's step-3 losses, learning
rate and updated parameters (tests/golden/make_golden_ckpt.py).  Reference: train.py:82-97,152-161, utils/model.py:15-28,
model/optimizer.py:19-51."""
import types

import numpy as np
import pytest
import torch

from tests.golden import configs
from tests.helpers import load_golden, oracle_written_checkpoint

pytestmark = pytest.mark.gpu


def test_resume_from_reference_format_checkpoint_step3_matches_reference(dev, tmp_path):
    from fastspeech2_amd.model import FastSpeech2Loss
    from fastspeech2_amd.utils import get_model
    g = load_golden("ckpt_resume")
    pcfg, mcfg, b, ock, names = oracle_written_checkpoint(str(tmp_path / "2.pth.tar"))
    tcfg = dict(configs.TRAIN, path=dict(configs.TRAIN["path"], ckpt_path=str(tmp_path)))
    model, opt = get_model(types.SimpleNamespace(restore_step=2), (pcfg, mcfg, tcfg), dev, train=True, compute_dtype="fp32")
    model.disable_dropout = True
    d = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in b.items()}
    batch12 = (None, None, d["speakers"], d["texts"], d["src_lens"], d["max_src_len"], d["mels"], d["mel_lens"], d["max_mel_len"],
               d["pitches"], d["energies"], d["durations"])
    before = {n: p.detach().clone() for n, p in model.named_parameters()}
    out = model(*batch12[2:])
    losses = FastSpeech2Loss(pcfg, mcfg)(batch12, out)
    losses[0].backward()
    opt.step_and_update_lr()                                   # clip (folded) + Adam at the schedule's step-3 learning rate
    opt.zero_grad()
    got = np.array([l.item() for l in losses])
    assert np.allclose(got, g["losses3"], rtol=1e-5, atol=1e-6), (got, g["losses3"])
    assert opt.current_step == 3 and opt._adam_step == 3
    assert abs(opt.last_lr - float(g["lr3"])) <= 1e-12 * float(g["lr3"])
    P = dict(model.named_parameters())
    for k in g.files:
        if not k.startswith("p3:"):
            continue
        n = k[3:]
        ref_delta = torch.from_numpy(g["p3:" + n] - g["p2:" + n]).double()
        delta = (P[n].detach().double() - before[n].double()).cpu()
        assert ref_delta.abs().max() > 0
        # Adam's update is lr * m_hat / (sqrt(v_hat) + eps) ~ 7e-7 per element: a few fp32 ulps of the parameter itself, so the
        # comparison is made on the UPDATE, to 2 % of its largest entry (one ulp of a 0.1-sized weight is 1 % of the update)
        assert (delta - ref_delta).abs().max().item() <= 0.02 * ref_delta.abs().max().item(), n
        assert torch.allclose(P[n].detach().cpu(), torch.from_numpy(g["p3:" + n]), rtol=0, atol=2e-8 + 1e-7 * float(np.abs(g["p3:" + n]).max())), n
    # BatchNorm's forward counter continues from the checkpoint
    assert int(dict(model.named_buffers())["postnet.convolutions.0.1.num_batches_tracked"]) == 3
