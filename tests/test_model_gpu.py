"""GPU parity of the whole hot path through the drop-in surface: HIP engine vs golden vectors from the live
reference and vs the CPU oracle on the same seeded inputs.  Bar (north_star): mel outputs within 1e-4 L1 in
fp32, LengthRegulator lengths / rounded durations bit-exact."""
import numpy as np
import pytest
import torch

from oracle import fs2_oracle as O
from oracle.weights import seeded_state_dict, synthetic_batch
from tests.golden import configs
from tests.helpers import grad_stats, load_golden, make_model, oracle_train_case
from tests.test_oracle_golden import TRAIN_CASES

pytestmark = pytest.mark.gpu
MEL_L1_TOL = 1e-4          # north_star tolerance (fp32)


def to_dev(b, dev):
    return {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in b.items()}


def run_train(model, pcfg, mcfg, b, dev):
    from fastspeech2_amd.model import FastSpeech2Loss
    d = to_dev(b, dev)
    out = model(d["speakers"], d["texts"], d["src_lens"], d["max_src_len"], d["mels"], d["mel_lens"], d["max_mel_len"],
                d["pitches"], d["energies"], d["durations"])
    batch12 = (None, None, d["speakers"], d["texts"], d["src_lens"], d["max_src_len"], d["mels"], d["mel_lens"],
               d["max_mel_len"], d["pitches"], d["energies"], d["durations"])
    losses = FastSpeech2Loss(pcfg, mcfg)(batch12, out)
    losses[0].backward()
    return out, losses


def test_length_hints_never_change_results(dev):
    """FastSpeech2.set_length_hint only steers whether the contractions skip padded tiles: a hint for one side only
    falls back to the tensor's own host copy on the other, a hint left over from ANOTHER batch (wrong size) is dropped - and in
    every case the outputs are those of the hint-free forward."""
    pcfg, mcfg = configs.make(dropout=False, dec_layers=2, enc_layers=2)
    model = make_model(pcfg, mcfg, "bf16")
    model.load_state_dict(seeded_state_dict(model.state_dict(), 4))
    model.to(dev).train()
    model.disable_dropout = True
    b = synthetic_batch(9, 6, 40, min_len_frac=0.3)
    d = to_dev(b, dev)
    args = (d["speakers"], d["texts"], d["src_lens"], d["max_src_len"], d["mels"], d["mel_lens"], d["max_mel_len"], d["pitches"],
            d["energies"], d["durations"])
    with torch.no_grad():
        ref = model(*args)
        outs = []
        for hint in ((None, b["mel_lens"].numpy()), (b["src_lens"].numpy(), None), (np.array([3, 4]), np.array([30, 40])),
                     (b["src_lens"].numpy(), b["mel_lens"].numpy())):
            model.set_length_hint(*hint)
            outs.append(model(*args))
            assert model._engine.length_hint is None                     # consumed (or dropped) by that forward
    for o in outs:
        assert torch.equal(o[9], ref[9])
        assert (o[1].float() - ref[1].float()).abs().max().item() <= 2e-2 * ref[1].float().abs().max().item()


@pytest.mark.parametrize("tag", list(TRAIN_CASES))
def test_train_step_matches_reference_golden(dev, tag):
    c = TRAIN_CASES[tag]
    g = load_golden(tag)
    pcfg, mcfg = configs.make(dropout=False, **c["cfg"])
    model = make_model(pcfg, mcfg, "fp32")
    model.load_state_dict(seeded_state_dict(model.state_dict(), int(g["seed"])))
    model.to(dev).train()
    model.disable_dropout = True
    nspk = 4 if c["cfg"].get("multi_speaker") else 1
    b = synthetic_batch(int(g["seed"]) + 1, c["B"], c["L"], n_speaker=nspk, frame_level=c["cfg"].get("frame_level", False),
                        max_seq_len=c.get("batch_max_seq_len") or mcfg["max_seq_len"])
    out, losses = run_train(model, pcfg, mcfg, b, dev)
    for name, idx in (("mel", 0), ("post", 1), ("p_pred", 2), ("e_pred", 3), ("logd", 4)):
        ref = torch.from_numpy(g[name])
        got = out[idx].detach().float().cpu()
        assert got.shape == ref.shape, name
        l1 = (got - ref).abs().mean().item()
        assert l1 < MEL_L1_TOL, (name, l1)
        assert l1 < 2e-5, (name, l1)            # fp32 MFMA path is far inside the bar
    assert np.array_equal(out[9].cpu().numpy(), g["mel_lens"])
    assert np.array_equal(out[7].cpu().numpy(), g["mel_masks"])
    assert np.allclose([l.item() for l in losses], g["losses"], rtol=1e-4)
    names = [str(n) for n in g["grad_names"]]
    grads = {n: p.grad for n, p in model.named_parameters() if p.grad is not None}
    assert sorted(grads.keys()) == names
    for i, n in enumerate(names):
        st = grad_stats(grads[n].cpu())
        ref = g["grad_stats"][i]
        if n.endswith("w_ks.bias"):
            assert st[2] < 1e-4
            continue
        assert abs(st[2] - ref[2]) <= 2e-3 * ref[2] + 1e-6, (n, st, ref)
    for k in g.files:
        if k.startswith("grad:"):
            ref = torch.from_numpy(g[k])
            got = grads[k[5:]].cpu()
            assert (got - ref).abs().max().item() <= 2e-3 * ref.abs().max().item() + 1e-6, k
        if k.startswith("bn:"):
            got = dict(model.named_buffers())[k[3:]].cpu().numpy()
            assert np.allclose(got, g[k], rtol=1e-4, atol=1e-5), k


def test_train_grads_match_oracle_elementwise(dev):
    """every parameter gradient elementwise vs the oracle (fp64 oracle run) on a fresh seeded case."""
    pcfg, mcfg = configs.make(dropout=False, dec_layers=2, enc_layers=2)
    model = make_model(pcfg, mcfg, "fp32")
    sd = seeded_state_dict(model.state_dict(), 31)
    model.load_state_dict(sd)
    model.to(dev).train()
    model.disable_dropout = True
    b = synthetic_batch(32, 4, 40)
    out, losses = run_train(model, pcfg, mcfg, b, dev)
    oout, olosses, ograds, _ = oracle_train_case(pcfg, mcfg, sd, b, dtype=torch.float64)
    assert abs(losses[0].item() - olosses[0].item()) < 1e-4
    grads = {n: p.grad.cpu().double() for n, p in model.named_parameters() if p.grad is not None}
    for n, og in ograds.items():
        if n.endswith("w_ks.bias"):
            continue
        scale = og.abs().max().item()
        err = (grads[n] - og).abs().max().item()
        assert err <= 2e-3 * scale + 1e-7, (n, err, scale)


@pytest.mark.parametrize("tag,B,L,cfg", [("eval_lj", 3, 20, {}),
                                         ("eval_multi", 2, 12, dict(multi_speaker=True, dec_layers=2, enc_layers=2)),
                                         # L = 40 and T ~ 500 both beyond max_seq_len = 32: position tables regenerated on the fly
                                         ("eval_long", 2, 40, dict(dec_layers=2, enc_layers=2, max_seq_len=32))])
def test_inference_matches_reference_golden(dev, tag, B, L, cfg):
    g = load_golden(tag)
    pcfg, mcfg = configs.make(**cfg)
    model = make_model(pcfg, mcfg, "fp32")
    sd = seeded_state_dict(model.state_dict(), int(g["seed"]))
    sd["variance_adaptor.duration_predictor.linear_layer.bias"] = torch.tensor([1.4])
    model.load_state_dict(sd)
    model.to(dev).eval()
    b = to_dev(synthetic_batch(int(g["seed"]) + 1, B, L, n_speaker=4 if cfg.get("multi_speaker") else 1), dev)
    pc, ec, dc = [float(x) for x in g["controls"]]
    with torch.no_grad():
        out = model(b["speakers"], b["texts"], b["src_lens"], b["max_src_len"], p_control=pc, e_control=ec, d_control=dc)
    assert np.array_equal(out[5].cpu().numpy(), g["d_rounded"])          # bit-exact rounding contract
    assert np.array_equal(out[9].cpu().numpy(), g["mel_lens"])            # bit-exact LengthRegulator lengths
    assert np.array_equal(out[7].cpu().numpy(), g["mel_masks"])
    for name, idx in (("mel", 0), ("post", 1)):
        l1 = (out[idx].float().cpu() - torch.from_numpy(g[name])).abs().mean().item()
        assert l1 < MEL_L1_TOL and l1 < 2e-5, (name, l1)
    # e_control must be ignored (reference quirk, model/modules.py:124)
    with torch.no_grad():
        out2 = model(b["speakers"], b["texts"], b["src_lens"], b["max_src_len"], p_control=pc, e_control=ec * 3, d_control=dc)
    assert torch.equal(out[1], out2[1])


def test_bf16_train_step_close_to_oracle(dev):
    """bf16 storage / bf16 MFMA path: judged on closeness to the fp32 oracle (no bf16 reference exists)."""
    pcfg, mcfg = configs.make(dropout=False)
    model = make_model(pcfg, mcfg, "bf16")
    sd = seeded_state_dict(model.state_dict(), 41)
    model.load_state_dict(sd)
    model.to(dev).train()
    model.disable_dropout = True
    b = synthetic_batch(42, 4, 48)
    out, losses = run_train(model, pcfg, mcfg, b, dev)
    oout, olosses, ograds, _ = oracle_train_case(pcfg, mcfg, sd, b)
    assert abs(losses[0].item() - olosses[0].item()) < 0.05 * olosses[0].item()
    valid = ~out[7].cpu()
    l1 = ((out[1].float().cpu() - oout[1].detach()).abs() * valid.unsqueeze(-1)).sum() / (valid.sum() * 80)
    assert l1 < 0.06, l1
    gnorm = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in model.parameters() if p.grad is not None)).item()
    onorm = torch.sqrt(sum((g.double() ** 2).sum() for g in ograds.values())).item()
    assert abs(gnorm - onorm) < 0.1 * onorm, (gnorm, onorm)


def test_dropout_training_runs_and_optimizer_steps(dev):
    from fastspeech2_amd.model import ScheduledOptim
    pcfg, mcfg = configs.make()
    model = make_model(pcfg, mcfg, "fp32")
    model.load_state_dict(seeded_state_dict(model.state_dict(), 51))
    model.to(dev).train()
    b = synthetic_batch(52, 4, 32)
    model._ensure_flat(dev)
    opt = ScheduledOptim(model, configs.TRAIN, mcfg, 0)
    def probe():                                   # the objective without dropout noise (train-mode BatchNorm, no update)
        model.disable_dropout = True
        try:
            return run_train(model, pcfg, mcfg, b, dev)[1][0].item()
        finally:
            model.disable_dropout = False
            opt.zero_grad()
    first = probe()
    for step in range(8):
        out, losses = run_train(model, pcfg, mcfg, b, dev)
        assert torch.isfinite(losses[0]).item()
        opt.step_and_update_lr()
        opt.zero_grad()
    # the same fixed batch for 8 steps: the (dropout-free) objective must go down; the dropout-on loss itself is noisier than
    # 8 warm-up steps of learning rate ~1e-6 can move it
    assert probe() < first
    # two forward passes in train mode differ (fresh dropout masks); eval is deterministic
    o1 = run_train(model, pcfg, mcfg, b, dev)[0][1]
    o2 = run_train(model, pcfg, mcfg, b, dev)[0][1]
    assert not torch.equal(o1, o2)


@pytest.mark.parametrize("frame_level", [False, True])
def test_fused_loss_matches_oracle(dev, frame_level):
    """fs2_loss_fwd / fs2_loss_bwd vs the oracle's masked_select formulation (model/loss.py:19-92): all six values, and the
    gradients for a non-trivial combination of the outputs; targets longer than the predictions (cropped batches)."""
    from fastspeech2_amd.model import FastSpeech2Loss
    from oracle import fs2_oracle as O
    torch.manual_seed(11)
    pcfg, mcfg = configs.make(frame_level=frame_level)
    B, L, T, Tt = 5, 37, 211, 230
    src_lens = torch.tensor([37, 30, 22, 9, 1]); mel_lens = torch.tensor([211, 250, 120, 40, 3])     # one un-cropped length > T
    S = T if frame_level else L
    preds = [torch.randn(B, T, 80), torch.randn(B, T, 80), torch.randn(B, S), torch.randn(B, S), torch.randn(B, L)]
    mel_t = torch.randn(B, Tt, 80); p_t = torch.randn(B, Tt if frame_level else L); e_t = torch.randn(B, Tt if frame_level else L)
    if frame_level:
        p_t, e_t = p_t[:, :T], e_t[:, :T]                      # non-contiguous views with their own row stride
    dur = torch.randint(0, 9, (B, L))
    src_masks = torch.arange(L).unsqueeze(0) >= src_lens.unsqueeze(1)
    mel_masks = torch.arange(T).unsqueeze(0) >= mel_lens.unsqueeze(1)
    w = torch.tensor([1.0, 2.0, 0.0, 0.5, 0.0, 3.0])

    def run(device, loss_fn):
        ps = [p.clone().to(device).requires_grad_(True) for p in preds]
        out = (*ps, None, src_masks.to(device), mel_masks.to(device), src_lens.to(device), mel_lens.to(device))
        losses = loss_fn(out, mel_t.to(device), p_t.to(device), e_t.to(device), dur.to(device))
        sum(wi * li for wi, li in zip(w.tolist(), losses)).backward()
        return [l.detach().cpu() for l in losses], [p.grad.cpu() for p in ps]

    fused = FastSpeech2Loss(pcfg, mcfg)
    lv, gv = run(dev, lambda out, m, p, e, d: fused((None,) * 6 + (m, None, None, p, e, d), out))
    lo, go = run("cpu", lambda out, m, p, e, d: O.fastspeech2_loss(pcfg, (m, p, e, d), out))
    for a, b in zip(lv, lo):
        assert abs(a.item() - b.item()) < 2e-6 * max(1.0, abs(b.item()))
    for a, b in zip(gv, go):
        assert torch.allclose(a, b, atol=1e-8, rtol=1e-5)
