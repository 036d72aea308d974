"""The hot path at BASELINE.json's FULL size (config 2: 4+4 layers, B=48, L=128, T~925): the oracle on the whole batch where it
finishes in seconds (one fp32 eval forward), and size-independent properties where it does not — LengthRegulator against an
integer restatement, padding invariance, batch-permutation equivariance, replay determinism, linearity of backward in the
upstream gradient, bf16 against fp32."""
import pytest
import torch

from oracle import fs2_oracle as O
from oracle.weights import seeded_state_dict, synthetic_batch
from tests.golden import configs
from tests.helpers import make_model
from tests.test_model_gpu import to_dev

pytestmark = pytest.mark.gpu
B, L = 48, 128


@pytest.fixture(scope="module")
def case(dev):
    pcfg, mcfg = configs.make(dec_layers=4, enc_layers=4, dropout=False)
    model = make_model(pcfg, mcfg, "fp32")
    sd = seeded_state_dict(model.state_dict(), 2024)
    model.load_state_dict(sd)
    model.to(dev).eval()
    b = synthetic_batch(1234, B, L, dur_lo=4, dur_hi=10, min_len_frac=0.75)
    assert b["max_mel_len"] > 850
    return pcfg, mcfg, model, sd, b


def fwd(model, d, rows=None, max_src_len=None, max_mel_len=None):
    r = slice(None) if rows is None else rows
    with torch.no_grad():
        return model(d["speakers"][r], d["texts"][r], d["src_lens"][r], max_src_len or d["max_src_len"], d["mels"][r], d["mel_lens"][r],
                     max_mel_len or d["max_mel_len"], d["pitches"][r], d["energies"][r], d["durations"][r])


def test_full_size_eval_forward_matches_oracle(dev, case):
    pcfg, mcfg, model, sd, b = case
    out = fwd(model, to_dev(b, dev))
    with torch.no_grad():
        ref = O.fastspeech2_forward(sd, mcfg, pcfg, b["speakers"], b["texts"], b["src_lens"], b["max_src_len"], b["mels"], b["mel_lens"],
                                    b["max_mel_len"], b["pitches"], b["energies"], b["durations"], training=False)
    assert torch.equal(out[9].cpu(), ref[9]) and torch.equal(out[7].cpu(), ref[7]) and torch.equal(out[5].cpu(), ref[5])
    for i in (0, 1):
        l1 = (out[i].float().cpu() - ref[i]).abs().mean().item()
        assert l1 < 1e-4, (i, l1)                                               # north_star bar (fp32); measured ~1e-6
    for i in (2, 3, 4):
        assert torch.allclose(out[i].float().cpu(), ref[i], atol=2e-4, rtol=1e-4), i


def test_full_size_length_regulator_is_the_integer_map(dev):
    from fastspeech2_amd import ops
    g = torch.Generator().manual_seed(3)
    dur = torch.randint(0, 11, (B, L), generator=g)
    dur[:, 100:] *= (torch.rand(B, 28, generator=g) > 0.5)
    x = torch.randn(B, L, 256, generator=g)
    T = 1000
    cum, idx, mel_len = ops.lr_index(dur.to(dev), T)
    out = ops.lr_gather_fwd(x.to(dev).view(B * L, 256), idx, None, B, L, T).view(B, T, 256).cpu()
    assert torch.equal(mel_len.cpu(), dur.sum(1))
    for bb in range(B):
        src = torch.repeat_interleave(torch.arange(L), dur[bb])[:T]             # frame -> phoneme, the reference's expand + cat
        assert torch.equal(out[bb, :len(src)], x[bb, src])                      # payload copied verbatim
        assert not out[bb, len(src):].any()                                     # zero padding (utils/tools.py:299-317)


def test_padding_invariance_and_permutation_equivariance(dev, case):
    """eval mode (BatchNorm on running statistics): utterances do not interact, padding never leaks into valid positions."""
    pcfg, mcfg, model, sd, b = case
    d = to_dev(b, dev)
    full = fwd(model, d)
    rows = slice(40, 48)                                                        # the 8 shortest utterances, on their own
    Ls, Ts = int(b["src_lens"][rows].max()), int(b["mel_lens"][rows].max())
    sub = dict(d)
    sub["texts"], sub["pitches"], sub["energies"], sub["durations"] = (d[k][:, :Ls] for k in ("texts", "pitches", "energies", "durations"))
    sub["mels"] = d["mels"][:, :Ts]
    small = fwd(model, sub, rows, max_src_len=Ls, max_mel_len=Ts)
    assert torch.equal(small[9], full[9][rows])
    # mel (before the PostNet) is padding-invariant everywhere.  The PostNet is not, in the reference either: it convolves the
    # UNMASKED mel, whose padded rows hold mel_linear's bias (model/fastspeech2.py:95-97), so the last 5 layers x 2 taps = 10
    # frames of an utterance see either bias rows or the conv's zero padding depending on the batch's max_mel_len.
    for i, tail in ((0, 0), (1, 10)):
        for k, r in enumerate(range(40, 48)):
            t = int(b["mel_lens"][r]) - tail
            assert torch.allclose(small[i][k, :t], full[i][r, :t], atol=2e-5, rtol=1e-5), (i, r)
    # Same for the variance predictors (model/modules.py:242-250): nothing is masked between their two k=3 convs, and the pitch
    # embedding is added at padded positions too (modules.py:113-118), so the last two valid phonemes' predictions see either
    # those values or the conv's zero padding.
    for i in (2, 3, 4):
        for k, r in enumerate(range(40, 48)):
            n = int(b["src_lens"][r]) - 2
            assert torch.allclose(small[i][k, :n], full[i][r, :n], atol=2e-5, rtol=1e-5), (i, r)
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(1)).to(dev)
    pd = {k: (v[perm] if isinstance(v, torch.Tensor) else v) for k, v in d.items()}
    p = fwd(model, pd)
    assert torch.equal(p[9], full[9][perm])
    assert torch.allclose(p[1], full[1][perm], atol=2e-5, rtol=1e-5)
    again = fwd(model, d)
    assert all(torch.equal(again[i], full[i]) for i in range(10))              # replay: bit-identical


def test_backward_is_linear_in_the_upstream_gradient(dev, case):
    """train mode without dropout at full size: gradients of 2 x loss = 2 x gradients of loss (fp32 atomics reorder sums, so
    to 1e-4 of the largest entry per tensor); BatchNorm batch statistics and every saved tensor are exercised at full size."""
    from fastspeech2_amd.model import FastSpeech2Loss
    pcfg, mcfg, model, sd, b = case
    d = to_dev(b, dev)
    model.train()
    model.disable_dropout = True                 # PostNet's p=0.5 is hard-coded (transformer/Layers.py:131-134); masks change per step
    try:
        loss_fn = FastSpeech2Loss(pcfg, mcfg)
        batch12 = (None, None, d["speakers"], d["texts"], d["src_lens"], d["max_src_len"], d["mels"], d["mel_lens"], d["max_mel_len"],
                   d["pitches"], d["energies"], d["durations"])
        grads = []
        model._ensure_flat(dev)                   # flat buffers are built lazily (first forward) when this test runs alone
        for scale in (1.0, 2.0):
            model.flat_gradients().zero_()
            out = model(*batch12[2:])
            (loss_fn(batch12, out)[0] * scale).backward()
            grads.append(model.flat_gradients().clone())
        assert torch.isfinite(grads[0]).all() and grads[0].abs().max() > 0
        gmax = grads[0].abs().max().item()
        for name, prm in model._trainable_in_backward_order():
            off, n = model._flat_offsets[name], prm.numel()
            g1, g2 = grads[0][off:off + n], grads[1][off:off + n]
            # per-tensor scale, plus a floor for tensors whose true gradient is 0 (conv biases in front of BatchNorm: rounding noise)
            tol = 2e-4 * g1.abs().max().item() + 1e-4 * gmax
            assert (g2 - 2 * g1).abs().max().item() <= tol, name
    finally:
        model.disable_dropout = False
        model.eval()
        model.load_state_dict(sd)                                               # BN running statistics moved in train mode


def test_bf16_full_size_close_to_fp32(dev, case):
    pcfg, mcfg, model, sd, b = case
    d = to_dev(b, dev)
    ref = fwd(model, d)
    m16 = make_model(pcfg, mcfg, "bf16")
    m16.load_state_dict(sd)
    m16.to(dev).eval()
    out = fwd(m16, d)
    assert torch.equal(out[9], ref[9])
    valid = (~ref[7]).unsqueeze(-1)
    for i in (0, 1):
        err = ((out[i].float() - ref[i]) * valid).abs().sum() / (valid.sum() * 80)
        scale = (ref[i] * valid).abs().sum() / (valid.sum() * 80)
        assert err < 0.03 * scale + 0.06, (i, err.item(), scale.item())
