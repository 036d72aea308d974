"""CPU: pin the oracle (oracle/fs2_oracle.py) against golden vectors produced by the LIVE reference
(tests/golden/make_golden.py), and pin the product module's state_dict schema against the reference's."""
import numpy as np
import pytest
import torch

from oracle import fs2_oracle as O
from oracle.weights import seeded_state_dict, synthetic_batch
from tests.golden import configs
from tests.helpers import grad_stats, load_golden, make_model, oracle_train_case

TRAIN_CASES = {
    "train_lj": dict(B=3, L=24, cfg={}),
    "train_multi_frame": dict(B=2, L=16, cfg=dict(multi_speaker=True, frame_level=True, dec_layers=2, enc_layers=2)),
    "train_trunc": dict(B=2, L=20, cfg=dict(dec_layers=1, enc_layers=1, max_seq_len=64), batch_max_seq_len=1000),
}


@pytest.mark.parametrize("tag", list(TRAIN_CASES))
def test_oracle_train_matches_reference(tag):
    c = TRAIN_CASES[tag]
    g = load_golden(tag)
    pcfg, mcfg = configs.make(dropout=False, **c["cfg"])
    model = make_model(pcfg, mcfg)
    tmpl = model.state_dict()
    # schema: same keys in the same order as the reference's state_dict (Appendix C)
    assert list(tmpl.keys()) == [str(k) for k in g["state_keys"]]
    sd = seeded_state_dict(tmpl, int(g["seed"]))
    nspk = 4 if c["cfg"].get("multi_speaker") else 1
    b = synthetic_batch(int(g["seed"]) + 1, c["B"], c["L"], n_speaker=nspk, frame_level=c["cfg"].get("frame_level", False),
                        max_seq_len=c.get("batch_max_seq_len") or mcfg["max_seq_len"])
    out, losses, grads, bn = oracle_train_case(pcfg, mcfg, sd, b)
    for name, idx in (("mel", 0), ("post", 1), ("p_pred", 2), ("e_pred", 3), ("logd", 4)):
        ref = torch.from_numpy(g[name])
        assert out[idx].shape == ref.shape
        assert (out[idx].detach() - ref).abs().mean().item() < 2e-6, name
    assert np.array_equal(out[9].numpy(), g["mel_lens"])                 # LengthRegulator lengths: bit-exact
    assert np.array_equal(out[7].numpy(), g["mel_masks"])
    assert np.allclose([l.item() for l in losses], g["losses"], rtol=2e-6)
    names = [str(n) for n in g["grad_names"]]
    assert sorted(grads.keys()) == names
    for i, n in enumerate(names):
        st = grad_stats(grads[n])
        ref = g["grad_stats"][i]
        if n.endswith("w_ks.bias"):         # mathematically zero gradient: pure rounding noise
            assert st[2] < 1e-5
            continue
        assert abs(st[2] - ref[2]) <= 2e-4 * ref[2] + 1e-7, (n, st, ref)
    for k in g.files:
        if k.startswith("grad:"):
            ref = torch.from_numpy(g[k])
            assert (grads[k[5:]] - ref).abs().max().item() <= 2e-4 * ref.abs().max().item() + 1e-7, k
        if k.startswith("bn:"):
            assert np.allclose(bn[k[3:]].numpy(), g[k], rtol=1e-5, atol=1e-6), k


@pytest.mark.parametrize("tag,B,L,cfg", [("eval_lj", 3, 20, {}),
                                         ("eval_multi", 2, 12, dict(multi_speaker=True, dec_layers=2, enc_layers=2)),
                                         # L = 40 and T ~ 500 both beyond max_seq_len = 32: position tables regenerated on the fly
                                         ("eval_long", 2, 40, dict(dec_layers=2, enc_layers=2, max_seq_len=32))])
def test_oracle_eval_matches_reference(tag, B, L, cfg):
    g = load_golden(tag)
    pcfg, mcfg = configs.make(**cfg)
    sd = seeded_state_dict(make_model(pcfg, mcfg).state_dict(), int(g["seed"]))
    sd["variance_adaptor.duration_predictor.linear_layer.bias"] = torch.tensor([1.4])
    b = synthetic_batch(int(g["seed"]) + 1, B, L, n_speaker=4 if cfg.get("multi_speaker") else 1)
    pc, ec, dc = [float(x) for x in g["controls"]]
    with torch.no_grad():
        out = O.fastspeech2_forward(sd, mcfg, pcfg, b["speakers"], b["texts"], b["src_lens"], b["max_src_len"],
                                    p_control=pc, e_control=ec, d_control=dc, training=False)
    assert np.array_equal(out[5].numpy(), g["d_rounded"])                # rounding contract: bit-exact
    assert np.array_equal(out[9].numpy(), g["mel_lens"])
    assert (out[0] - torch.from_numpy(g["mel"])).abs().mean().item() < 2e-6
    assert (out[1] - torch.from_numpy(g["post"])).abs().mean().item() < 2e-6


def test_oracle_length_regulator_matches_reference():
    g = load_golden("length_regulator")
    x, d = torch.from_numpy(g["x"]), torch.from_numpy(g["d"])
    for tag, max_len in (("none", None), ("crop", 10), ("pad", 30)):
        out, ln = O.length_regulate(x, d, max_len)
        assert np.array_equal(out.numpy(), g[f"out_{tag}"])
        assert np.array_equal(ln.numpy(), g[f"len_{tag}"])


def test_oracle_hifigan_matches_reference():
    g = load_golden("hifigan")
    sd = hifigan_weights(int(g["seed"]), [str(k) for k in g["keys"]])
    wav = O.hifigan_forward(O.remove_weight_norm_sd(sd), configs.HIFIGAN, torch.from_numpy(g["mel"]))
    assert (wav - torch.from_numpy(g["wav"])).abs().max().item() < 2e-5


def hifigan_weights(seed, keys, gain=1.2):
    """same recipe as tests/golden/make_golden.py:case_hifigan (shapes from the HiFi-GAN v1 config)."""
    shapes = hifigan_shapes()
    assert sorted(shapes.keys()) == keys
    gen = torch.Generator().manual_seed(seed)
    new = {}
    for k in sorted(shapes.keys()):
        shp = shapes[k]
        if k.endswith("weight_v"):
            fan_in = shp[1] * shp[2] if "ups" not in k else shp[0] * shp[2]
            new[k] = torch.randn(shp, generator=gen) * (gain / fan_in ** 0.5)
        elif k.endswith("weight_g"):
            new[k] = None
        else:
            new[k] = torch.randn(shp, generator=gen) * 0.05
    for k in list(new.keys()):
        if k.endswith("weight_g"):
            v = new[k[:-1] + "v"]
            norm = v.reshape(v.shape[0], -1).norm(dim=1).view(shapes[k])
            new[k] = norm * (0.8 + 0.4 * torch.rand(shapes[k], generator=gen))
    return new


def test_hifigan_stored_oracle_is_the_oracle_when_nothing_is_rounded():
    """oracle.hifigan_forward_stored (the product's storage points made explicit: the form the bf16 bars are derived from) with
    no rounding equals hifigan_forward (pinned by the reference golden) up to fp64 summation order; the stage probes and the
    per-stage input substitution are consistent; with bf16 storage it moves by a bf16-sized amount."""
    keys = sorted(hifigan_shapes().keys())
    sd = O.remove_weight_norm_sd({k: v.double() for k, v in hifigan_weights(5, keys).items()})
    mel = torch.clamp(torch.randn(2, 80, 9, generator=torch.Generator().manual_seed(3)) * 2 - 5, -11.5, 2.0).double()
    with torch.no_grad():
        ref = O.hifigan_forward(sd, configs.HIFIGAN, mel)
        stages = []
        got = O.hifigan_forward_stored(sd, configs.HIFIGAN, mel, stages=stages)
        assert (got - ref).abs().max().item() < 1e-12 and len(stages) == 5
        assert [tuple(t.shape[1:]) for t in stages] == [(512, 9), (256, 72), (128, 576), (64, 1152), (32, 2304)]
        # feeding every stage its own recorded input reproduces the run
        again = O.hifigan_forward_stored(sd, configs.HIFIGAN, mel, stage_inputs=[None] + stages)
        assert (again - got).abs().max().item() < 1e-12
        emu = O.hifigan_forward_stored(sd, configs.HIFIGAN, mel, store=O.bf16_store, weight_store=O.bf16_store)
    rel = float((emu - ref).norm() / ref.norm())
    assert 1e-4 < rel < 5e-2, rel


def hifigan_shapes(h=configs.HIFIGAN):
    """state_dict schema of hifigan.Generator (reference hifigan/models.py:113-147) with weight-norm keys."""
    s = {}

    def conv(name, cout, cin, k, transposed=False):
        s[name + ".bias"] = (cout,)
        s[name + ".weight_v"] = (cin, cout, k) if transposed else (cout, cin, k)
        s[name + ".weight_g"] = (cin, 1, 1) if transposed else (cout, 1, 1)

    c0 = h["upsample_initial_channel"]
    conv("conv_pre", c0, 80, 7)
    nk = len(h["resblock_kernel_sizes"])
    ch = c0
    for i, (u, k) in enumerate(zip(h["upsample_rates"], h["upsample_kernel_sizes"])):
        conv(f"ups.{i}", c0 // 2 ** (i + 1), c0 // 2 ** i, k, transposed=True)
        ch = c0 // 2 ** (i + 1)
        for j, rk in enumerate(h["resblock_kernel_sizes"]):
            for m in range(3):
                conv(f"resblocks.{i * nk + j}.convs1.{m}", ch, ch, rk)
                conv(f"resblocks.{i * nk + j}.convs2.{m}", ch, ch, rk)
    conv("conv_post", 1, ch, 7)
    return s


def test_oracle_stft_matches_reference():
    """audio/stft.py run live (tests/golden/make_golden.py:case_stft) vs the oracle's restatement.  The mel
    filterbank in BOTH is the restated Slaney formula (librosa absent): that boundary is parity-unpinned."""
    g = load_golden("stft")
    y = torch.from_numpy(g["y"])
    mel, energy = O.mel_spectrogram(y)
    assert mel.shape == g["mel"].shape and energy.shape == g["energy"].shape
    assert (mel - torch.from_numpy(g["mel"])).abs().max().item() < 1e-4
    assert np.allclose(energy.numpy(), g["energy"], rtol=1e-5, atol=1e-5)
    basis = O.stft_basis(1024, 1024)
    assert np.array_equal(basis[::101].numpy(), g["forward_basis_rows"])       # windowed DFT basis: bit-exact


def test_oracle_pcm16_matches_numpy_cast():
    wav = torch.tensor([0.0, 0.5, -0.5, 0.99999, -1.0, 1.0 / 32768.0 * 2.9, -1.0 / 32768.0 * 2.9])
    assert O.pcm16(wav).tolist() == [0, 16384, -16384, 32767, -32768, 2, -2]
