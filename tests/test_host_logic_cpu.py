"""CPU: host-side logic above the C ABI that needs no kernel launch — LR schedule, flat parameter layout,
checkpoint (state_dict) round trip, tap-major weight views."""
import pytest
import torch

from oracle import fs2_oracle as O
from tests.golden import configs
from tests.helpers import make_model
from fastspeech2_amd.model import ScheduledOptim


def test_lr_schedule_matches_oracle():
    """reference model/optimizer.py:33-51: current_step increments before the lr is computed."""
    pcfg, mcfg = configs.make(dec_layers=1, enc_layers=1)
    opt = ScheduledOptim(make_model(pcfg, mcfg), configs.TRAIN, mcfg, 0)
    oc = configs.TRAIN["optimizer"]
    for step in (1, 2, 100, 3999, 4000, 4001, 299999, 300000, 300001, 400001, 500001, 900000):
        opt.current_step = step
        got = opt.init_lr * opt._get_lr_scale()
        want = O.lr_at_step(step, 256, oc["warm_up_step"], oc["anneal_steps"], oc["anneal_rate"])
        assert abs(got - want) <= 1e-12 + 1e-9 * want, step


def test_lr_schedule_and_adam_hyperparameters_match_reference_golden():
    """tests/golden/optim.json: the reference's own ScheduledOptim (model/optimizer.py) after `_update_learning_rate()`."""
    import json
    import os
    g = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "optim.json")))
    pcfg, mcfg = configs.make(dec_layers=1, enc_layers=1)
    oc = configs.TRAIN["optimizer"]
    for step, want in g["lr"].items():
        step = int(step)
        opt = ScheduledOptim(make_model(pcfg, mcfg), configs.TRAIN, mcfg, step - 1)     # restored at step - 1, as get_model does
        opt.current_step += 1                                                            # what _update_learning_rate does first
        got = opt.init_lr * opt._get_lr_scale()
        assert abs(got - want) <= 1e-12 * want + 1e-18, (step, got, want)
        assert abs(O.lr_at_step(step, 256, oc["warm_up_step"], oc["anneal_steps"], oc["anneal_rate"]) - want) <= 1e-12 * want + 1e-18
    opt = ScheduledOptim(make_model(pcfg, mcfg), configs.TRAIN, mcfg, 0)
    assert list(opt.betas) == g["betas"] and opt.eps == g["eps"] and opt.weight_decay == g["weight_decay"]
    assert abs(opt.init_lr - g["init_lr"]) <= 1e-15


@pytest.mark.parametrize("kind", ["linear", "log"])
def test_quantisation_boundaries_match_reference_golden(tmp_path, kind):
    """tests/golden/bins.npz: VarianceAdaptor's pitch / energy bins (model/modules.py:41-78) from the reference's constructor."""
    import json
    import numpy as np
    from tests.helpers import load_golden
    g = load_golden("bins")
    json.dump(json.loads(str(g["stats"])), open(tmp_path / "stats.json", "w"))
    pcfg, mcfg = configs.make(dec_layers=1, enc_layers=1)
    pcfg["path"]["preprocessed_path"] = str(tmp_path)
    mcfg["variance_embedding"]["pitch_quantization"] = kind
    mcfg["variance_embedding"]["energy_quantization"] = kind
    m = make_model(pcfg, mcfg)
    sd = m.state_dict()
    assert np.array_equal(sd["variance_adaptor.pitch_bins"].numpy(), g[kind + "_pitch"])       # same torch ops: bit-identical
    assert np.array_equal(sd["variance_adaptor.energy_bins"].numpy(), g[kind + "_energy"])


def test_flat_layout_aliases_every_trainable_parameter():
    pcfg, mcfg = configs.make(dec_layers=2, enc_layers=1, multi_speaker=True)
    m = make_model(pcfg, mcfg)
    before = {k: v.clone() for k, v in m.state_dict().items()}
    m._ensure_flat(torch.device("cpu"))
    after = m.state_dict()
    assert list(before) == list(after)
    for k in before:
        assert torch.equal(before[k], after[k]), k
    flat = m.flat_parameters()
    lo, hi = flat.data_ptr(), flat.data_ptr() + flat.numel() * 4
    n = 0
    for name, p in m.named_parameters():
        if p.requires_grad:
            assert lo <= p.data_ptr() < hi, name
            n += p.numel()
    assert n <= flat.numel() < n + 4 * len(list(m.parameters()))
    # conv weights are stored tap-major: the (Cout, Cin, k) parameter is a permuted view of [Cout][k][Cin]
    w = m.encoder.layer_stack[0].pos_ffn.w_1.weight
    assert w.shape == (1024, 256, 9) and w.stride() == (9 * 256, 1, 256)
    # QKV weights are adjacent so the fused [3d, d] projection is one matrix
    a = m.decoder.layer_stack[1].slf_attn
    assert a.w_ks.weight.data_ptr() == a.w_qs.weight.data_ptr() + 256 * 256 * 4
    assert a.w_vs.weight.data_ptr() == a.w_ks.weight.data_ptr() + 256 * 256 * 4
    # writing through a parameter is visible in the flat buffer (optimizer/engine see the same memory)
    with torch.no_grad():
        m.mel_linear.bias.fill_(3.5)
    o = m._flat_offsets["mel_linear.bias"]
    assert torch.all(flat[o:o + 80] == 3.5)


def test_checkpoint_round_trip(tmp_path):
    """train.py:152-161 format {"model": sd, "optimizer": sd}: model keys load into a fresh module unchanged."""
    pcfg, mcfg = configs.make(dec_layers=1, enc_layers=1)
    m1 = make_model(pcfg, mcfg)
    m1._ensure_flat(torch.device("cpu"))
    path = tmp_path / "10.pth.tar"
    torch.save({"model": m1.state_dict()}, path)
    m2 = make_model(pcfg, mcfg)
    m2._ensure_flat(torch.device("cpu"))
    m2.load_state_dict(torch.load(path)["model"])
    for (k1, v1), (k2, v2) in zip(m1.state_dict().items(), m2.state_dict().items()):
        assert k1 == k2 and torch.equal(v1, v2), k1
    # loading must keep the parameters aliased to the flat buffer
    flat = m2.flat_parameters()
    assert flat.data_ptr() <= m2.mel_linear.weight.data_ptr() < flat.data_ptr() + flat.numel() * 4


@pytest.mark.parametrize("tag,kw", [("lj_4_6", dict(dec_layers=6)), ("lj_4_4", dict(dec_layers=4)),
                                    ("multi_4_4", dict(dec_layers=4, multi_speaker=True)), ("frame_4_4", dict(dec_layers=4, frame_level=True))])
def test_state_dict_schema_matches_reference(tag, kw):
    """checkpoint format: key names, shapes, dtypes of state_dict() AND the order of parameters() (torch.optim.Adam's
    state is indexed by it) equal the live reference's (tests/golden/make_golden_schema.py)."""
    import json
    import os
    from tests.golden import configs
    from fastspeech2_amd.model import FastSpeech2
    g = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "state_schema.json")))[tag]
    m = FastSpeech2(*configs.make(**kw))
    assert [[k, list(v.shape), str(v.dtype)] for k, v in m.state_dict().items()] == g["state_dict"]
    assert [[k, bool(p.requires_grad)] for k, p in m.named_parameters()] == g["parameters"]


def test_hifigan_schema_matches_reference():
    import json
    import os
    from fastspeech2_amd import hifigan, utils
    g = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "state_schema.json")))["hifigan_v1"]
    gen = hifigan.Generator(hifigan.AttrDict(utils.HIFIGAN_V1))
    # (key order inside a weight-normed layer differs - bias / weight_g / weight_v - which load_state_dict does not care about)
    assert sorted([k, list(v.shape), str(v.dtype)] for k, v in gen.state_dict().items()) == sorted(g["state_dict"])


def test_mel_filterbank_equals_the_independent_decimal_computation():
    """a16 (VERDICT r02 missing 5): the 80 x 513 Slaney table comes from librosa==0.7.2 `filters.mel` (audio/stft.py:145-147),
    absent here.  The product's and the oracle's restatements are compared BIT FOR BIT with a third, independent computation
    (tests/golden/make_mel_checksum.py: scalar 60-digit decimal arithmetic, librosa's float32 rounding order), committed as a
    sha256 of the float32 bytes plus samples."""
    import hashlib
    import json
    import os
    import struct

    import numpy as np

    from fastspeech2_amd.audio import slaney_mel_filterbank
    from oracle import fs2_oracle as O

    g = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "mel_filterbank.json")))
    for name, t in (("product", slaney_mel_filterbank(22050, 1024, 80, 0, 8000)), ("oracle", O.slaney_mel_filterbank(22050, 1024, 80, 0, 8000))):
        assert t.dtype == np.float32 and list(t.shape) == g["shape"], name
        assert hashlib.sha256((t + np.float32(0)).tobytes()).hexdigest() == g["sha256_float32_le"], name
        assert int((t != 0).sum()) == g["nonzero"]
        for key, bits in g["samples"].items():
            i, k = map(int, key.split(","))
            assert struct.unpack("<I", struct.pack("<f", float(t[i, k])))[0] == bits, (name, key)
        assert [int(np.nonzero(r)[0][0]) for r in t] == g["first_nonzero"] and [int(np.nonzero(r)[0][-1]) for r in t] == g["last_nonzero"]


def test_padded_tile_fraction_from_the_host_copy_of_the_lengths():
    """Engine._skip_fraction (the per-batch decision whether the FFT blocks' contractions carry lens) == the rule fs2_tile_map
    applies on the device: a 256-row tile of the [B * S] row space is padded when all of its rows lie in ONE sequence's tail;
    utils.lens_to_device hands the engine the host copy it needs."""
    import numpy as np
    import torch
    from fastspeech2_amd.engine import Engine
    from fastspeech2_amd.utils import lens_to_device

    def brute(lens, S, rows=256):
        M = len(lens) * S
        n = pad = 0
        for m0 in range(0, M, rows):
            ms = range(m0, min(m0 + rows, M))
            n += 1
            pad += all((m % S) >= min(lens[m // S], S) for m in ms) and len({m // S for m in ms}) == 1
        return pad / n

    rng = np.random.RandomState(5)
    for B, S in ((48, 925), (48, 988), (3, 100), (7, 256), (5, 1000)):
        for lo in (0.05, 0.5, 0.9):
            lens = rng.randint(int(S * lo), S + 1, size=B)
            assert abs(Engine._skip_fraction(lens, S) - brute(lens.tolist(), S)) < 1e-12, (B, S, lo)
    assert Engine._skip_fraction(None, 925) == 0.0
    assert Engine._skip_fraction(np.full(48, 925), 925) == 0.0
    t = lens_to_device(np.array([5, 3, 9], dtype=np.int64), torch.device("cpu"))
    assert t.dtype == torch.int64 and t.tolist() == [5, 3, 9] and t._fs2_host.tolist() == [5, 3, 9]


def test_committed_pmc_traffic_file_was_measured_on_the_current_contraction_sources():
    """bench.py's roofline.traffic comes from the newest profiles/*pmc_traffic.json (counters cannot be read inside the timed run);
    the file records a hash of the contraction-kernel sources it was measured on and bench.py reports null on a mismatch.  This
    test makes the mismatch visible HERE, on the CPU: a commit that edits csrc/fs2_gemm* (or fs2_sched.h) without re-running
    tools/gpu_milestone.sh turns it red before the GPU suite's bench-contract test does."""
    import glob
    import json
    import os
    from fastspeech2_amd._lib import kernel_source_sha
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cands = sorted(glob.glob(os.path.join(root, "profiles", "*pmc_traffic.json")))
    assert cands, "no PMC traffic file committed under profiles/"
    doc = json.load(open(cands[-1]))
    assert doc.get("kernel_source_sha") == kernel_source_sha(), (
        f"{os.path.basename(cands[-1])} was measured on other contraction-kernel sources: re-run tools/gpu_milestone.sh and commit "
        f"its *_pmc_traffic.json")
    # and the file names the kernel the roofline line is about
    assert any(k.startswith("conv_gemm_p_kernel<false") for k in doc["kernels"])


def test_sample_lengths_use_the_host_copy_when_the_engine_left_one():
    """utils._sample_lengths (utils/tools.py:176: frames x hop_length): from the `_fs2_host` vector the engine's own length round trip
    leaves on the returned mel_lens, else from the tensor - the same numbers either way."""
    import numpy as np
    import torch
    from fastspeech2_amd.utils import _sample_lengths
    pcfg = {"preprocessing": {"stft": {"hop_length": 256}}}
    t = torch.tensor([7, 0, 925], dtype=torch.int64)
    assert _sample_lengths(t, pcfg) == [1792, 0, 236800]
    t._fs2_host = np.array([7, 0, 925], dtype=np.int64)
    out = _sample_lengths(t, pcfg)
    assert out == [1792, 0, 236800] and all(type(n) is int for n in out)


def test_hardware_queue_setting_is_an_explicit_decision_not_an_import_side_effect():
    """VERDICT r05 weak 8 / ADVICE r05 medium: importing the package must not rewrite os.environ; the entry points that own their
    process call configure_hw_queues() - the same value at every world size, an exported value wins, ranks started by a launcher
    that made the decision inherit it AS this package's decision (the bench line's config.hw_queues says which)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import os, json, sys; sys.path.insert(0, %r)\n"
        "before = os.environ.get('GPU_MAX_HW_QUEUES')\n"
        "import fastspeech2_amd\n"
        "after_import = os.environ.get('GPU_MAX_HW_QUEUES')\n"
        "rec = fastspeech2_amd.configure_hw_queues(int(os.environ.get('N', '16')))\n"
        "print(json.dumps([before, after_import, os.environ.get('GPU_MAX_HW_QUEUES'), rec]))\n" % root)
    base = {k: v for k, v in os.environ.items() if k not in ("GPU_MAX_HW_QUEUES", "FASTSPEECH2_AMD_HW_QUEUES", "WORLD_SIZE")}

    def run(**env):
        out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(base, **env), timeout=120)
        assert out.returncode == 0, out.stderr[-1500:]
        return json.loads(out.stdout.strip().split("\n")[-1])

    for world in ("1", "8"):                                   # the SAME decision whatever the world size
        before, after_import, after_cfg, rec = run(WORLD_SIZE=world)
        assert before is None and after_import is None and after_cfg == "16"
        assert rec == {"value": 16, "source": "fastspeech2_amd"}
    assert run(GPU_MAX_HW_QUEUES="8")[2:] == ["8", {"value": 8, "source": "user"}]                       # an exported value wins
    assert run(GPU_MAX_HW_QUEUES="16", FASTSPEECH2_AMD_HW_QUEUES="16")[3] == {"value": 16, "source": "fastspeech2_amd"}   # a launched rank
    assert run(N="0")[2:] == [None, {"value": None, "source": "runtime default"}]                       # 0 = leave the runtime default
