"""End-to-end drop-in check of the entry points around the hot path (GPU): train.py for a few steps on a tiny synthetic
preprocessed directory (checkpoint in the reference's format), restore + evaluate.py, synthesize.py --mode batch / single
with the HiFi-GAN kernels.  The YAML files are written in the reference's schema and parsed with the scripts' own code."""
import copy
import os

import numpy as np
import pytest
import torch
import yaml

from tests.golden import configs
from tests.helpers import make_preprocessed_dir

pytestmark = pytest.mark.gpu


def _write_configs(root, dec_layers=1, enc_layers=1, batch=4, total_step=6):
    data = make_preprocessed_dir(os.path.join(root, "data"), seed=31, n_train=24, n_val=6, lo=6, hi=20)
    pcfg, mcfg = configs.make(dec_layers=dec_layers, enc_layers=enc_layers)
    pcfg["path"]["preprocessed_path"] = data
    tcfg = copy.deepcopy(configs.TRAIN)
    tcfg["path"] = {k: os.path.join(root, "out", k.split("_")[0]) for k in ("ckpt_path", "log_path", "result_path")}
    tcfg["optimizer"]["batch_size"] = batch
    tcfg["step"] = {"total_step": total_step, "log_step": 2, "synth_step": 100, "val_step": 3, "save_step": 3}
    paths = []
    for name, cfg in (("preprocess.yaml", pcfg), ("model.yaml", mcfg), ("train.yaml", tcfg)):
        p = os.path.join(root, name)
        with open(p, "w") as f:
            yaml.safe_dump(cfg, f)
        paths.append(p)
    return paths, tcfg


def test_train_evaluate_synthesize_cli(dev, tmp_path):
    import evaluate as evaluate_cli
    import synthesize as synth_cli
    import train as train_cli
    (pp, mp, tp), tcfg = _write_configs(str(tmp_path))
    args = train_cli.parse_args(["-p", pp, "-m", mp, "-t", tp, "--no_vocoder"])
    cfgs = tuple(yaml.load(open(p), Loader=yaml.FullLoader) for p in (pp, mp, tp))
    torch.manual_seed(0)
    model, opt = train_cli.main(args, cfgs)
    ck3, ck6 = (os.path.join(tcfg["path"]["ckpt_path"], f"{s}.pth.tar") for s in (3, 6))
    assert os.path.exists(ck3) and os.path.exists(ck6)
    log = open(os.path.join(tcfg["path"]["log_path"], "train", "log.txt")).read().strip().split("\n")
    assert len(log) == 3 and log[0].startswith("Step 2/6, Total Loss: ")
    vlog = open(os.path.join(tcfg["path"]["log_path"], "val", "log.txt")).read()
    assert vlog.startswith("Validation Step 3, Total Loss: ")
    # checkpoint: reference schema ({"model", "optimizer"}), compact contiguous tensors, Adam state for every trainable tensor
    ck = torch.load(ck6, map_location="cpu")
    assert set(ck) == {"model", "optimizer"}
    w = ck["model"]["decoder.layer_stack.0.pos_ffn.w_1.weight"]
    assert w.shape == (1024, 256, 9) and w.is_contiguous() and w.untyped_storage().nbytes() == w.numel() * 4
    n_trainable = sum(1 for p in model.parameters() if p.requires_grad)
    assert len(ck["optimizer"]["state"]) == n_trainable and ck["optimizer"]["param_groups"][0]["betas"] == (0.9, 0.98)
    live = model.state_dict()
    for k, v in ck["model"].items():
        assert torch.equal(v, live[k].cpu()), k
    # restore + continue: the restored run reproduces the live model's next evaluation exactly
    eargs = evaluate_cli.parse_args(["--restore_step", "6", "-p", pp, "-m", mp, "-t", tp])
    from fastspeech2_amd.utils import get_model
    m2 = get_model(eargs, cfgs, dev, train=False)
    msg_restored = evaluate_cli.evaluate(m2, 6, cfgs, device=dev)
    model.eval()
    msg_live = evaluate_cli.evaluate(model, 6, cfgs, device=dev)
    assert msg_restored == msg_live and "nan" not in msg_live.lower()
    m3, o3 = get_model(train_cli.parse_args(["--restore_step", "6", "-p", pp, "-m", mp, "-t", tp]), cfgs, dev, train=True)
    m3._ensure_flat(dev); o3._ensure()
    assert o3.current_step == 6 and o3._adam_step == 6
    assert torch.equal(o3._m, opt._m) and torch.equal(o3._v, opt._v)
    # batch synthesis (random-init vocoder: the released generator checkpoint is a download)
    src = os.path.join(cfgs[0]["path"]["preprocessed_path"], "val.txt")
    sargs = synth_cli.parse_args(["--restore_step", "6", "--mode", "batch", "--source", src, "-p", pp, "-m", mp, "-t", tp,
                                  "--random_vocoder", "--batch_size", "4", "--duration_control", "1.2"])
    synth_cli.main(sargs)
    from scipy.io import wavfile
    names = [l.split("|")[0] for l in open(src).read().strip().split("\n")]
    for n in names:
        sr, wav = wavfile.read(os.path.join(tcfg["path"]["result_path"], n + ".wav"))
        assert sr == 22050 and wav.dtype == np.int16 and len(wav) % 256 == 0
    sargs = synth_cli.parse_args(["--restore_step", "6", "--mode", "single", "--text", "{HH AH0 L OW1 sp W ER1 L D}", "-p", pp,
                                  "-m", mp, "-t", tp, "--random_vocoder"])
    synth_cli.main(sargs)
    assert os.path.exists(os.path.join(tcfg["path"]["result_path"], "HH_AH0_L_OW1_sp_W_ER1_L_D.wav"))
    with pytest.raises(SystemExit):
        synth_cli.single_batch(synth_cli.parse_args(["--restore_step", "6", "--mode", "single", "--text", "hello", "-p", pp, "-m", mp,
                                                     "-t", tp]), cfgs[0])
