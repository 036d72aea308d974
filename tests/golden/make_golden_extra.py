"""More golden vectors from the live reference (run in the build container; needs /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_extra.py

  eval_long.npz   free-running inference with BOTH sequence lengths beyond max_seq_len (L=40, T~120 > 32): the encoder and the
                  decoder regenerate their sinusoid tables on the fly (transformer/Models.py:82-91,145-162), nothing is truncated
  optim.json      model/optimizer.py ScheduledOptim: learning rate after `_update_learning_rate()` at a list of steps (fresh and
                  restored runs), and the Adam hyper-parameters it hands to torch.optim.Adam
  bins.npz        VarianceAdaptor's quantisation boundaries (model/modules.py:41-78) for linear and log quantisation from a
                  stats.json with a positive pitch minimum (log of a non-positive minimum is NaN in the reference)
"""
import json
import os
import sys
import tempfile

import numpy as np

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402  (installs the unidecode / inflect stubs, sets sys.path)
from tests.golden import configs  # noqa: E402

BIN_STATS = {"pitch": [71.0, 795.8, 207.6, 46.8], "energy": [0.018, 315.0, 37.3, 26.0]}


def case_optim():
    from model import ScheduledOptim
    pcfg, mcfg = configs.make(dec_layers=1, enc_layers=1)
    model = MG.build_reference(pcfg, mcfg)
    steps = [1, 2, 100, 3999, 4000, 4001, 299999, 300000, 300001, 400000, 400001, 500001, 900000]
    lrs = {}
    for s in steps:
        opt = ScheduledOptim(model, configs.TRAIN, mcfg, s - 1)        # current_step = s - 1, then the update increments first
        opt._update_learning_rate()
        lrs[str(s)] = opt._optimizer.param_groups[0]["lr"]
    opt = ScheduledOptim(model, configs.TRAIN, mcfg, 0)
    g = opt._optimizer.param_groups[0]
    json.dump({"lr": lrs, "betas": list(g["betas"]), "eps": g["eps"], "weight_decay": g["weight_decay"], "init_lr": opt.init_lr},
              open(os.path.join(HERE, "optim.json"), "w"), indent=1)
    print("optim", lrs)


def case_bins():
    from model.modules import VarianceAdaptor
    out = {}
    with tempfile.TemporaryDirectory() as d:
        json.dump(BIN_STATS, open(os.path.join(d, "stats.json"), "w"))
        for kind in ("linear", "log"):
            pcfg, mcfg = configs.make()
            pcfg["path"]["preprocessed_path"] = d
            mcfg["variance_embedding"]["pitch_quantization"] = kind
            mcfg["variance_embedding"]["energy_quantization"] = kind
            va = VarianceAdaptor(pcfg, mcfg)
            out[kind + "_pitch"] = va.pitch_bins.detach().numpy()
            out[kind + "_energy"] = va.energy_bins.detach().numpy()
    np.savez_compressed(os.path.join(HERE, "bins.npz"), stats=json.dumps(BIN_STATS), **out)
    print("bins", {k: (v.shape, float(v[0]), float(v[-1])) for k, v in out.items()})


if __name__ == "__main__":
    os.chdir(MG.REF)
    MG.case_eval("eval_long", 808, B=2, L=40, controls=(1.0, 1.0, 1.0), dec_layers=2, enc_layers=2, max_seq_len=32)
    case_optim()
    case_bins()
