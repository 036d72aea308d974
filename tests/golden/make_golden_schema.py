"""state_dict / parameter-order schema of the LIVE reference modules (build container only):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_schema.py  ->  tests/golden/state_schema.json

Checkpoints are the data format on the OUTPUT side of training (train.py:152-161: {"model": state_dict, "optimizer": Adam
state_dict whose per-parameter entries are indexed by `model.parameters()` ORDER), so key names, shapes, dtypes and the
parameter order are all part of the format."""
import json
import os
import sys
import types

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)
for name in ("unidecode", "inflect"):
    sys.modules[name] = types.ModuleType(name)
sys.modules["unidecode"].unidecode = lambda x: x
sys.modules["inflect"].engine = lambda: None

import torch  # noqa: E402
from tests.golden import configs  # noqa: E402


def schema(m):
    return {"state_dict": [[k, list(v.shape), str(v.dtype)] for k, v in m.state_dict().items()],
            "parameters": [[k, bool(p.requires_grad)] for k, p in m.named_parameters()]}


def main():
    os.chdir(REF)
    from model import FastSpeech2
    import hifigan
    out = {}
    for tag, kw in (("lj_4_6", dict(dec_layers=6)), ("lj_4_4", dict(dec_layers=4)), ("multi_4_4", dict(dec_layers=4, multi_speaker=True)),
                    ("frame_4_4", dict(dec_layers=4, frame_level=True))):
        pcfg, mcfg = configs.make(**kw)
        out[tag] = schema(FastSpeech2(pcfg, mcfg))
    h = hifigan.AttrDict(json.load(open(os.path.join(REF, "hifigan", "config.json"))))
    out["hifigan_v1"] = schema(hifigan.Generator(h))
    json.dump(out, open(os.path.join(HERE, "state_schema.json"), "w"))
    print({k: len(v["state_dict"]) for k, v in out.items()})


if __name__ == "__main__":
    main()
