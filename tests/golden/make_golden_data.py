"""Golden fixtures for the data formats on the input side of the hot path, produced by the LIVE reference
(/root/reference; build container only):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_data.py

* symbols.json   — the reference's symbol inventory (text/symbols.py) and the id sequences its `text_to_sequence`
                   yields for the first phoneme strings of the shipped LJSpeech / AISHELL3 / LibriTTS val.txt.
* collate.npz    — the reference `Dataset(...).collate_fn` (sort=True, drop_last=True) and `TextDataset.collate_fn`
                   on a tiny synthetic preprocessed directory that tests regenerate from the same seed
                   (tests/helpers.py:make_preprocessed_dir).
Nothing is copied from the reference: it is imported and executed.
"""
import json
import os
import sys
import tempfile
import types

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)

import numpy as np

for name in ("unidecode", "inflect"):
    sys.modules[name] = types.ModuleType(name)
sys.modules["unidecode"].unidecode = lambda x: x
sys.modules["inflect"].engine = lambda: None

from tests.helpers import make_preprocessed_dir  # noqa: E402
from tests.golden import configs  # noqa: E402


def main():
    os.chdir(REF)
    from text import text_to_sequence
    from text.symbols import symbols
    cases = []
    for ds in ("LJSpeech", "AISHELL3", "LibriTTS"):
        path = os.path.join(REF, "preprocessed_data", ds, "val.txt")
        with open(path, encoding="utf-8") as f:
            for line in f.readlines()[:6]:
                phones = line.strip("\n").split("|")[2]
                cases.append({"dataset": ds, "phones": phones, "ids": text_to_sequence(phones, ["english_cleaners"])})
    with open(os.path.join(HERE, "symbols.json"), "w") as f:
        json.dump({"symbols": symbols, "cases": cases}, f, ensure_ascii=False)

    import dataset as ref_dataset
    with tempfile.TemporaryDirectory() as d:
        make_preprocessed_dir(d, seed=77, n_train=11, n_val=5)
        pcfg, _ = configs.make()
        pcfg["path"]["preprocessed_path"] = d
        tcfg = {"optimizer": {"batch_size": 4}}
        ds = ref_dataset.Dataset("train.txt", pcfg, tcfg, sort=True, drop_last=True)
        batches = ds.collate_fn([ds[i] for i in range(len(ds))])
        out = {"n_batches": np.array(len(batches))}
        for bi, b in enumerate(batches):
            out[f"b{bi}_ids"] = np.array(b[0])
            for j, name in zip(range(2, 12), ("speakers", "texts", "text_lens", "max_text_len", "mels", "mel_lens", "max_mel_len",
                                              "pitches", "energies", "durations")):
                out[f"b{bi}_{name}"] = np.asarray(b[j])
        tds = ref_dataset.TextDataset(os.path.join(d, "val.txt"), pcfg)
        tb = tds.collate_fn([tds[i] for i in range(len(tds))])
        out["t_ids"] = np.array(tb[0]); out["t_speakers"] = tb[2]; out["t_texts"] = tb[3]; out["t_lens"] = tb[4]
        out["t_max"] = np.asarray(tb[5])
        np.savez_compressed(os.path.join(HERE, "collate.npz"), **out)
    print("wrote symbols.json, collate.npz")


if __name__ == "__main__":
    main()
