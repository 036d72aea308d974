"""Derives the bars of tests/test_a_prodshape_gpu.py::test_full_size_train_step_bf16_per_tensor_budget from bf16 ITSELF
(the round-2 bars were moved four times to sit above the newest measurement).

For each of N_SEEDS seeded (weights, batch) pairs at the bench's full size (B = 48, L = 128, T ~ 925, 4 + 4 layers, dropout
off) the fp64 oracle is run twice on the network the bf16 engine differentiates (matrices rounded to bf16):
  exact    - no rounding anywhere;
  emulated - `oracle.storage(round_st_bf16)`: every activation the product stores between two kernels is rounded to bf16 and the
             gradient flowing back through that edge is rounded the same way (oracle/fs2_oracle.py marks the points).
The emulation's own distance to the exact run - per parameter tensor (relative Frobenius), per output (valid-frame L1), per
loss - is what bf16 storage costs by construction; the committed bar of every quantity is 2 x its maximum over the seeds (plus
nothing else), and the product's measured distance must stay below it.  The table is tests/golden/bf16_bars.json; the test only
reads it.  Run time: ~2 min per seed on 8 cores.

    python tests/golden/make_bf16_bars.py [n_seeds]
"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import fs2_oracle as O  # noqa: E402
from oracle.weights import seeded_state_dict, synthetic_batch  # noqa: E402
from tests.golden import configs  # noqa: E402
from tests.helpers import bf16_matrix, make_model, oracle_train_case  # noqa: E402

B, L = 48, 128
FACTOR = 2.0


def one(seed):
    pcfg, mcfg = configs.make(dec_layers=4, enc_layers=4, dropout=False)
    model = make_model(pcfg, mcfg, "fp32")
    sd = seeded_state_dict(model.state_dict(), 2025 + 17 * seed)
    b = synthetic_batch(1234 + seed, B, L, dur_lo=4, dur_hi=10, min_len_frac=0.75)
    sdr = {k: (v.to(torch.bfloat16).to(v.dtype) if bf16_matrix(k, v) else v) for k, v in sd.items()}
    eo, el, eg, _ = oracle_train_case(pcfg, mcfg, sdr, b, dtype=torch.float64)
    with O.storage(O.round_st_bf16):
        mo, ml, mg, _ = oracle_train_case(pcfg, mcfg, sdr, b, dtype=torch.float64)
    valid = (~eo[7]).unsqueeze(-1)
    nval = valid.sum().item() * 80
    res = {"mel_l1": [((mo[i].detach() - eo[i].detach()).abs() * valid).sum().item() / nval for i in (0, 1)],
           "loss_rel": [abs(a.item() - o.item()) / max(1.0, abs(o.item())) for a, o in zip(ml, el)], "grad": {}}
    gmax = max(g.abs().max().item() for g in eg.values())
    for n, g in eg.items():
        if g.abs().max().item() < 1e-9 * gmax:
            continue                                        # true gradient zero: judged by an absolute bound in the test
        res["grad"][n] = ((mg[n] - g).norm() / g.norm()).item()
    return res


def main():
    n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    runs = []
    for s in range(n_seeds):
        t0 = time.time()
        runs.append(one(s))
        g = sorted(runs[-1]["grad"].values())
        print(f"seed {s}: {time.time() - t0:.0f} s  mel L1 {runs[-1]['mel_l1']}  grad rel-Frobenius median {g[len(g) // 2]:.2e} max {g[-1]:.2e}", flush=True)
    names = sorted(runs[0]["grad"])
    table = {"what": "bf16 storage emulation vs exact, fp64 oracle with bf16-rounded matrices, B=48 L=128 4+4 layers, dropout off",
             "n_seeds": n_seeds, "factor": FACTOR,
             "mel_l1": {"emulated_max": [max(r["mel_l1"][i] for r in runs) for i in (0, 1)]},
             "loss_rel": {"emulated_max": [max(r["loss_rel"][i] for r in runs) for i in range(6)]},
             "grad": {n: {"emulated_max": max(r["grad"].get(n, 0.0) for r in runs),
                          "emulated_mean": sum(r["grad"].get(n, 0.0) for r in runs) / n_seeds} for n in names}}
    table["mel_l1"]["bar"] = [FACTOR * v for v in table["mel_l1"]["emulated_max"]]
    table["loss_rel"]["bar"] = [max(FACTOR * v, 1e-4) for v in table["loss_rel"]["emulated_max"]]
    for n in names:
        table["grad"][n]["bar"] = FACTOR * table["grad"][n]["emulated_max"]
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "bf16_bars.json")
    json.dump(table, open(out, "w"), indent=1)
    bars = sorted(v["bar"] for v in table["grad"].values())
    print(f"wrote {out}: {len(names)} tensors, bars median {bars[len(bars) // 2]:.2e} max {bars[-1]:.2e}; mel L1 bars {table['mel_l1']['bar']}")


if __name__ == "__main__":
    main()
