"""The bf16 WHOLE-STEP budget at production size (B = 48, L = 128, T ~ 925, 4 + 4 layers) - collected LAST on purpose.

It is the one GPU test whose bar is statistical (a distance to an fp64 oracle compared with the distance bf16 storage costs by
construction) rather than elementwise; every kernel, golden, STFT, optimiser, length-regulator, DDP and CLI test runs before it
under `pytest -x` (a 1.4 % overshoot of one tensor here, in the first-collected file, hid 115 tests from the
driver).  The fp32 whole-step test (elementwise, every gradient tensor) stays in tests/test_a_prodshape_gpu.py.
Reference: model/fastspeech2.py:43-110, model/loss.py:19-92."""
import pytest
import torch

from tests.conftest import train_step_grads
from tests.helpers import bf16_matrix, oracle_train_case

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def full_case_bf16_weights(full_case):
    """the same fp64 oracle run with every floating-point parameter ROUNDED to bf16 first: the bf16 engine multiplies with
    bf16 copies of the fp32 master weights, i.e. it differentiates that slightly different network; against THIS oracle only the
    rounding of stored activations / gradients (and fp32 accumulation) is left."""
    pcfg, mcfg, sd, b, _, _, _ = full_case
    sdr = {k: (v.to(torch.bfloat16).to(v.dtype) if bf16_matrix(k, v) else v) for k, v in sd.items()}
    oout, olosses, ograds, _ = oracle_train_case(pcfg, mcfg, sdr, b, dtype=torch.float64)
    return oout, olosses, ograds


def test_full_size_train_step_bf16_per_tensor_budget(dev, full_case, full_case_bf16_weights):
    """bf16 storage + bf16 MFMA (fp32 accumulate / statistics / master weights) at full size, judged against bars DERIVED from
    bf16 itself, ON THE ORACLE SIDE ONLY, and frozen in tests/golden/bf16_bars.json (not chosen, not to be edited without the table
    changing).

    tests/golden/make_bf16_bars.py runs the fp64 oracle on the network the bf16 engine differentiates (matrices rounded to bf16):
    exact, and EMULATED with `oracle.storage(...)`, which rounds every activation the product stores between two kernels (and the
    gradient flowing back through that edge) to bf16 - over 8 seeded (weights, batch) pairs x 4 rounding REALISATIONS each (the
    plain emulation + 3 runs of `oracle.realisation`: a relative 2^-19 perturbation in front of every rounding, the size of an
    fp32 accumulation's own error, which flips the roundings near a tie; the step is chaotic at its rounding level, so each
    realisation is an independent draw of the heavy-tailed per-tensor distance - what round 4 sampled by re-running the product).
    The emulation's own distance to the exact run is what bf16 storage costs by construction: per parameter tensor (relative
    Frobenius), per output (valid-frame L1), per loss.  bar = 2 x the maximum over the 32 emulated runs, for EVERY tensor by name.
    Round 4's bars also folded in the product's own measured maximum (a self-referential bar, up to 2.86 x wider,
    measured on a build that no longer exists): gone.  The product / emulated ratios are PRINTED for information.
    The comparison is to a distance, not to the emulation's values.  The BatchNorm column sums are bit-reproducible since r04, so
    this test measures ONE number per build, the same on every box.  Against the fp32-master-weight oracle (a slightly different
    network: the weight rounding adds its share) the numbers are printed for the record only."""
    import json
    import os
    bars = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "bf16_bars.json")))
    assert bars["n_seeds"] >= 8 and bars["factor"] == 2.0 and bars["n_emulated_runs"] >= 32 and "product" not in bars
    assert all(abs(row["bar"] - 2.0 * row["emulated_max"]) <= 1e-12 for row in bars["grad"].values())       # oracle side only
    pcfg, mcfg, sd, b, oout, olosses, ograds = full_case
    wout, wlosses, ograds_w = full_case_bf16_weights
    out, losses, grads = train_step_grads(dev, pcfg, mcfg, sd, b, "bf16")
    assert torch.equal(out[9].cpu(), oout[9])
    valid = (~oout[7]).unsqueeze(-1)
    nval = valid.sum().item() * 80
    failures = []
    for i in (0, 1):
        l1 = ((out[i].detach().float().cpu().double() - wout[i].detach()).abs() * valid).sum().item() / nval
        print(f"bf16 full-size valid-frame mel L1 [{i}] = {l1:.3e}  (bar {bars['mel_l1']['bar'][i]:.3e} = 2 x emulated {bars['mel_l1']['emulated_max'][i]:.3e})")
        if l1 > bars["mel_l1"]["bar"][i]:
            failures.append(("mel_l1", i, l1))
    for i, (a, o) in enumerate(zip(losses, wlosses)):
        rel = abs(a.item() - o.item()) / max(1.0, abs(o.item()))
        if rel > bars["loss_rel"]["bar"][i]:
            failures.append(("loss", i, rel, bars["loss_rel"]["bar"][i]))
    gmax = max(g.abs().max().item() for g in ograds.values())
    worst, ratios = [], []
    for n, og in ograds_w.items():
        if ograds[n].abs().max().item() < 1e-9 * gmax:   # true gradient zero (w_ks.bias, conv biases in front of BatchNorm): noise only
            if grads[n].abs().max().item() > 1e-3 * gmax:
                failures.append((n, "zero-gradient tensor", grads[n].abs().max().item()))
            continue
        fro = ((grads[n] - og).norm() / og.norm()).item()
        bar = bars["grad"][n]["bar"]
        worst.append((fro, n))
        ratios.append((fro / bars["grad"][n]["emulated_max"], n))
        if fro > bar:
            failures.append((n, fro, bar))
    worst.sort(reverse=True)
    ratios.sort(reverse=True)
    big = sorted(f for f, n in worst if ograds_w[n].numel() > 1024)
    print(f"bf16 full-size per-tensor relative Frobenius error vs the fp64 oracle with bf16-rounded weights: weight tensors median "
          f"{big[len(big) // 2]:.2e} max {big[-1]:.2e}; worst 6: {[(f'{f:.2e}', n) for f, n in worst[:6]]}")
    print(f"largest product / emulated-max ratios (information; the bar is at 2.00): {[(f'{r:.2f}', n) for r, n in ratios[:8]]}")
    fm = sorted(((grads[n] - og).norm() / og.norm()).item() for n, og in ograds.items() if og.abs().max().item() >= 1e-9 * gmax and og.numel() > 1024)
    print(f"for the record, vs the fp32-master-weight oracle (adds the weight rounding): weight tensors median {fm[len(fm) // 2]:.2e} max {fm[-1]:.2e}")
    assert not failures, failures
