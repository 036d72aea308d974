"""Op-by-op LOCAL check of the product's bf16 PostNet backward chain on the tensors of a real full-size step.

  python tools/spread_local.py [reps]

For each kernel of the chain (BatchNorm backward i = 4..0, k = 5 data gradient i = 4..0, the final add) the product's OUTPUT is
compared with an fp64 evaluation of the same op on the product's own INPUTS (so upstream noise does not enter): relative error,
the share of it that is coherent over rows, the largest elementwise error in units of the output's bf16 spacing, and how many
elements are off by more than 1 spacing.  A healthy bf16 kernel shows ~2.3e-3 relative, coherent share ~1/sqrt(rows), max <= 0.5-1.
With reps > 1 the runs are also diffed against each other (where does run-to-run nondeterminism enter, and how is it laid out).
oracle-free: the fp64 evaluation is plain torch on the CPU."""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.weights import seeded_state_dict, synthetic_batch  # noqa: E402
from tests.golden import configs  # noqa: E402
from tests.helpers import make_model  # noqa: E402


def spacing(ref):
    """bf16 spacing at |ref| (8 significant bits)"""
    e = torch.floor(torch.log2(ref.abs().clamp_min(1e-45)))
    return torch.pow(2.0, e - 7)


def report(tag, got, ref, valid):
    got = got.double().cpu()
    E = (got - ref)
    R = E.shape[0]
    n = E.norm().item()
    coh = E.mean(0).norm().item() * R ** 0.5 / max(n, 1e-300)
    Ev = E[valid]
    cohv = Ev.mean(0).norm().item() * Ev.shape[0] ** 0.5 / max(Ev.norm().item(), 1e-300)
    u = (E.abs() / spacing(ref))
    bad = (u > 1.0)
    msg = (f"  {tag:<10} rel {n / ref.norm().item():.2e}  coherent share all rows {coh:.3f} valid rows {cohv:.3f} (noise {R ** -0.5:.3f})  "
           f"max err {u.max().item():.2f} spacings, > 1 spacing: {int(bad.sum())} of {E.numel()}")
    if bad.any():
        rows = bad.any(1).nonzero().flatten()
        msg += f"  rows {rows[:12].tolist()}{'...' if len(rows) > 12 else ''} ({len(rows)} rows)"
    print(msg, flush=True)


def substitution(cap, sd, b, pcfg, mcfg, B, T, valid):
    """where does the error of d mel (projected on the decoder output: mel_linear.weight.grad) come from?  fp64 backward chains
    that mix the product's tensors with the exact oracle's."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import spread
    from tests.helpers import bf16_matrix
    sdr = {k: (v.to(torch.bfloat16).to(v.dtype) if bf16_matrix(k, v) else v) for k, v in sd.items()}
    if not hasattr(substitution, "ex"):
        substitution.ex = spread.oracle_case(pcfg, mcfg, sdr, b)[1]
    ex = substitution.ex
    X = ex["x"]
    dW = ex["dmel"].t() @ X
    n = int(b["mel_lens"].sum()) * 80
    c0 = 1.0 / n

    def proj(dy):
        return ((dy - ex["dmel"]).t() @ X).norm().item() / dW.norm().item(), ((dy - ex["dmel"])[valid].norm() / ex["dmel"][valid].norm()).item()

    def chain(dpost, dmel, fwd):
        """exact fp64 backward through the 5 PostNet layers: fwd[i] = (c_i (rows), mean, rstd, gamma, beta, act)"""
        g = dpost
        for i in (4, 3, 2, 1, 0):
            xd, mu, rs, gm, bt, act = fwd[i]
            xh = (xd - mu) * rs
            gg = g
            if act:
                t = torch.tanh(xh * gm + bt)
                gg = g * (1 - t * t)
            M = xd.shape[0]
            dc = gm * rs * (gg - gg.sum(0) / M - xh * (gg * xh).sum(0) / M)
            w = sdr[f"postnet.convolutions.{i}.0.conv.weight"].double()
            g = F.conv_transpose1d(dc.view(B, T, -1).transpose(1, 2), w, padding=2).transpose(1, 2).reshape(B * T, -1)
        return g + dpost + dmel

    fwd_p = {}
    for j, (x, dout, mr, gm, bt, act, r) in enumerate(cap["bn"]):
        C = x.shape[1]
        fwd_p[4 - j] = (x.double().cpu(), mr[:C].double().cpu(), mr[C:].double().cpu(), gm.double().cpu(), bt.double().cpu(), act == 2)
    dpost_p = cap["bn"][0][1].double().cpu()
    dmel_p = cap["add"][1].double().cpu()
    # the exact oracle's forward tensors and loss gradients
    fwd_e = {}
    for i in range(5):
        c = ex["fwd_c"][i]
        mu, var = c.mean(0), c.var(0, unbiased=False)
        fwd_e[i] = (c, mu, (var + 1e-5).rsqrt(), fwd_p[i][3], fwd_p[i][4], i < 4)
    dpost_e, dmel_e = ex["dpost"], ex["dmel_loss"]
    print("  fp64 backward chains (E^T X / |dW|, rel err of d mel over valid rows):")
    print("    exact forward,   exact loss gradients (sanity, ~0):        %.2e  %.2e" % proj(chain(dpost_e, dmel_e, fwd_e)))
    print("    exact forward,   product loss gradients (flips + c0 rounding): %.2e  %.2e" % proj(chain(dpost_p, dmel_p, fwd_e)))
    print("    product forward, exact loss gradients (saved activations):  %.2e  %.2e" % proj(chain(dpost_e, dmel_e, fwd_p)))
    print("    product forward, product loss gradients (no bwd roundings): %.2e  %.2e" % proj(chain(dpost_p, dmel_p, fwd_p)))
    print("    the product's d mel:                                        %.2e  %.2e" % proj(cap["add"][2].double().cpu()))
    for i in range(5):
        E_ = (fwd_p[i][0] - fwd_e[i][0])[valid]
        print(f"    forward c{i}: rel err {E_.norm().item() / fwd_e[i][0][valid].norm().item():.2e}  coherent share {E_.mean(0).norm().item() * E_.shape[0] ** 0.5 / E_.norm().item():.3f}")


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    torch.set_num_threads(min(64, os.cpu_count() or 8))
    pcfg, mcfg = configs.make(dec_layers=4, enc_layers=4, dropout=False)
    sd = seeded_state_dict(make_model(pcfg, mcfg, "fp32").state_dict(), 2025)
    b = synthetic_batch(1234, 48, 128, dur_lo=4, dur_hi=10, min_len_frac=0.75)
    B, T = 48, int(b["max_mel_len"])
    valid = (torch.arange(T)[None, :] < b["mel_lens"][:, None]).reshape(-1)
    dev = torch.device("cuda:0")
    from fastspeech2_amd import engine as E, ops
    from tests.test_model_gpu import run_train
    bn0, add0, dg0 = ops.bn_bwd_acc, ops.add, E.Engine._dgemm
    cap = {}

    def bn_hook(x, dout, mean_rstd, gamma, beta, act, *a, **k):
        r = bn0(x, dout, mean_rstd, gamma, beta, act, *a, **k)
        cap.setdefault("bn", []).append((x, dout, mean_rstd, gamma, beta, act, r))
        return r

    def add_hook(x, y):
        r = add0(x, y)
        if r.dim() == 2 and r.shape[1] == 80:
            cap["add"] = (x, y, r)
        return r

    def dg_hook(self, W, key, dy, S, **k):
        r = dg0(self, W, key, dy, S, **k)
        if key.startswith("postnet"):
            cap.setdefault("dg", []).append((key, dy, k.get("res"), r))
        return r

    ops.bn_bwd_acc, ops.add, E.Engine._dgemm = bn_hook, add_hook, dg_hook
    runs = []
    for rep in range(reps):
        cap.clear()
        m = make_model(pcfg, mcfg, "bf16")
        m.load_state_dict(sd)
        m.to(dev).train()
        m.disable_dropout = True
        m._ensure_flat(dev)
        m._engine.use_side_stream = False
        out, losses = run_train(m, pcfg, mcfg, b, dev)
        torch.cuda.synchronize()
        snap = {"mel": out[0].detach().clone(), "post": out[1].detach().clone()}
        for j, (x, dout, mr, gm, bt, act, r) in enumerate(cap["bn"]):
            i = 4 - j
            snap[f"c{i}"], snap[f"gin{i}"], snap[f"dc{i}"] = x, dout, r
        for j, (key, dy, res, r) in enumerate(cap["dg"]):
            snap[f"g{4 - j}"] = r
        snap["dmel"] = cap["add"][2]
        runs.append({k: v.detach().clone() for k, v in snap.items()})
        if rep == 0:
            print("LOCAL checks (product output vs fp64 evaluation of the same op on the product's inputs), run 0:")
            for j, (x, dout, mr, gm, bt, act, r) in enumerate(cap["bn"]):
                i = 4 - j
                xd, gd = x.double().cpu(), dout.double().cpu()
                C = xd.shape[1]
                mu, rs = mr[:C].double().cpu(), mr[C:].double().cpu()
                # statistics the kernel was GIVEN (fp32, from the forward) vs the exact ones of the stored bf16 tensor
                mu_x, var_x = xd.mean(0), xd.var(0, unbiased=False)
                print(f"  bn{i}: forward statistics vs exact statistics of the stored tensor: mean abs diff / std {((mu - mu_x).abs() / var_x.sqrt()).max().item():.2e}, "
                      f"rstd rel diff {((rs - (var_x + 1e-5).rsqrt()).abs() / rs).max().item():.2e}")
                xh = (xd - mu) * rs
                gg = gd
                if act == ops.ACT_TANH:
                    t = torch.tanh(xh * gm.double().cpu() + bt.double().cpu())
                    gg = gd * (1 - t * t)
                M = xd.shape[0]
                ref = gm.double().cpu() * rs * (gg - gg.sum(0) / M - xh * (gg * xh).sum(0) / M)
                report(f"bn_bwd{i}", r, ref, valid)
                key, dy, res, g = cap["dg"][j]
                w = sd[f"postnet.convolutions.{i}.0.conv.weight"].to(torch.bfloat16).double()        # (Cout, Cin, k)
                dyd = dy.double().cpu().view(B, T, -1).transpose(1, 2)
                ref = F.conv_transpose1d(dyd, w, padding=2).transpose(1, 2).reshape(B * T, -1)
                if res is not None:
                    ref = ref + res.double().cpu()
                report(f"dgrad{i}", g, ref, valid)
            x, y, r = cap["add"]
            report("add", r, x.double().cpu() + y.double().cpu(), valid)
            if os.environ.get("SPREAD_SUBST"):
                substitution(cap, sd, b, pcfg, mcfg, B, T, valid)
        del m
    if reps > 1:
        print("run-to-run differences (run k vs run 0): elements that differ / total, rows touched, max |diff| in spacings")
        order = ["mel", "post"] + [f"{p}{i}" for i in (4, 3, 2, 1, 0) for p in ("c", "gin", "dc", "g")] + ["dmel"]
        for k in range(1, reps):
            parts = []
            for name in order:
                if name not in runs[0]:
                    continue
                a, c = runs[0][name].float().cpu(), runs[k][name].float().cpu()
                a, c = a.reshape(-1, a.shape[-1]), c.reshape(-1, c.shape[-1])
                d = (a != c)
                if not d.any():
                    parts.append(f"{name}: =")
                    continue
                u = ((a - c).abs().double() / spacing(a.double()))[d]
                parts.append(f"{name}: {int(d.sum())}/{d.numel()} in {int(d.any(1).sum())} rows, max {u.max().item():.1f}")
            print(f"  run {k}: " + "   ".join(parts), flush=True)


if __name__ == "__main__":
    main()
