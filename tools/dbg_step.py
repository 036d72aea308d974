"""Dev tool: one full-size bf16 train step (dropout off) under two dev-library settings; per-tensor difference of the outputs and
gradients BETWEEN the two runs (no oracle): how far an arithmetic-neutral kernel change moves the step.
    python tools/dbg_step.py "FS2_LN_FAST=0" "FS2_LN_FAST=1" """
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(tag):
    import torch
    from oracle.weights import seeded_state_dict, synthetic_batch
    from tests.golden import configs
    from tests.helpers import make_model
    from tests.test_model_gpu import run_train
    dev = torch.device("cuda:0")
    pcfg, mcfg = configs.make(dec_layers=4, enc_layers=4, dropout=False)
    model = make_model(pcfg, mcfg, "fp32")
    sd = seeded_state_dict(model.state_dict(), 2025)
    b = synthetic_batch(1234, 48, 128, dur_lo=4, dur_hi=10, min_len_frac=0.75)
    model = make_model(pcfg, mcfg, "bf16")
    model.load_state_dict(sd)
    model.to(dev).train()
    model.disable_dropout = True
    out, losses = run_train(model, pcfg, mcfg, b, dev)
    grads = {n: p.grad.detach().cpu().double() for n, p in model.named_parameters() if p.grad is not None}
    torch.save(dict(mel=out[0].detach().float().cpu(), post=out[1].detach().float().cpu(), target=b["mels"], mask=out[7].cpu(),
                    losses=[l.item() for l in losses], grads=grads), f"/tmp/dbg_step_{tag}.pt")


def main(cfgs):
    import torch
    for i, c in enumerate(cfgs):
        e = dict(os.environ, FS2_LIB_PATH=os.path.join(ROOT, "fastspeech2_amd", "libfs2hip_dev.so"))
        e.update(dict(kv.split("=") for kv in c.split(",") if kv))
        subprocess.run([sys.executable, os.path.abspath(__file__), "child", str(i)], env=e, check=True, timeout=900)
    a = torch.load("/tmp/dbg_step_0.pt")
    for i in range(1, len(cfgs)):
        b = torch.load(f"/tmp/dbg_step_{i}.pt")
        print(f"== {cfgs[0]}  vs  {cfgs[i]}")
        print("  losses", a["losses"], b["losses"])
        valid = (~a["mask"]).unsqueeze(-1)
        tgt = torch.as_tensor(a["target"]).float()
        for k in ("mel", "post"):
            x, y = a[k], b[k]
            nd = ((x != y) & valid).sum().item()
            sa, sb = torch.sign(x - tgt), torch.sign(y - tgt)
            print(f"  {k}: differing {nd}/{int(valid.sum()) * 80} ({nd / (valid.sum().item() * 80):.2e}), max |diff| {((x - y).abs() * valid).max().item():.3e}, "
                  f"rel-Frobenius {(((x - y) * valid).norm() / (x * valid).norm()).item():.3e}; L1 sign flips between the runs {((sa != sb) & valid).sum().item()}; "
                  f"|out - target| < 0.05 on {(((x - tgt).abs() < 0.05) & valid).sum().item()} elements")
        rows = sorted((((a["grads"][n] - b["grads"][n]).norm() / a["grads"][n].norm().clamp_min(1e-30)).item(), n) for n in a["grads"] if a["grads"][n].abs().max() > 0)
        rows.reverse()
        big = [r for r in rows if a["grads"][r[1]].numel() > 1024]
        print(f"  gradient difference between the runs: weight tensors median {big[len(big) // 2][0]:.2e} max {big[0][0]:.2e}")
        for f, n in rows[:12]:
            print(f"    {f:.3e}  {n}")


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "child":
        child(sys.argv[2])
    else:
        main(sys.argv[1:])
