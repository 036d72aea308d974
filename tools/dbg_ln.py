"""Dev tool: the C = 256 bf16 LayerNorm fast kernels against the generic ones (dev library, FS2_LN_FAST=0 / 3) and against an
fp64 torch reference, at the decoder's shape (48 x 925 rows, ragged lens), forward and backward, with and without dropout.
    python tools/dbg_ln.py          (parent: runs itself twice, compares the dumps)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(tag):
    import torch
    from fastspeech2_amd import ops
    dev = torch.device("cuda:0")
    B, S, C = 48, 925, 256
    g = torch.Generator().manual_seed(5)
    lens = torch.randint(S * 3 // 4, S + 1, (B,), generator=g).to(torch.int32).to(dev)
    out = {}
    for case, (p_pre, p_post, relu) in {"plain": (0.0, 0.0, False), "drop": (0.2, 0.0, False), "pred": (0.0, 0.5, True)}.items():
        y = (torch.randn(B * S, C, generator=g) * 1.5 + 0.3).to(dev).to(torch.bfloat16)
        res = torch.randn(B * S, C, generator=g).to(dev).to(torch.bfloat16) if not relu else None
        if relu:
            y = torch.relu(y)
        gamma = (1 + 0.1 * torch.randn(C, generator=g)).to(dev)
        beta = (0.1 * torch.randn(C, generator=g)).to(dev)
        dout = torch.randn(B * S, C, generator=g).to(dev).to(torch.bfloat16)
        d1a = torch.randn(B * S, C, generator=g).to(dev).to(torch.bfloat16)
        y0 = y.clone()
        o, mean, rstd = ops.ln_fwd(y, res, gamma, beta, lens, B, S, p_pre=p_pre, seed_pre=11, p_post=p_post, seed_post=12)
        dg = torch.zeros(C, device=dev); db = torch.zeros(C, device=dev)
        d1, d2 = ops.ln_bwd(y, dout, gamma, lens, mean, rstd, dg, db, B, S, want_d1=not relu, want_d2=True, d1_add=None if relu else d1a,
                            p_pre=p_pre, seed_pre=11, p_post=p_post, seed_post=12, relu_bwd=relu)
        torch.cuda.synchronize()
        out[case] = dict(y0=y0.cpu(), res=None if res is None else res.cpu(), z=y.cpu(), o=o.cpu(), mean=mean.cpu(), rstd=rstd.cpu(), dg=dg.cpu(), db=db.cpu(),
                         d1=None if d1 is None else d1.cpu(), d2=d2.cpu(), gamma=gamma.cpu(), beta=beta.cpu(), dout=dout.cpu(), d1a=d1a.cpu(), lens=lens.cpu())
    torch.save(out, f"/tmp/dbg_ln_{tag}.pt")


def main():
    import torch
    for tag, v in (("gen", "0"), ("fast", "3")):
        e = dict(os.environ, FS2_LIB_PATH=os.path.join(ROOT, "fastspeech2_amd", "libfs2hip_dev.so"), FS2_LN_FAST=v)
        subprocess.run([sys.executable, os.path.abspath(__file__), "child", tag], env=e, check=True, timeout=600)
    a, b = torch.load("/tmp/dbg_ln_gen.pt"), torch.load("/tmp/dbg_ln_fast.pt")
    for case in a:
        A, Bf = a[case], b[case]
        print(f"== {case}")
        for k in ("z", "o", "mean", "rstd", "d1", "d2", "dg", "db"):
            if A[k] is None:
                continue
            x, y = A[k].double(), Bf[k].double()
            nd = (x != y).sum().item()
            print(f"  {k:5s}: differing {nd}/{x.numel()} ({nd / x.numel():.2e}), max |diff| {(x - y).abs().max().item():.3e}, rel-Frobenius {((x - y).norm() / x.norm()).item():.3e}")
        if case == "plain":                                    # fp64 reference of the dropout-free case
            z = A["y0"].double() + A["res"].double()
            zr = z.to(torch.bfloat16).double()
            mu = zr.mean(1, keepdim=True); var = ((zr - mu) ** 2).mean(1, keepdim=True); rstd = (var + 1e-5).rsqrt()
            xh = (zr - mu) * rstd
            S = 925
            valid = (torch.arange(S).unsqueeze(0) < A["lens"].unsqueeze(1)).reshape(-1, 1).double()
            o = (xh * A["gamma"].double() + A["beta"].double()) * valid
            gdo = A["dout"].double() * valid
            gg = gdo * A["gamma"].double()
            dz = rstd * (gg - gg.mean(1, keepdim=True) - xh * (gg * xh).mean(1, keepdim=True))
            d1 = dz + A["d1a"].double()
            for name, X in (("generic", A), ("fast", Bf)):
                print(f"  {name} vs fp64: o {((X['o'].double() - o).norm() / o.norm()).item():.3e}  d1 {((X['d1'].double() - d1).norm() / d1.norm()).item():.3e}  "
                      f"d2 {((X['d2'].double() - dz).norm() / dz.norm()).item():.3e}  dgamma {((X['dg'].double() - (gdo * xh).sum(0)).norm() / (gdo * xh).sum(0).norm()).item():.3e}  "
                      f"dbeta {((X['db'].double() - gdo.sum(0)).norm() / gdo.sum(0).norm()).item():.3e}")


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "child":
        child(sys.argv[2])
    else:
        main()
