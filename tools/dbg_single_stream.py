"""Which gradient tensors differ between two identically seeded eager steps, by stream mode (fp32, small batch)?
side=1 runs are reproducible to fp32-atomics noise; this prints what side=0 runs do."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

dev = torch.device("cuda:0")


def grads(side, nsteps=1):
    args = bench.parse(["--batch", "8", "--phonemes", "40", "--dtype", "fp32", "--side-stream", str(side)])
    torch.manual_seed(1234)
    model, loss_fn, opt, b, _, _ = bench.build(args, dev, 0, 1)
    model._engine.reseed(seed=4242, rank=0)
    step, fwd_bwd = bench.make_step(model, loss_fn, opt, b, None)
    for _ in range(nsteps - 1):
        step()
    opt.zero_grad()
    fwd_bwd()
    torch.cuda.synchronize()
    return {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}


for ns in (1, 3):
    a0, b0, a1, b1 = grads(0, ns), grads(0, ns), grads(1, ns), grads(1, ns)
    def worst(x, y):
        r = sorted((((x[n] - y[n]).norm() / (x[n].norm() + 1e-30)).item(), n) for n in x
                   if "w_ks.bias" not in n and not (n.startswith("postnet") and n.endswith("0.conv.bias")))      # true-zero gradients: noise only
        return [(f"{v:.1e}", n) for v, n in r[-5:]] + [('median', f'{r[len(r) // 2][0]:.1e}')]
    print(f"after {ns} step(s): single vs single {worst(a0, b0)}")
    print(f"                 forked vs forked {worst(a1, b1)}")
    print(f"                 single vs forked {worst(a0, a1)}", flush=True)
