"""Dev tool: where the HOST time of an eager train step goes (cProfile over 10 steps, no device sync inside)."""
import cProfile, pstats, sys, os, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

class A: pass
args = A(); args.dtype = "bf16"; args.batch = 48; args.phonemes = 128
dev = torch.device("cuda:0")
model, loss_fn, opt, b, pcfg, mcfg = bench.build(args, dev, 0, 1)
step, _ = bench.make_step(model, loss_fn, opt, b, None)
for _ in range(5):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    step()
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
print("\n".join(l[:150] for l in s.getvalue().split("\n")[:60]))
