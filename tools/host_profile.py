"""Dev tool: where the HOST time of an eager train step goes - cProfile over 20 steps of a SHORT batch (48 utterances of ~25
phonemes / ~190 frames: the LibriTTS-shaped steps that run at the ~4.2 ms issue floor whatever their size, profiles/r06a_libritts_sweep.log),
no device synchronisation inside the profiled region.  Prints wall ms/step, host issue ms/step, launches per step and the top
functions by own and by cumulative time."""
import cProfile, pstats, sys, os, io, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fastspeech2_amd
fastspeech2_amd.configure_hw_queues()
import torch
import bench
from fastspeech2_amd import _lib
from fastspeech2_amd.synthetic import synthetic_batch
from fastspeech2_amd.utils import lens_to_device

L = int(sys.argv[1]) if len(sys.argv) > 1 else 25
args = bench.parse(["--workload", "libritts"])
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
torch.cuda.set_stream(torch.cuda.Stream(device=dev, priority=-1))
model, loss_fn, opt, b0, pcfg, mcfg = bench.build(args, dev, 0, 1)
g = torch.Generator().manual_seed(5)
lens = torch.randint(max(5, L // 2), L + 1, (48,), generator=g).tolist()
b = synthetic_batch(77, 0, 0, dur_lo=4, dur_hi=10, n_speaker=2456, src_lens=lens)
b = {k: (lens_to_device(v, dev) if k in ("src_lens", "mel_lens") else v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in b.items()}
step, _ = bench.make_step(model, loss_fn, opt, b, None)
for _ in range(5):
    step()
torch.cuda.synchronize()
ncalls = [0]
real = _lib.call
def counted(*a):
    ncalls[0] += 1
    return real(*a)
_lib.call = counted
import fastspeech2_amd.ops as ops
ops._lib.call = counted
step(); torch.cuda.synchronize()
per_step = ncalls[0]
_lib.call = real; ops._lib.call = real
t0 = time.perf_counter()
for _ in range(20):
    step()
th = (time.perf_counter() - t0) / 20
torch.cuda.synchronize()
tw = (time.perf_counter() - t0) / 20
print(f"L={b['max_src_len']} T={b['max_mel_len']}: wall {tw*1e3:.3f} ms/step, host issue {th*1e3:.3f} ms/step, {per_step} C-ABI calls per step "
      f"-> {th*1e6/per_step:.1f} us of host time per call")
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    step()
pr.disable()
torch.cuda.synchronize()
for key in ("tottime", "cumulative"):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(32)
    print("\n".join(l[:160] for l in s.getvalue().split("\n")[:48]))
