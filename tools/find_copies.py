"""Dev tool: which Python lines of an eager train step end in a device copy / fill (torch profiler, CPU side: aten::copy_,
aten::clone, aten::zero_, aten::fill_, aten::cat, aten::to with their Python callers)."""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from torch.profiler import profile, ProfilerActivity

class A: pass
a = A(); a.dtype = "bf16"; a.batch = 48; a.phonemes = 128; a.workload = "ljspeech"; a.dec_layers = 4; a.frame_level = False; a.side_stream = 1
dev = torch.device("cuda:0")
model, loss_fn, opt, b, _, _ = bench.build(a, dev, 0, 1)
step, _ = bench.make_step(model, loss_fn, opt, b, None)
for _ in range(5):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=True) as prof:
    for _ in range(2):
        step()
torch.cuda.synchronize()
cnt = collections.Counter()
for ev in prof.events():
    if ev.name in ("aten::copy_", "aten::zero_", "aten::fill_", "aten::cat", "aten::clone", "aten::_to_copy", "aten::contiguous"):
        st = [s for s in (ev.stack or []) if "fastspeech2_amd" in s or "bench.py" in s]
        shp = str(ev.input_shapes)[:60]
        cnt[(ev.name, st[0] if st else "?", shp)] += 1
for (name, where, shp), n in sorted(cnt.items(), key=lambda kv: -kv[1])[:40]:
    print(f"{n / 2:5.1f}/step  {name:16s} {where[-90:]:90s} {shp}")
