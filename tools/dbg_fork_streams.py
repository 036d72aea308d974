"""Dev tool: which HIP streams the pieces of a train step run on inside a forked hipGraph capture (tools/dbg_fork_capture.py found
the first backward kernels reading the forward's last outputs before they are written)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fastspeech2_amd
fastspeech2_amd.configure_hw_queues()
import torch
import bench
from fastspeech2_amd import engine as E, ops

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
args = bench.parse(["--batch", "8", "--phonemes", "40", "--dtype", "fp32"])
torch.manual_seed(1234)
model, loss_fn, opt, b, _, _ = bench.build(args, dev, 0, 1)
model.disable_dropout = True
model._engine.device_seed = True
step, fwd_bwd = bench.make_step(model, loss_fn, opt, b, None)
for _ in range(2):
    fwd_bwd()
torch.cuda.synchronize()
eng = model._engine
log = []
real_begin = E.Engine._side_begin
def begin(self):
    real_begin(self)
    log.append(("backward: _main", self._main.cuda_stream, "side", None if self._side is None else self._side.cuda_stream,
                "pinned", ops._stream(), "thread", __import__("threading").current_thread().name))
E.Engine._side_begin = begin
real_lb = ops.loss_bwd
def lb(*a, **k):
    log.append(("loss_bwd on", ops._stream(), "thread", __import__("threading").current_thread().name))
    return real_lb(*a, **k)
ops.loss_bwd = lb
import fastspeech2_amd.model as M
if hasattr(M, "ops"):
    M.ops.loss_bwd = lb
real_fwd = E.Engine._forward
def fwd(self, *a, **k):
    log.append(("forward on", ops._stream(), "current", torch.cuda.current_stream().cuda_stream))
    return real_fwd(self, *a, **k)
E.Engine._forward = fwd
calls = []
real_call = ops._lib.call
def traced(name, *a):
    calls.append((name, a[-1], __import__("threading").current_thread().name))
    return real_call(name, *a)
ops._lib.call = traced
for mode in ("eager", "capture-forked"):
    log.clear()
    calls.clear()
    if mode == "eager":
        fwd_bwd()
    else:
        eng.fork_in_capture = True
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            cs = torch.cuda.current_stream().cuda_stream
            fwd_bwd()
        log.append(("capture stream", cs))
    torch.cuda.synchronize()
    print(mode, "| side stream object:", None if eng._side_stream is None else eng._side_stream.cuda_stream)
    for l in log:
        print("   ", l)
    i0 = next((i for i, c in enumerate(calls) if c[0] == "fs2_loss_bwd"), 0)
    names = {}
    def nm(h):
        return names.setdefault(h, "C" if not names else "S%d" % len(names))
    for c in calls[:i0]:
        nm(c[1])
    print("    stream of every C-ABI call from 6 before fs2_loss_bwd to 30 after (C = the stream of the first call of the step):")
    print("    " + " ".join(f"{c[0][4:]}@{nm(c[1])}" for c in calls[max(0, i0 - 6):i0 + 30]))
