"""Dev tool: HOST time of one eager train step = the wall time of the same step on a batch so small that the device is never the
limiter (same launches, same Python)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

class A: pass
for batch, ph in ((2, 16), (48, 128)):
    a = A(); a.dtype = "bf16"; a.batch = batch; a.phonemes = ph; a.workload = "ljspeech"; a.dec_layers = 4; a.frame_level = False; a.side_stream = 1
    dev = torch.device("cuda:0")
    torch.cuda.set_stream(torch.cuda.Stream(device=dev, priority=-1))
    model, loss_fn, opt, b, _, _ = bench.build(a, dev, 0, 1)
    step, _ = bench.make_step(model, loss_fn, opt, b, None)
    for _ in range(6):
        step()
    ts = []
    for _ in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20):
            step()
        t1 = time.perf_counter()
        torch.cuda.synchronize(); t2 = time.perf_counter()
        ts.append(((t1 - t0) / 20 * 1e3, (t2 - t0) / 20 * 1e3))
    ts.sort()
    print(f"B={batch} L={ph}: enqueue {ts[1][0]:.3f} ms/step, wall {ts[1][1]:.3f} ms/step", flush=True)
