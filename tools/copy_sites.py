"""Dev tool: which Python lines of a train step launch the small torch kernels (copies, fills, elementwise) that sit between the
library's launches?  One profiled step (torch.profiler, CPU + device activities, Python stacks); prints torch-op call sites that
are NOT fs2_* launches, with their device time and count."""
import collections, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench


def main():
    class A: pass
    a = A(); a.dtype = "bf16"; a.batch = 48; a.phonemes = 128; a.workload = "ljspeech"; a.dec_layers = 4; a.frame_level = False; a.side_stream = 1
    dev = torch.device("cuda:0")
    torch.cuda.set_stream(torch.cuda.Stream(device=dev, priority=-1))
    model, loss_fn, opt, b, _, _ = bench.build(a, dev, 0, 1)
    step, _ = bench.make_step(model, loss_fn, opt, b, None)
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        step()
        torch.cuda.synchronize()
    agg = collections.defaultdict(lambda: [0, 0.0])
    for ev in prof.events():
        if not ev.name.startswith("aten::") or ev.device_time_total <= 0 and not ev.kernels:
            continue
        if not ev.kernels:
            continue
        site = "?"
        for fr in ev.stack or []:
            if "/root/repo" in fr or "fastspeech2_amd" in fr or "bench.py" in fr:
                site = fr.strip()
                break
        k = (ev.name, site[-110:])
        agg[k][0] += 1
        agg[k][1] += sum(kk.duration for kk in ev.kernels)
    rows = sorted(agg.items(), key=lambda kv: -kv[1][0])
    print(f"{'op':28s} {'n':>4s} {'dev us':>8s}  site")
    for (name, site), (n, us) in rows[:60]:
        print(f"{name:28s} {n:4d} {us:8.1f}  {site}")
    print("total torch-op launches per step:", sum(v[0] for v in agg.values()), " device us:", round(sum(v[1] for v in agg.values()), 1))


if __name__ == "__main__":
    main()
