"""Dev tool: decoder-shape attention (B=48, S=925, H=2, d=128, the bench batch's lengths) forward and backward, HIP-event timed."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fastspeech2_amd import ops
from fastspeech2_amd.synthetic import synthetic_batch
dev = torch.device("cuda:0")
b = synthetic_batch(1234, 48, 128, dur_lo=4, dur_hi=10, min_len_frac=0.75)
lens64 = b["mel_lens"] if isinstance(b, dict) else None
B, S, H = 48, int(b["max_mel_len"]), 2
lens = torch.as_tensor(lens64).to(torch.int32).to(dev)
qkv = (torch.randn(B * S, 3 * H * 128, device=dev) * 0.5).to(torch.bfloat16)
dctx = torch.randn(B * S, H * 128, device=dev).to(torch.bfloat16)
ctx, lse = ops.attn_fwd(qkv, lens, B, S, H)
pairs = float((lens.double() ** 2).sum().item()) * H
def timeit(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): f()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / n)
    return sorted(ts)[2]
t = timeit(lambda: ops.attn_fwd(qkv, lens, B, S, H))
print(f"  fwd      {t * 1e3:7.1f} us  {4 * 128 * pairs / t / 1e9:7.1f} TF (valid pairs)", flush=True)
t = timeit(lambda: ops.attn_bwd(qkv, ctx, dctx, lse, lens, B, S, H))
print(f"  bwd      {t * 1e3:7.1f} us  {14 * 128 * pairs / t / 1e9:7.1f} TF (7 products, valid pairs)", flush=True)
if os.environ.get("ATTN_STAMPS"):
    dbg = torch.zeros(16 * 64, dtype=torch.int64, device=dev)
    os.environ["FS2_ATTN_DBG_PTR"] = str(dbg.data_ptr())
    ops.attn_fwd(qkv, lens, B, S, H); torch.cuda.synchronize()
    dall = dbg.cpu().view(16, 64); d = dall[:8]; dv = dall[8:]
    t0 = int(d[:, 0].min())
    for w in (0, 4):
        r = [int(x) - t0 if int(x) else -1 for x in d[w]]
        print(f"wave {w}: start {r[0]} init {r[1]} | " + " ".join(f"[t{t}: sync {r[4*t-2]} a {r[4*t-1]} bar {r[4*t]} b {r[4*t+1]}]" for t in range(1, 6)) + f" | loop end {r[62]} exit {r[63]}")
        per = [(r[4 * t - 1] - r[4 * t - 2], r[4 * t] - r[4 * t - 1], r[4 * t + 1] - r[4 * t], (r[4 * t + 2] - r[4 * t + 1]) if t < 13 else 0) for t in range(2, 12)]
        print("   per tile (phase A work, wait at barrier, phase B work, wait at even sync):", per)
    for t in (5, 6):
        print(f"tile {t}: per wave [vm-wait done, sync exit, A end, barrier exit, B end] relative to wave 0's sync exit")
        ref = int(d[0, 4 * t - 2])
        for w in range(8):
            print("   wave", w, [int(dv[w, t]) - ref] + [int(d[w, 4 * t - 2 + i]) - ref for i in range(4)])
