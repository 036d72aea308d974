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
