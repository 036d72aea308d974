"""Dev tool: the multi-tap contractions of the bench step's decoder / PostNet (B=48, T=925), forward and data gradient, timed
alone with HIP events; valid-row TFLOP/s (rows of fully padded tiles are skipped by the kernel)."""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fastspeech2_amd import ops, _lib

dev = torch.device("cuda:0")
B, S = 48, int(os.environ.get("BENCH_S", "925"))
M = B * S
g = torch.Generator().manual_seed(1)
lens = torch.randint(int(S * 0.75), S + 1, (B,), generator=g).to(torch.int32)
lens[0] = S
lens = torch.sort(lens, descending=True)[0].to(dev)
valid = int(lens.sum().item())
tmap = ops.tile_map(lens, B, S)
tws = ops.tail_workspace(dev)
# (name, Cin, N, taps, residual, act, lens?)
SHAPES = [("ffn k9 fwd+relu", 256, 1024, 9, False, ops.ACT_RELU, True), ("ffn k9 fwd+relu (no lens)", 256, 1024, 9, False, ops.ACT_RELU, False),
          ("ffn k9 dgrad+res", 1024, 256, 9, True, ops.ACT_NONE, True), ("ffn k9 dgrad+res (no lens)", 1024, 256, 9, True, ops.ACT_NONE, False),
          ("postnet k5 fwd", 512, 512, 5, False, ops.ACT_NONE, False), ("postnet k5 dgrad", 512, 512, 5, False, ops.ACT_NONE, False),
          ("postnet k5 80->512", 80, 512, 5, False, ops.ACT_NONE, False), ("postnet k5 512->80", 512, 80, 5, False, ops.ACT_NONE, False),
          ("pred k3 256->256", 256, 256, 3, False, ops.ACT_RELU, False), ("k9 512->512", 512, 512, 9, False, ops.ACT_RELU, False)]
ONLY = os.environ.get('BENCH_ONLY')
for name, K, N, taps, res, act, use_lens in SHAPES:
    if ONLY and not any(t in name for t in ONLY.split(',')):
        continue
    x = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, taps, K, device=dev) / math.sqrt(K * taps)).to(torch.bfloat16)
    bias = torch.randn(N, device=dev) if "fwd" in name else None
    r = torch.randn(M, N, device=dev).to(torch.bfloat16) if res else None
    l = lens if use_lens else None
    f = lambda: ops.conv_gemm(x, w, bias, S, taps=taps, pad=(taps - 1) // 2, lens=l, res=r, act=act, tmap=tmap if use_lens else None, tail_ws=tws)
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            f()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 10)
    ms = sorted(ts)[2]
    rows = valid if use_lens else M
    var = _lib.load().fs2_conv_gemm_variant(K, N, N if res else 0, int(use_lens), int(use_lens), M, N, K, S, taps, 1, 0, 0.0, 1)
    print(f"  {name:28s} variant {var} Cin={K:4d} N={N:4d} taps {taps}: {ms * 1e3:7.1f} us  {2.0 * rows * K * taps * N / ms / 1e9:7.1f} TF (valid rows {rows})", flush=True)
