"""Dev tool: the encoder-side (M = 48 x 128 = 6144 rows) contractions of the bench step, each timed alone with HIP events, with the kernel
the dispatcher picks for it - the launches that are bound by occupancy / launch latency rather than by a roofline."""
import sys, os, math, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fastspeech2_amd import ops, _lib
dev = torch.device("cuda:0")
B, S = 48, 128
M = B * S
lens = torch.randint(96, 129, (B,), generator=torch.Generator().manual_seed(1)).to(torch.int32).to(dev)
SH = [("qkv fwd", 256, 768, 1), ("qkv dgrad", 768, 256, 1), ("fc fwd", 256, 256, 1), ("w_1 k9 fwd", 256, 1024, 9), ("w_1 k9 dgrad", 1024, 256, 9),
      ("w_2 fwd", 1024, 256, 1), ("w_2 dgrad", 256, 1024, 1), ("pred k3", 256, 256, 3), ("pred k3 (M = 44400)", 256, 256, 3)]
tws = ops.tail_workspace(dev)
lib = _lib.load()
for name, K, N, taps in SH:
    m = 48 * 925 if "44400" in name else M
    s = 925 if "44400" in name else S
    x = torch.randn(m, K, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, taps, K, device=dev) / math.sqrt(K * taps)).to(torch.bfloat16)
    bias = torch.randn(N, device=dev)
    y = torch.empty(m, N, device=dev, dtype=torch.bfloat16)
    f = lambda: ops.conv_gemm(x, w, bias, s, taps=taps, pad=(taps - 1) // 2, act=ops.ACT_RELU, out=y, tail_ws=tws)
    for _ in range(5): f()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): f()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 20)
    var = lib.fs2_conv_gemm_variant(K, N, 0, 0, 0, m, N, K, s, taps, 1, 0, ctypes.c_float(0.0), 1)
    print(f"{name:22s} M={m:6d} K={K * taps:5d} N={N:5d}  {best * 1e3:7.1f} us  {2.0 * m * N * K * taps / best / 1e9:7.1f} TF  variant {var}", flush=True)
