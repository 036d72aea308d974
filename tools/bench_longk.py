"""Dev tool: one-tap contractions with the reduction lengths of the k=9 / k=5 convolutions, to compare kernel STRUCTURES at long K
(wide 256x256 self-loading kernel vs persistent loader/consumer 256x128): FS2_LIB_PATH=...dev.so FS2_GEMM_W=0/1."""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fastspeech2_amd import ops, _lib
dev = torch.device("cuda:0")
B, S = 48, 925
M = B * S
tws = ops.tail_workspace(dev)
for K, N in ((2304, 1024), (9216, 256), (2560, 512), (1024, 1024), (256, 1024)):
    x = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, 1, K, device=dev) / math.sqrt(K)).to(torch.bfloat16)
    f = lambda: ops.conv_gemm(x, w, None, S, tail_ws=tws)
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
    var = _lib.load().fs2_conv_gemm_variant(K, N, 0, 1, 1, M, N, K, S, 1, 1, 0, 0.0, 1)
    print(f"  K={K:5d} N={N:4d} variant {var}: {ms * 1e3:7.1f} us  {2.0 * M * K * N / ms / 1e9:7.1f} TF", flush=True)
