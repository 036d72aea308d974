"""Dev tool: time ONE conv_gemm shape (for PMC runs / A-B of kernel variants).
usage: python tools/bench_one.py M N Cin taps [S] [reps]"""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fastspeech2_amd import ops

M, N, Cin, taps = [int(v) for v in sys.argv[1:5]]
S = int(sys.argv[5]) if len(sys.argv) > 5 else M
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 10
dev = torch.device("cuda:0")
x = torch.randn(M, Cin, device=dev).to(torch.bfloat16)
w = (torch.randn(N, taps, Cin, device=dev) / math.sqrt(Cin * taps)).to(torch.bfloat16)
b = torch.randn(N, device=dev)
y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
for _ in range(3):
    ops.conv_gemm(x, w, b, S, taps=taps, pad=(taps - 1) // 2, out=y)
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    ops.conv_gemm(x, w, b, S, taps=taps, pad=(taps - 1) // 2, out=y)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
print(f"M={M} N={N} Cin={Cin} taps={taps}: {ms:.3f} ms {2.0 * M * N * Cin * taps / ms / 1e9:.1f} TF")


