"""Dev tool: weight-gradient kernel at the train-step shapes (decoder T=925 and encoder L=128 rows per sequence)."""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fastspeech2_amd import ops
from tools.bench_ops import timeit

dev = torch.device("cuda:0")
B = 48
for S in (925, 128):
    M = B * S
    lens = torch.randint(int(S * 0.75), S + 1, (B,), device=dev, dtype=torch.int32)
    for (name, Cin, Cout, k) in [("w_1 k9", 256, 1024, 9), ("w_2 k1", 1024, 256, 1), ("qkv", 256, 768, 1), ("fc", 256, 256, 1),
                                 ("postnet k5", 512, 512, 5), ("pred k3", 256, 256, 3), ("mel", 256, 80, 1)]:
        if S == 128 and name in ("postnet k5", "mel"):
            continue
        x = torch.randn(M, Cin, device=dev).to(torch.bfloat16)
        dy = torch.randn(M, Cout, device=dev).to(torch.bfloat16)
        dw = torch.zeros(Cout, k, Cin, device=dev)
        db = torch.zeros(Cout, device=dev)
        ms = timeit(lambda: ops.conv_wgrad(dy, x, dw, S, taps=k, pad=(k - 1) // 2, lens=lens, dbias=db), n=20)
        fl = 2.0 * M * Cin * Cout * k
        print(f"  S={S:4d} wgrad {name:10s} {ms * 1e3:8.1f} us  {fl / ms / 1e9:8.1f} TF")
