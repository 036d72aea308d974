"""Dev tool: run ONE weight-gradient shape a few times (for PMC passes).  usage: one_wgrad.py S Cin Cout k [reps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fastspeech2_amd import ops
S, Cin, Cout, k = [int(v) for v in sys.argv[1:5]]
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 5
dev = torch.device("cuda:0")
B = 48
M = B * S
torch.manual_seed(0)
lens = torch.randint(int(S * 0.75), S + 1, (B,), device=dev, dtype=torch.int32)
x = torch.randn(M, Cin, device=dev).to(torch.bfloat16)
dy = torch.randn(M, Cout, device=dev).to(torch.bfloat16)
dw = torch.zeros(Cout, k, Cin, device=dev)
db = torch.zeros(Cout, device=dev)
for _ in range(reps):
    ops.conv_wgrad(dy, x, dw, S, taps=k, pad=(k - 1) // 2, lens=lens, dbias=db)
torch.cuda.synchronize()
