"""Dev tool: weight-gradient kernels at the train-step shapes (decoder T=925 and encoder L=128 rows per sequence), the round-3
workspace path (LDS-DMA tap-group kernel / slab split-K + finalize) against the round-1/2 atomic path, interleaved rounds in one
process.  The production step's launches (4+4 model, B=48) and their per-step counts give the serial weight-gradient time."""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fastspeech2_amd import ops

dev = torch.device("cuda:0")
B = 48
# (name, Cin, Cout, k, S, launches per step)
SHAPES = [("dec w_1 k9", 256, 1024, 9, 925, 4), ("postnet k5", 512, 512, 5, 925, 3), ("postnet in k5", 80, 512, 5, 925, 1),
          ("postnet out k5", 512, 80, 5, 925, 1), ("dec w_2 k1", 1024, 256, 1, 925, 4), ("dec qkv", 256, 768, 1, 925, 4),
          ("dec fc", 256, 256, 1, 925, 4), ("mel", 256, 80, 1, 925, 1), ("enc w_1 k9", 256, 1024, 9, 128, 4),
          ("enc w_2 k1", 1024, 256, 1, 128, 4), ("enc qkv", 256, 768, 1, 128, 4), ("enc fc", 256, 256, 1, 128, 4),
          ("pred k3", 256, 256, 3, 128, 6)]
only = sys.argv[1] if len(sys.argv) > 1 else ""
rows = []
SHAPES = [r for r in SHAPES if only in r[0]]
for (name, Cin, Cout, k, S, per_step) in SHAPES:
    M = B * S
    g = torch.Generator().manual_seed(1)
    lens = torch.randint(int(S * 0.75), S + 1, (B,), generator=g).to(torch.int32).to(dev)
    has_lens = not name.startswith("pred") and not name.startswith("postnet") and name != "mel"
    x = torch.randn(M, Cin, device=dev).to(torch.bfloat16)
    dy = torch.randn(M, Cout, device=dev).to(torch.bfloat16)
    dw = torch.zeros(Cout, k, Cin, device=dev)
    db = torch.zeros(Cout, device=dev)
    fns = {"ws": lambda: ops.conv_wgrad(dy, x, dw, S, taps=k, pad=(k - 1) // 2, lens=lens if has_lens else None, dbias=db),
           "atomic": lambda: ops.conv_wgrad(dy, x, dw, S, taps=k, pad=(k - 1) // 2, lens=lens if has_lens else None, dbias=db, use_ws=False)}
    best = {}
    for rnd in range(4):
        for key, f in fns.items():
            for _ in range(3):
                f()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                f()
            e1.record()
            torch.cuda.synchronize()
            best.setdefault(key, []).append(e0.elapsed_time(e1) / 10)
    fl = 2.0 * M * Cin * Cout * k
    ws, at = sorted(best["ws"])[1], sorted(best["atomic"])[1]
    rows.append((name, per_step, ws, at, fl))
    print(f"  S={S:4d} wgrad {name:15s} ws {ws * 1e3:7.1f} us {fl / ws / 1e9:7.1f} TF | atomic {at * 1e3:7.1f} us {fl / at / 1e9:7.1f} TF | x{at / ws:5.2f}", flush=True)
tw = sum(r[1] * r[2] for r in rows)
ta = sum(r[1] * r[3] for r in rows)
tf = sum(r[1] * r[4] for r in rows)
print(f"per step (serial, {sum(r[1] for r in rows)} launches): ws {tw:.3f} ms ({tf / tw / 1e9:.0f} TF) | atomic {ta:.3f} ms ({tf / ta / 1e9:.0f} TF)")
