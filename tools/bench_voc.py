"""Dev tool: every distinct convolution launch of the HiFi-GAN V1 generator (hifigan/models.py:113-174) at batch-synthesis size
(B = 8 utterances of T mel frames, default 700), timed alone with HIP events: which launches make up the vocoder's time, at what
MFMA / HBM rate.  usage: python tools/bench_voc.py [T]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fastspeech2_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
B, T = 8, int(sys.argv[1]) if len(sys.argv) > 1 else 700


def timeit(f, reps=10):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            f()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / reps)
    return sorted(ts)[1]


rows = []
tws = ops.tail_workspace(dev)


def conv(name, S, Cin, N, taps, dil=1, in_act=False, res=False, count=1):
    M = B * S
    x = torch.randn(M, Cin, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, taps, Cin, device=dev) / (Cin * taps) ** 0.5).to(torch.bfloat16)
    b = torch.randn(N, device=dev)
    r = torch.randn(M, N, device=dev).to(torch.bfloat16) if res else None
    kw = dict(in_act=ops.ACT_LRELU, in_slope=0.1) if in_act else {}
    f = lambda: ops.conv_gemm(x, w, b, S, taps=taps, dil=dil, pad=(taps - 1) // 2 * dil, res=r, tail_ws=tws, **kw)
    ms = timeit(f)
    fl = 2.0 * M * N * Cin * taps
    by = M * Cin * 2 + M * N * 2 * (2 if res else 1)
    rows.append((name, count, ms, fl, by))
    print(f"{name:34s} x{count:2d}  {ms * 1e3:7.1f} us  {fl / ms / 1e9:7.1f} TF  {by / ms / 1e6:6.0f} GB/s  ({M} rows)", flush=True)


conv("conv_pre 80->512 k7", T, 80, 512, 7)
S, C = T, 512
for i, (u, k) in enumerate(zip((8, 8, 2, 2), (16, 16, 4, 4))):
    cout = C // 2
    conv(f"up{i} {C}->{u}x{cout} (3 taps)", S, C, u * cout, 3, in_act=True)
    S, C = S * u, cout
    if C >= 128:
        for rk in (3, 7, 11):
            for d in (1, 3, 5):
                conv(f"stage{i} C={C} k={rk} conv1 d={d}", S, C, C, rk, dil=d, in_act=True, count=1)
                if os.environ.get("VOC_NO_PROLOGUE"):      # the same launch without the leaky-ReLU prologue: what the prologue costs
                    conv(f"   (no prologue) k={rk} d={d}", S, C, C, rk, dil=d, in_act=False, count=0)
            conv(f"stage{i} C={C} k={rk} conv2 (+res)", S, C, C, rk, res=True, count=3)
tot = sum(c * ms for _, c, ms, _, _ in rows)
print(f"sum over the launches above (x count): {tot:.3f} ms; {sum(c * fl for _, c, _, fl, _ in rows) / tot / 1e9:.0f} TF")
