"""Dev tool: the one-tap contractions of a decoder FFT block (B=48, T=925) forward and data gradient, timed with HIP events;
run it under FS2_LIB_PATH=...dev.so with FS2_GEMM_W=0 / 1 to compare the wide-tile kernel with the persistent 256x128 one."""
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
tmap = ops.tile_map(lens, B, S)
tws = ops.tail_workspace(dev)
# (name, K, N, residual, gate)
SHAPES = [("qkv fwd", 256, 768, False, False), ("fc fwd", 256, 256, False, False), ("w_2 fwd", 1024, 256, False, False),
          ("w_2 dgrad+gate", 256, 1024, True, True), ("fc dgrad", 256, 256, False, False), ("qkv dgrad+res", 768, 256, True, False)]
tot = 0.0
for name, K, N, res, gate in SHAPES:
    x = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, 1, K, device=dev) / math.sqrt(K)).to(torch.bfloat16)
    bias = torch.randn(N, device=dev) if "fwd" in name else None
    r = torch.randn(M, N, device=dev).to(torch.bfloat16) if res else None
    f = lambda: ops.conv_gemm(x, w, bias, S, lens=lens, res=r, act=ops.ACT_GATE if gate else ops.ACT_NONE, tmap=tmap, tail_ws=tws)
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
    var = _lib.load().fs2_conv_gemm_variant(K, N, N if res else 0, 1, 1, M, N, K, S, 1, 1, 0, 0.0, 1)
    byts = M * (K + N + (N if res else 0)) * 2
    tot += ms
    print(f"  {name:16s} K={K:4d} N={N:4d} variant {var}: {ms * 1e3:7.1f} us  {2.0 * M * K * N / ms / 1e9:7.1f} TF  {byts / ms / 1e9:6.2f} TB/s (unique bytes)", flush=True)
print(f"  per decoder layer: {tot * 1e3:.1f} us -> x4 layers {tot * 4:.3f} ms")
