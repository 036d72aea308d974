"""Dev tool: fc projection + dropout + residual + LayerNorm at the decoder's shape (B = 48, T = 925): one launch (fs2_gemm_res_ln_fwd)
against two (fs2_conv_gemm + fs2_ln_fwd), HIP events."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fastspeech2_amd import ops, _lib

dev = torch.device("cuda:0")
B, S, K, N = 48, int(os.environ.get("BENCH_S", "925")), 256, 256
M = B * S
g = torch.Generator().manual_seed(1)
lens = torch.sort(torch.randint(int(S * 0.75), S + 1, (B,), generator=g), descending=True)[0].to(torch.int32).to(dev)
tmap = ops.tile_map(lens, B, S)
x = torch.randn(M, K, generator=g).to(dev).to(torch.bfloat16)
w = (torch.randn(N, 1, K, generator=g) / math.sqrt(K)).to(dev).to(torch.bfloat16)
bias = torch.randn(N, generator=g).to(dev)
res = torch.randn(M, N, generator=g).to(dev).to(torch.bfloat16)
gamma = torch.ones(N, device=dev); beta = torch.zeros(N, device=dev)
print("streams:", _lib.load().fs2_gemm_res_ln_streams(M, N, K, S, 1))


def timeit(f):
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            f()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 10 * 1e3)
    return sorted(ts)[2]


for p in (0.0, 0.2):
    one = lambda: ops.gemm_res_ln(x, w, bias, res, gamma, beta, lens, tmap, B, S, p_pre=p, seed_pre=5)
    def two():
        y = ops.conv_gemm(x, w, bias, S)
        return ops.ln_fwd(y, res, gamma, beta, lens, B, S, p_pre=p, seed_pre=5)
    def two_lens():
        y = ops.conv_gemm(x, w, bias, S, lens=lens, tmap=tmap)
        return ops.ln_fwd(y, res, gamma, beta, lens, B, S, p_pre=p, seed_pre=5)
    print(f"p={p}: one launch {timeit(one):.1f} us   two launches {timeit(two):.1f} us   (with lens on the contraction {timeit(two_lens):.1f} us)", flush=True)
