"""Micro-benchmarks of the individual HIP kernels at config-2 shapes (B=48, L=128, T=900). Dev tool."""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fastspeech2_amd import ops

dev = torch.device("cuda:0")


def timeit(fn, n=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    B, T = 48, 900
    M = B * T
    for dtype in (torch.float32, torch.bfloat16):
        print("dtype", dtype)
        for (name, Cin, Cout, k) in [("w_1 k9", 256, 1024, 9), ("w_2 k1", 1024, 256, 1), ("qkv", 256, 768, 1), ("fc", 256, 256, 1),
                                     ("postnet k5", 512, 512, 5), ("mel", 256, 80, 1)]:
            x = torch.randn(M, Cin, device=dev).to(dtype)
            w = torch.randn(Cout, k, Cin, device=dev) / math.sqrt(Cin * k)
            b = torch.randn(Cout, device=dev)
            wf, wd = ops.pack_weight(w, dtype)
            y = torch.empty(M, Cout, device=dev, dtype=dtype)
            ms = timeit(lambda: ops.conv_gemm(x, wf, b, T, taps=k, pad=(k - 1) // 2, out=y))
            fl = 2.0 * M * Cin * Cout * k
            print(f"  fwd {name:12s} {ms:8.3f} ms  {fl / ms / 1e9:8.1f} TF")
            dy = torch.randn(M, Cout, device=dev).to(dtype)
            dw = torch.zeros(Cout, k, Cin, device=dev)
            ms = timeit(lambda: ops.conv_wgrad(dy, x, dw, T, taps=k, pad=(k - 1) // 2))
            print(f"  wgrad {name:10s} {ms:8.3f} ms  {fl / ms / 1e9:8.1f} TF")
        H = 2
        qkv = torch.randn(M, 768, device=dev).to(dtype)
        lens = torch.full((B,), T, device=dev, dtype=torch.int32)
        ms = timeit(lambda: ops.attn_fwd(qkv, lens, B, T, H))
        fl = 4.0 * B * H * T * T * 128
        print(f"  attn fwd {ms:8.3f} ms {fl / ms / 1e9:8.1f} TF")
        ctx, lse = ops.attn_fwd(qkv, lens, B, T, H)
        dctx = torch.randn_like(ctx)
        ms = timeit(lambda: ops.attn_bwd(qkv, ctx, dctx, lse, lens, B, T, H))
        print(f"  attn bwd {ms:8.3f} ms {fl * 3.5 / ms / 1e9:8.1f} TF(7 gemms)")
        x = torch.randn(M, 256, device=dev).to(dtype); r = torch.randn(M, 256, device=dev).to(dtype)
        g = torch.ones(256, device=dev); bb = torch.zeros(256, device=dev)
        ms = timeit(lambda: ops.ln_fwd(x, r, g, bb, lens, B, T, p_pre=0.2, seed_pre=1))
        by = M * 256 * x.element_size() * 4
        print(f"  ln fwd {ms:8.3f} ms {by / ms / 1e6:8.1f} GB/s")


if __name__ == "__main__":
    main()
