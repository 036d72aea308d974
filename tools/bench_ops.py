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


def vocoder_shapes(dtype=torch.bfloat16):
    """HiFi-GAN V1 layer shapes at B=8, T=900 mel frames (bench.py --mode synth)."""
    B, T = 8, 900
    print("vocoder shapes", dtype)
    tot = 0.0
    for (name, C, k, d, up) in [("pre 80->512 k7", 512, 7, 1, 1), ("rb C256 k3", 256, 3, 1, 8), ("rb C256 k7", 256, 7, 3, 8), ("rb C256 k11", 256, 11, 5, 8),
                                ("rb C128 k3", 128, 3, 1, 64), ("rb C128 k7", 128, 7, 1, 64), ("rb C128 k11", 128, 11, 1, 64),
                                ("rb C64 k3", 64, 3, 1, 128), ("rb C64 k11", 64, 11, 1, 128), ("rb C32 k3", 32, 3, 1, 256), ("rb C32 k11", 32, 11, 5, 256)]:
        S = T * up
        M = B * S
        Cin = 80 if name.startswith("pre") else C
        x = torch.randn(M, Cin, device=dev).to(dtype)
        w = (torch.randn(C, k, Cin, device=dev) / math.sqrt(Cin * k)).to(dtype)
        b = torch.randn(C, device=dev)
        y = torch.empty(M, C, device=dev, dtype=dtype)
        for in_act in ((0,) if name.startswith("pre") else (3, 0)):
            ms = timeit(lambda: ops.conv_gemm(x, w, b, S, taps=k, dil=d, pad=(k * d - d) // 2, out=y, in_act=in_act, in_slope=0.1))
            fl = 2.0 * M * Cin * C * k
            by = M * (Cin + C) * x.element_size()
            print(f"  {name:16s} in_act={in_act} M={M:8d} {ms:8.3f} ms  {fl / ms / 1e9:8.1f} TF  {by / ms / 1e6:8.1f} GB/s(min traffic)")


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "voc":
        return vocoder_shapes()
    B, T = 48, 900
    M = B * T
    for dtype in ((torch.bfloat16,) if len(sys.argv) > 1 and sys.argv[1] == "bf16" else (torch.float32, torch.bfloat16)):
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
            if k > 1 or Cin != Cout:
                dy_ = torch.randn(M, Cout, device=dev).to(dtype)
                dx = torch.empty(M, Cin, device=dev, dtype=dtype)
                ms = timeit(lambda: ops.conv_gemm(dy_, wd, None, T, taps=k, pad=(k - 1) - (k - 1) // 2, out=dx))
                print(f"  dgrad {name:10s} {ms:8.3f} ms  {fl / ms / 1e9:8.1f} TF")
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
