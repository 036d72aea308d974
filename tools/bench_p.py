"""Dev tool: the train step's large contractions at production shapes (B=48, T=925 / L=128, the bench batch's own ragged
lengths) - persistent kernel vs ring / 128^2 kernels, with the dev library's ablation switches.

    python tools/bench_p.py                 one process, current environment
    python tools/bench_p.py sweep           re-runs itself under FS2_LIB_PATH=libfs2hip_dev.so with FS2_GEMM_P / FS2_GEMM_ABL /
                                            FS2_GEMM_DBG settings and prints one table (A/B on the same box)
"""
import math
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SHAPES = [("w_1 k9 fwd", 256, 1024, 9, 925), ("w_1 k9 dgrad", 1024, 256, 9, 925), ("postnet k5", 512, 512, 5, 925),
          ("w_2 k1 fwd", 1024, 256, 1, 925), ("w_2 k1 dgrad", 256, 1024, 1, 925), ("qkv fwd", 256, 768, 1, 925),
          ("qkv dgrad", 768, 256, 1, 925), ("fc", 256, 256, 1, 925), ("mel", 256, 80, 1, 925), ("postnet out", 512, 80, 5, 925),
          ("enc w_1 k9", 256, 1024, 9, 128), ("enc w_1 dgrad", 1024, 256, 9, 128)]


def run():
    import torch
    from fastspeech2_amd import ops
    from fastspeech2_amd.synthetic import synthetic_batch
    dev = torch.device("cuda:0")
    b = synthetic_batch(1234, 48, 128, dur_lo=4, dur_hi=10, min_len_frac=0.75)
    lens_by_S = {b["max_mel_len"]: b["mel_lens"].to(torch.int32).to(dev), 128: b["src_lens"].to(torch.int32).to(dev)}
    T = b["max_mel_len"]
    out = []
    tws = ops.tail_workspace(dev)                                  # as the engine passes it (dev library: FS2_P_TKS=1 turns the split off)
    for (name, Cin, Cout, k, S) in SHAPES:
        S = T if S == 925 else S
        M = 48 * S
        lens = lens_by_S[S]
        tmap = ops.tile_map(lens, 48, S)
        x = torch.randn(M, Cin, device=dev).to(torch.bfloat16)
        w = (torch.randn(Cout, k, Cin, device=dev) / math.sqrt(Cin * k)).to(torch.bfloat16)
        bias = torch.randn(Cout, device=dev)
        y = torch.empty(M, Cout, device=dev, dtype=torch.bfloat16)
        fn = lambda: ops.conv_gemm(x, w, bias, S, taps=k, pad=(k - 1) // 2, act=ops.ACT_RELU, lens=lens, tmap=tmap, out=y, tail_ws=tws)
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                fn()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 10)
        out.append((name, best * 1e3, 2.0 * M * Cout * Cin * k / best / 1e9))
    print("RESULT " + ";".join(f"{n}|{us:.1f}|{tf:.0f}" for n, us, tf in out), flush=True)


def sweep(custom=None):
    cfgs = [(c, dict(kv.split("=") for kv in c.split(",") if kv)) for c in custom] if custom else [("ring/128 (P off)", {"FS2_GEMM_P": "0"}), ("persistent", {}), ("P no-MFMA", {"FS2_GEMM_ABL": "1"}),
            ("P no-reads", {"FS2_GEMM_ABL": "2"}), ("P no-MFMA no-reads", {"FS2_GEMM_ABL": "3"}), ("P no-epilogue", {"FS2_GEMM_ABL": "4"}),
            ]
    rows = {}
    for tag, env in cfgs:
        e = dict(os.environ)
        e["FS2_LIB_PATH"] = os.path.join(ROOT, "fastspeech2_amd", "libfs2hip_dev.so")
        e.update(env)
        p = subprocess.run([sys.executable, os.path.abspath(__file__)], env=e, capture_output=True, text=True, timeout=600)
        line = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")]
        if not line:
            print(f"{tag}: FAILED\n{p.stdout[-2000:]}\n{p.stderr[-2000:]}")
            continue
        rows[tag] = [x.split("|") for x in line[0][7:].split(";")]
    names = [r[0] for r in next(iter(rows.values()))]
    print("| shape | " + " | ".join(rows) + " |")
    print("|---|" + "---|" * len(rows))
    for i, n in enumerate(names):
        print(f"| {n} | " + " | ".join(f"{rows[t][i][1]} us / {rows[t][i][2]} TF" for t in rows) + " |")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "sweep":
        sweep()
    elif len(sys.argv) > 1 and sys.argv[1] == "ab":          # python tools/bench_p.py ab FS2_P_ORDER=0 FS2_P_ORDER=1,FS2_X=2 ...
        sweep(sys.argv[2:])
    else:
        run()
