"""Dev tool: where does the 8-consumer-wave persistent kernel differ from the 4-wave one?  (dev library, FS2_P_CW)"""
import os, subprocess, sys, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

def run(tag):
    import torch
    from fastspeech2_amd import ops
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    S, B, Cin, Cout, k = 925, 48, 256, 1024, 9
    M = B * S
    x = torch.randn(M, Cin, device=dev).to(torch.bfloat16)
    w = (torch.randn(Cout, k, Cin, device=dev) / math.sqrt(Cin * k)).to(torch.bfloat16)
    bias = torch.randn(Cout, device=dev)
    y = ops.conv_gemm(x, w, bias, S, taps=k, pad=(k - 1) // 2, act=ops.ACT_RELU)
    torch.cuda.synchronize()
    torch.save(y.float().cpu(), f"/tmp/cw_{tag}.pt")

if len(sys.argv) > 1:
    run(sys.argv[1])
else:
    import torch
    for cw in ("4", "8"):
        e = dict(os.environ, FS2_LIB_PATH=os.path.join(ROOT, "fastspeech2_amd", "libfs2hip_dev.so"), FS2_P_CW=cw)
        subprocess.run([sys.executable, os.path.abspath(__file__), cw], env=e, check=True)
    a, b = torch.load("/tmp/cw_4.pt"), torch.load("/tmp/cw_8.pt")
    bad = (a - b).abs() > 1e-2
    print("bad", int(bad.sum()), "of", bad.numel())
    rows = bad.any(1).nonzero().flatten()
    cols = bad.any(0).nonzero().flatten()
    print("bad rows", len(rows), "first", rows[:40].tolist())
    print("rows mod 256 histogram (32-row blocks):", torch.bincount((rows % 256) // 32, minlength=8).tolist())
    print("row tiles with bad rows:", torch.unique(rows // 256)[:40].tolist(), "count", len(torch.unique(rows // 256)))
    print("bad cols", len(cols), "cols mod 128 // 8 histogram:", torch.bincount((cols % 128) // 8, minlength=16).tolist())
    print("col tiles:", torch.unique(cols // 128).tolist())
    r0 = int(rows[0])
    print("row", r0, "bad cols:", bad[r0].nonzero().flatten()[:64].tolist())
    cs = bad[r0].nonzero().flatten()[:8].tolist()
    print("a", a[r0, cs].tolist()); print("b", b[r0, cs].tolist())
    torch.manual_seed(0)
    import math
    bias = None
    # per row tile: fraction of bad elements in the nb = 3 column blocks
    t = bad.view(-1, 925 * 48 // 1, 1024) if False else None
    for mt in (31, 32, 33, 100, 173):
        blk = bad[mt * 256:(mt + 1) * 256]
        print("row tile", mt, "bad per 32-row block:", [int(blk[i * 32:(i + 1) * 32].sum()) for i in range(8)], "bad rows within block 0 (mod 32):", blk[:32].any(1).nonzero().flatten().tolist())
    d = (b - a)[bad]
    print("diff stats: mean abs", float(d.abs().mean()), "max", float(d.abs().max()), "frac where cw8 == 0:", float((b[bad] == 0).float().mean()), "frac where cw4 == 0:", float((a[bad] == 0).float().mean()))
    # is a bad element ever in a row < 8192?
    print("min bad row", int(rows.min()), "max bad row", int(rows.max()))
    nbad_by_tile = torch.stack([bad[i * 256:(i + 1) * 256].sum() for i in range(174)])
    print("bad per row tile:", nbad_by_tile.tolist())
