"""r05zb: how often does tests/test_vocoder_stft_gpu.py::test_resstage_fused_equals_three_block_launches[32-1900] disagree?  Repeats
the test body N times (same inputs: the launches are deterministic functions of them) and counts mismatching repeats, and which side
(three block launches / the stage launch) moved against its own first result."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from fastspeech2_amd import ops

dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
for (C, S) in [(32, 1900), (64, 700)]:
    B, dil = 2, (1, 3, 5)
    g = torch.Generator().manual_seed(C + S)
    x = (torch.randn(B * S, C, generator=g) * 0.7).to(torch.bfloat16).to(dev)
    blocks = []
    for k in (3, 7, 11):
        w1 = (torch.randn(3, C, k, C, generator=g) * (1.0 / (C * k) ** 0.5)).to(torch.bfloat16).to(dev)
        w2 = (torch.randn(3, C, k, C, generator=g) * (1.0 / (C * k) ** 0.5)).to(torch.bfloat16).to(dev)
        blocks.append((w1, w2, (torch.randn(3, C, generator=g) * 0.1).to(dev), (torch.randn(3, C, generator=g) * 0.1).to(dev), k))
    first = None
    n_pair = n_blk = n_stage = 0
    worst = 0.0
    for it in range(N):
        xs = None
        for w1, w2, b1, b2, k in blocks:
            xs = ops.resblock_fwd(x, w1, w2, b1, b2, B, S, k, dil, xs=xs, out_scale=1.0 / 3)
        st = ops.resstage_fwd(x, blocks, B, S, dil)
        if it % 7 == 0:
            torch.cuda.synchronize()
        if first is None:
            torch.cuda.synchronize()
            first = (xs.clone(), st.clone())
        if not torch.equal(st, xs):
            n_pair += 1
            worst = max(worst, float((st.float() - xs.float()).abs().max()))
        n_blk += int(not torch.equal(xs, first[0]))
        n_stage += int(not torch.equal(st, first[1]))
    print(f"C={C} S={S}: {N} repeats: stage != blocks {n_pair} (worst {worst:.4f}); blocks moved {n_blk}; stage moved {n_stage}", flush=True)
