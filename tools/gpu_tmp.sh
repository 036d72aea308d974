export TMPDIR=/tmp; mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_a_prodshape_gpu.py -q -s -x ) > gpurun_out/r02f_pytest_prod.log 2>&1; grep -E "valid-frame|Frobenius|passed|failed|^E " gpurun_out/r02f_pytest_prod.log | cut -c1-900 | tail -14
timeout 120 python - <<'PY'
import math, sys, torch
sys.path.insert(0, ".")
from fastspeech2_amd import ops
dev = torch.device("cuda:0"); B, S, Cin, Cout, k = 48, 128, 1024, 256, 9
M = B * S
x = torch.randn(M, Cin, device=dev).to(torch.bfloat16); w = (torch.randn(Cout, k, Cin, device=dev) / 96).to(torch.bfloat16)
lens = torch.randint(96, 129, (B,), device=dev, dtype=torch.int32); tmap = ops.tile_map(lens, B, S)
res = torch.randn(M, Cout, device=dev).to(torch.bfloat16); ws = torch.zeros(M, Cout, device=dev); y = torch.empty(M, Cout, device=dev, dtype=torch.bfloat16)
for ks in (1, 2, 4):
    fn = lambda: ops.conv_gemm(x, w, None, S, taps=k, pad=4, res=res, lens=lens, tmap=tmap, ksplit=ks, ws=ws if ks > 1 else None, out=y)
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    print(f"enc k9 dgrad ksplit={ks}: {e0.elapsed_time(e1)/20*1e3:.1f} us")
PY
( timeout 900 python -m pytest tests -m gpu -q -x --deselect tests/test_a_prodshape_gpu.py ) > gpurun_out/r02f_pytest.log 2>&1; tail -4 gpurun_out/r02f_pytest.log
for i in 1 2; do timeout 300 python bench.py --no-cpu-baseline --no-fp32 2>&1 | tail -1 | cut -c1-1900; done
