#!/bin/bash
# round 2, call A: persistent kernel correctness first (short timeouts: a hang must not eat the box), A/B + ablation sweep,
# production-shape parity suite, the whole -m gpu suite, bench line, kernel trace.
TAG=${1:-r02x}
export TMPDIR=/tmp
mkdir -p gpurun_out
echo "== contraction parity (persistent kernel)"; ( time timeout 420 python -m pytest tests/test_a_prodshape_gpu.py -x -q -k contraction ) > gpurun_out/${TAG}_pytest_contraction.log 2>&1; tail -5 gpurun_out/${TAG}_pytest_contraction.log
echo "== sweep"; timeout 600 python tools/bench_p.py sweep > gpurun_out/${TAG}_bench_p.md 2>&1; cat gpurun_out/${TAG}_bench_p.md
if [ -n "$QUICK" ]; then
  timeout 600 python bench.py --no-cpu-baseline --no-fp32 > gpurun_out/${TAG}_bench_bf16.log 2>&1; tail -1 gpurun_out/${TAG}_bench_bf16.log | cut -c1-1800
  exit 0
fi
echo "== production-shape suite"; ( time timeout 900 python -m pytest tests/test_a_prodshape_gpu.py -x -q -s -k "not contraction" ) > gpurun_out/${TAG}_pytest_prodshape.log 2>&1; grep -E "rel-Frobenius|L1|passed|failed|Error|assert" gpurun_out/${TAG}_pytest_prodshape.log | tail -20
echo "== full suite"; ( time timeout 1200 python -m pytest tests -m gpu -q --deselect tests/test_a_prodshape_gpu.py ) > gpurun_out/${TAG}_pytest.log 2>&1; tail -15 gpurun_out/${TAG}_pytest.log
echo "== bench"; timeout 600 python bench.py > gpurun_out/${TAG}_bench_bf16.log 2>&1; tail -1 gpurun_out/${TAG}_bench_bf16.log | cut -c1-1500
rm -rf gpurun_out/prof; mkdir -p gpurun_out/prof
timeout 300 rocprofv3 --kernel-trace -d gpurun_out/prof -o bench -- python bench.py --steps 6 --warmup 2 --windows 1 --side-stream 0 --no-cpu-baseline --no-roofline --no-fp32 > gpurun_out/prof.log 2>&1
DB=$(find gpurun_out/prof -name '*.db' | head -1)
python tools/rocpd_summary.py $DB 10 shapes > gpurun_out/${TAG}_kernel_trace_side0.md 2>&1
rm -rf gpurun_out/prof
head -40 gpurun_out/${TAG}_kernel_trace_side0.md
