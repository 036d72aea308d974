#!/bin/bash
TAG=${1:-r03c}
export TMPDIR=/tmp
mkdir -p gpurun_out
( time timeout 900 python -m pytest tests/test_ops_gpu.py -q -k "weight_gradient_workspace or conv_grads" ) > gpurun_out/${TAG}_pytest_wgrad.log 2>&1; tail -15 gpurun_out/${TAG}_pytest_wgrad.log | cut -c1-300
( time timeout 600 python -m pytest tests/test_a_prodshape_gpu.py -q -k "weight_gradient" ) > gpurun_out/${TAG}_pytest_wgrad_prod.log 2>&1; tail -8 gpurun_out/${TAG}_pytest_wgrad_prod.log | cut -c1-300
timeout 600 python tools/bench_wgrad.py > gpurun_out/${TAG}_bench_wgrad.log 2>&1; cat gpurun_out/${TAG}_bench_wgrad.log
( time timeout 1500 python -m pytest tests -m gpu -q -x --deselect tests/test_a_prodshape_gpu.py ) > gpurun_out/${TAG}_pytest.log 2>&1; tail -8 gpurun_out/${TAG}_pytest.log | cut -c1-300
( time timeout 1200 python -m pytest tests/test_a_prodshape_gpu.py -q -s ) > gpurun_out/${TAG}_pytest_prodshape.log 2>&1; grep -E "rel-Frobenius|L1|passed|failed|Error|assert|worst" gpurun_out/${TAG}_pytest_prodshape.log | tail -20 | cut -c1-300
for i in 1 2 3; do
( cd _ab && timeout 600 python bench.py --no-cpu-baseline --no-roofline --no-fp32 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('round-2 tree ', d['ms_per_step'], d['value'], d['config']['window_ms_per_step'])" ) | tee -a gpurun_out/${TAG}_ab.log
timeout 600 python bench.py --no-cpu-baseline --no-roofline --no-fp32 --no-synth 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('current tree ', d['ms_per_step'], d['value'], d['config']['window_ms_per_step'])" | tee -a gpurun_out/${TAG}_ab.log
done
timeout 600 python bench.py --no-cpu-baseline --no-fp32 --no-synth > gpurun_out/${TAG}_bench_bf16.log 2>&1; tail -1 gpurun_out/${TAG}_bench_bf16.log | cut -c1-3000
