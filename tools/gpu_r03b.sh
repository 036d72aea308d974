#!/bin/bash
# round 3, call B: the new weight-gradient path - parity first (short timeouts: a hang must not eat the box), then A/B timing,
# then the step A/B and the bench line
TAG=${1:-r03b}
export TMPDIR=/tmp
mkdir -p gpurun_out
( time timeout 600 python -m pytest tests/test_ops_gpu.py -q -x -k "weight_gradient_workspace or conv_grads" ) > gpurun_out/${TAG}_pytest_wgrad.log 2>&1; tail -15 gpurun_out/${TAG}_pytest_wgrad.log
( time timeout 600 python -m pytest tests/test_a_prodshape_gpu.py -q -x -k "weight_gradient" ) > gpurun_out/${TAG}_pytest_wgrad_prod.log 2>&1; tail -8 gpurun_out/${TAG}_pytest_wgrad_prod.log
timeout 600 python tools/bench_wgrad.py > gpurun_out/${TAG}_bench_wgrad.log 2>&1; cat gpurun_out/${TAG}_bench_wgrad.log
( time timeout 900 python -m pytest tests/test_graph_gpu.py tests/test_model_gpu.py tests/test_fullsize_gpu.py -q -x ) > gpurun_out/${TAG}_pytest_model.log 2>&1; tail -8 gpurun_out/${TAG}_pytest_model.log
timeout 600 python bench.py --no-cpu-baseline --no-fp32 --no-synth > gpurun_out/${TAG}_bench_bf16.log 2>&1; tail -1 gpurun_out/${TAG}_bench_bf16.log | cut -c1-3000
