#!/bin/bash
TAG=${1:-r05e}
export TMPDIR=/tmp
mkdir -p gpurun_out
( time timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_a_prodshape_gpu.py -x -q -k "weight_gradient" ) > gpurun_out/${TAG}_pytest_wgrad.log 2>&1; tail -5 gpurun_out/${TAG}_pytest_wgrad.log | cut -c1-400
timeout 600 python tools/bench_wgrad.py k1 > gpurun_out/${TAG}_bench_wgrad1.log 2>&1; cat gpurun_out/${TAG}_bench_wgrad1.log | cut -c1-200
timeout 300 python tools/bench_wgrad.py qkv >> gpurun_out/${TAG}_bench_wgrad1.log 2>&1; timeout 300 python tools/bench_wgrad.py fc >> gpurun_out/${TAG}_bench_wgrad1.log 2>&1; timeout 300 python tools/bench_wgrad.py mel >> gpurun_out/${TAG}_bench_wgrad1.log 2>&1; tail -12 gpurun_out/${TAG}_bench_wgrad1.log | cut -c1-200
for i in 1 2 3; do
  timeout 300 python bench.py --no-cpu-baseline --no-fp32 --no-synth --no-graph-line --no-roofline > gpurun_out/${TAG}_bench_$i.log 2>&1; tail -1 gpurun_out/${TAG}_bench_$i.log | cut -c1-160
done
