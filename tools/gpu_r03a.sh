#!/bin/bash
# round 3, call A: the new config-5 pins, graph test, bench contract, one full bench line
TAG=${1:-r03a}
export TMPDIR=/tmp
mkdir -p gpurun_out
( time timeout 1200 python -m pytest tests/test_a_prodshape_gpu.py -q -s -k "skinny or polyphase or conv_post or per_stage" ) > gpurun_out/${TAG}_pytest_voc.log 2>&1; tail -25 gpurun_out/${TAG}_pytest_voc.log
( time timeout 900 python -m pytest tests/test_graph_gpu.py tests/test_bench_contract_gpu.py -q ) > gpurun_out/${TAG}_pytest_bench.log 2>&1; tail -25 gpurun_out/${TAG}_pytest_bench.log
timeout 600 python bench.py > gpurun_out/${TAG}_bench_bf16.log 2>&1; tail -1 gpurun_out/${TAG}_bench_bf16.log | cut -c1-4000
