#!/bin/bash
TAG=${1:-r03j}
export TMPDIR=/tmp
mkdir -p gpurun_out
( time timeout 900 python -m pytest tests/test_a_prodshape_gpu.py -q -s -k "full_size" ) > gpurun_out/${TAG}_pytest_bars.log 2>&1; grep -E "valid-frame|ratios|median|passed|failed|^E  " gpurun_out/${TAG}_pytest_bars.log | cut -c1-1200
( time timeout 900 python -m pytest tests/test_graph_gpu.py tests/test_bench_contract_gpu.py -q -s -k "graph or libritts" ) > gpurun_out/${TAG}_pytest_misc.log 2>&1; grep -E "eager-vs|passed|failed|^E  " gpurun_out/${TAG}_pytest_misc.log | cut -c1-600
