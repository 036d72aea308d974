#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests/test_graph_gpu.py tests/test_cli_gpu.py tests/test_ops_gpu.py -m gpu -q -k "graph or cli or lrelu_prologue" ) > gpurun_out/r03n_pytest_fix.log 2>&1; tail -5 gpurun_out/r03n_pytest_fix.log | cut -c1-300
