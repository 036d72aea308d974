#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests -m gpu -q ) > gpurun_out/r03r_pytest_full.log 2>&1; tail -6 gpurun_out/r03r_pytest_full.log | cut -c1-300
timeout 600 python bench.py > gpurun_out/r03r_bench_bf16.log 2>&1; tail -1 gpurun_out/r03r_bench_bf16.log | cut -c1-400
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
