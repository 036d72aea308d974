#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 600 python bench.py --no-cpu-baseline --no-fp32 --no-synth > gpurun_out/r04c_bench_bf16.log 2>&1; tail -1 gpurun_out/r04c_bench_bf16.log | cut -c1-330
