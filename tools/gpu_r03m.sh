#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_a_prodshape_gpu.py -q -x -k "attention" ) > gpurun_out/r03t_pytest_attn.log 2>&1; tail -3 gpurun_out/r03t_pytest_attn.log | cut -c1-300
timeout 300 python tools/bench_attn.py 2>&1 | grep -v amdgpu.ids
