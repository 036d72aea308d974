#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_a_prodshape_gpu.py -x -q -m gpu -k "contraction" 2>&1 | tail -3 ) > gpurun_out/r02q_pytest.log 2>&1
tail -2 gpurun_out/r02q_pytest.log | cut -c1-300
bash tools/ab_step.sh > gpurun_out/r02q_ab.log 2>&1; cat gpurun_out/r02q_ab.log
python tools/bench_p.py 2>&1 | tail -1 | tr ';' '\n' | grep -E "dgrad"
