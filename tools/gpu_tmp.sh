#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_a_prodshape_gpu.py -x -q -m gpu -k "contraction or vocoder or splitk" 2>&1 | tail -5 ) > gpurun_out/r02t_pytest.log 2>&1
tail -4 gpurun_out/r02t_pytest.log | cut -c1-500
python tools/bench_p.py 2>&1 | tail -1 | tr ';' '\n' > gpurun_out/r02t_bench_p.txt; cat gpurun_out/r02t_bench_p.txt
