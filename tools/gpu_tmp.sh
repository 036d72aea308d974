#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
python tools/bench_p.py ab FS2_P_TKS=1 FS2_P_TKS=8 2>&1 | tail -16 > gpurun_out/r02n_bench_p2.md; cat gpurun_out/r02n_bench_p2.md
bash tools/ab_step.sh > gpurun_out/r02n_ab.log 2>&1; cat gpurun_out/r02n_ab.log
