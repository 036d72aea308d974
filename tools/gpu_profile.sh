#!/bin/bash
# kernel trace of the eager train step -> gpurun_out/prof_summary.md (per kernel + per launch geometry)
export TMPDIR=/tmp
rm -rf gpurun_out/prof; mkdir -p gpurun_out/prof
timeout 300 rocprofv3 --kernel-trace -d gpurun_out/prof -o bench -- python bench.py --steps 6 --warmup 2 --graph 0 --no-cpu-baseline --no-roofline > gpurun_out/prof.log 2>&1
DB=$(find gpurun_out/prof -name '*.db' | head -1)
python tools/rocpd_summary.py $DB 10 shapes > gpurun_out/prof_summary.md 2>&1
rm -rf gpurun_out/prof
