#!/bin/bash
# kernel trace of the eager train step -> gpurun_out/prof_summary_side{0,1}.md (per kernel + per launch geometry + idle gaps)
export TMPDIR=/tmp
for side in ${1:-0}; do
  rm -rf gpurun_out/prof; mkdir -p gpurun_out/prof
  FS2_SIDE_STREAM=$side timeout 300 rocprofv3 --kernel-trace -d gpurun_out/prof -o bench -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-roofline > gpurun_out/prof.log 2>&1
  DB=$(find gpurun_out/prof -name '*.db' | head -1)
  python tools/rocpd_summary.py $DB 10 shapes > gpurun_out/prof_summary_side${side}.md 2>&1
done
rm -rf gpurun_out/prof
