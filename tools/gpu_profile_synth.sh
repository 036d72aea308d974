#!/bin/bash
export TMPDIR=/tmp
rm -rf gpurun_out/prof; mkdir -p gpurun_out/prof
timeout 300 rocprofv3 --kernel-trace -d gpurun_out/prof -o bench -- python bench.py --mode synth --steps 6 --warmup 2 > gpurun_out/prof.log 2>&1
DB=$(find gpurun_out/prof -name '*.db' | head -1)
python tools/rocpd_summary.py $DB 8 shapes > gpurun_out/prof_summary_synth.md 2>&1
rm -rf gpurun_out/prof
