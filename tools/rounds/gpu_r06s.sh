#!/bin/bash
# r06s: kernel trace of the LibriTTS-shaped step (46 % valid rows) - which kernels do not shrink with the padding?
export TMPDIR=/tmp
cd /root/repo
mkdir -p gpurun_out
rm -rf gpurun_out/prof; mkdir -p gpurun_out/prof
timeout 300 rocprofv3 --kernel-trace -d gpurun_out/prof -o bench -- python bench.py --workload libritts --steps 6 --warmup 2 --no-cpu-baseline --no-roofline --no-synth --no-fp32 --no-graph-line --side-stream 0 > gpurun_out/prof.log 2>&1
DB=$(find gpurun_out/prof -name '*.db' | head -1)
python tools/rocpd_summary.py $DB 10 shapes > gpurun_out/r06s_kernel_trace_libritts_side0.md 2>&1
rm -rf gpurun_out/prof
head -45 gpurun_out/r06s_kernel_trace_libritts_side0.md
