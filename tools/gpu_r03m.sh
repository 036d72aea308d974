#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
for g in 0 256; do echo "FS2_W_G=$g"; FS2_LIB_PATH=$PWD/fastspeech2_amd/libfs2hip_dev.so FS2_W_G=$g timeout 300 python tools/bench_w.py 2>&1 | grep -v amdgpu.ids; done > gpurun_out/r03q_bench_w.log; cat gpurun_out/r03q_bench_w.log
timeout 900 python tools/ab_env.py "" FS2_W_G=256 > gpurun_out/r03q_ab_env.log 2>&1; cat gpurun_out/r03q_ab_env.log
