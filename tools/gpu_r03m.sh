#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
FS2_LIB_PATH=$PWD/fastspeech2_amd/libfs2hip_dev.so W_STAMPS=1 timeout 300 python tools/bench_w.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r03o_bench_w.log; cat gpurun_out/r03o_bench_w.log
