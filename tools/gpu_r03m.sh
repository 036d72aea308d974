#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
for g in 0 64 128 256; do echo "FS2_W_STAGGER=$g"; FS2_LIB_PATH=$PWD/fastspeech2_amd/libfs2hip_dev.so FS2_W_STAGGER=$g timeout 300 python tools/bench_w.py 2>&1 | grep -v amdgpu.ids | grep -E "qkv fwd|w_2 dgrad|per dec"; done > gpurun_out/r03q_bench_w_stagger.log; cat gpurun_out/r03q_bench_w_stagger.log
