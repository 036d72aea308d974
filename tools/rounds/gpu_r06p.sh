#!/bin/bash
# r06p: lane-constant fragment addresses in the forward and dQ attention kernels - parity, bench against the previous commit's library
export TMPDIR=/tmp
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "attention or attn" ) 2>&1 | grep -v amdgpu.ids | tail -3 | cut -c1-800 | tee gpurun_out/r06p_pytest_attn.log
for l in fastspeech2_amd/libfs2hip_prev.so fastspeech2_amd/libfs2hip.so fastspeech2_amd/libfs2hip_prev.so fastspeech2_amd/libfs2hip.so; do echo $l; FS2_LIB_PATH=$l python tools/bench_attn.py 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/r06p_bench_attn.log
