#!/bin/bash
# r06x: one-tap weight gradient with 128 (n) x 256 (c) workgroup tiles (dev switch FS2_WGRAD_TG1_WIDE) - parity, per-shape bench, step A/B
export TMPDIR=/tmp
mkdir -p gpurun_out
( FS2_LIB_PATH=fastspeech2_amd/libfs2hip_dev.so FS2_WGRAD_TG1_WIDE=1 timeout 900 python -m pytest tests/test_a_prodshape_gpu.py tests/test_ops_gpu.py -x -q -m gpu -k "wgrad or grads or weight" ) 2>&1 | grep -v amdgpu.ids | tail -3 | cut -c1-500 | tee gpurun_out/r06x_pytest_wide.log
for v in 0 1 0 1; do echo "FS2_WGRAD_TG1_WIDE=$v"; FS2_LIB_PATH=fastspeech2_amd/libfs2hip_dev.so FS2_WGRAD_TG1_WIDE=$v python tools/bench_wgrad.py k1 2>&1 | grep -v amdgpu.ids | grep "S=\|per step"; FS2_LIB_PATH=fastspeech2_amd/libfs2hip_dev.so FS2_WGRAD_TG1_WIDE=$v python tools/bench_wgrad.py qkv 2>&1 | grep "S="; FS2_LIB_PATH=fastspeech2_amd/libfs2hip_dev.so FS2_WGRAD_TG1_WIDE=$v python tools/bench_wgrad.py fc 2>&1 | grep "S="; done | tee gpurun_out/r06x_bench_wgrad_wide.log
python tools/ab_env.py "" FS2_WGRAD_TG1_WIDE=1 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06x_ab_env.log
