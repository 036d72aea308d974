#!/bin/bash
# r06m: software-pipelined dK/dV attention kernel (v2) - parity, then A/B against the phase-serial form (dev library switch)
export TMPDIR=/tmp
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "attention or attn" ) 2>&1 | grep -v amdgpu.ids | tail -4 | cut -c1-800 | tee gpurun_out/r06m_pytest_attn.log
for v in 1 2 1 2; do echo "FS2_ATTN_DKV=$v"; FS2_LIB_PATH=fastspeech2_amd/libfs2hip_dev.so FS2_ATTN_DKV=$v python tools/bench_attn.py 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/r06m_bench_attn.log
