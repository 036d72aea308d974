#!/bin/bash
# r06j: persistent kernel with EIGHT consumer waves (32 x 128 each, two per SIMD) against the four-wave form - dev library, same box
export TMPDIR=/tmp
mkdir -p gpurun_out
python tools/dbg_cw8.py 2>&1 | grep -v amdgpu.ids | head -3 | cut -c1-300
( FS2_LIB_PATH=fastspeech2_amd/libfs2hip_dev.so FS2_P_CW=8 timeout 900 python -m pytest tests/test_a_prodshape_gpu.py tests/test_ops_gpu.py -x -q -m gpu ) 2>&1 | grep -v amdgpu.ids | tail -3 | cut -c1-600 | tee gpurun_out/r06j_pytest_cw8.log
python tools/bench_p.py ab FS2_P_CW=4 FS2_P_CW=8 FS2_P_CW=4,X=1 FS2_P_CW=8,X=1 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06j_bench_p_cw.log
