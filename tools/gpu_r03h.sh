#!/bin/bash
TAG=${1:-r03h}
export TMPDIR=/tmp
mkdir -p gpurun_out
DEV=fastspeech2_amd/libfs2hip_dev.so
( time timeout 900 python -m pytest tests/test_a_prodshape_gpu.py -q -x -k "contraction and (qkv or fc or w_2)" ) > gpurun_out/${TAG}_pytest_w.log 2>&1; tail -12 gpurun_out/${TAG}_pytest_w.log | cut -c1-300
( time timeout 600 python -m pytest tests/test_ops_gpu.py -q -k "conv_gemm" ) > gpurun_out/${TAG}_pytest_ops.log 2>&1; tail -4 gpurun_out/${TAG}_pytest_ops.log | cut -c1-300
for w in 0 1; do echo "== FS2_GEMM_W=$w"; FS2_LIB_PATH=$DEV FS2_GEMM_W=$w timeout 300 python tools/bench_w.py 2>&1 | grep -v amdgpu.ids; done > gpurun_out/${TAG}_bench_w.log 2>&1; cat gpurun_out/${TAG}_bench_w.log
timeout 1500 python tools/ab_env.py "" FS2_GEMM_W=0 > gpurun_out/${TAG}_ab_env.log 2>&1; cat gpurun_out/${TAG}_ab_env.log
