#!/bin/bash
TAG=${1:-r03l}
export TMPDIR=/tmp
mkdir -p gpurun_out
( time timeout 900 python -m pytest tests/test_a_prodshape_gpu.py -q -x -k "gemm_res_ln" ) > gpurun_out/${TAG}_pytest_resln.log 2>&1; tail -12 gpurun_out/${TAG}_pytest_resln.log | cut -c1-400
( time timeout 900 python -m pytest tests/test_model_gpu.py tests/test_fullsize_gpu.py tests/test_ddp_gpu.py -q -x ) > gpurun_out/${TAG}_pytest_model.log 2>&1; tail -6 gpurun_out/${TAG}_pytest_model.log | cut -c1-400
timeout 1500 python tools/ab_env.py "" AB_WGRAD_LATE=0 AB_FUSE_LN=1 FS2_WGRAD_TG_WGS=256 FS2_WGRAD_TG_WGS=128 FS2_WGRAD_TG1_WGS=192 > gpurun_out/${TAG}_ab_env.log 2>&1; cat gpurun_out/${TAG}_ab_env.log
