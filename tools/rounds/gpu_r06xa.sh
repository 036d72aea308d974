#!/bin/bash
# r06xa: shipped library with the wide one-tap weight-gradient rule: parity, step A/B against the previous commit's library
export TMPDIR=/tmp
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests/test_a_prodshape_gpu.py tests/test_ops_gpu.py tests/test_model_gpu.py tests/test_fullsize_gpu.py tests/test_z_bf16_budget_gpu.py -x -q -m gpu ) 2>&1 | grep -v amdgpu.ids | tail -3 | cut -c1-500 | tee gpurun_out/r06xa_pytest.log
bash tools/ab_lib.sh 3 | tee gpurun_out/r06xa_ab_step.log
