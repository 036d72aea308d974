#!/bin/bash
# r06l: shipped library with the per-shape consumer-wave rule: contraction parity, then the step A/B against the previous commit's build
export TMPDIR=/tmp
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_a_prodshape_gpu.py tests/test_ops_gpu.py tests/test_model_gpu.py -x -q -m gpu ) 2>&1 | grep -v amdgpu.ids | tail -3 | cut -c1-600 | tee gpurun_out/r06l_pytest.log
bash tools/ab_lib.sh 3 | tee gpurun_out/r06l_ab_step.log
