#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_model_gpu.py tests/test_a_prodshape_gpu.py tests/test_checkpoint_gpu.py -x -q -m gpu -k "not contraction and not vocoder and not weight_gradient" 2>&1 | tail -5 ) > gpurun_out/r02r_pytest.log 2>&1
tail -3 gpurun_out/r02r_pytest.log | cut -c1-600
bash tools/ab_step.sh > gpurun_out/r02r_ab.log 2>&1; cat gpurun_out/r02r_ab.log
