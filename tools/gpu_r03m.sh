#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_a_prodshape_gpu.py tests/test_model_gpu.py -q -x -k "batch_norm or batchnorm or bn_ or postnet or model or parity or step" ) > gpurun_out/r04e_pytest_bn.log 2>&1; tail -3 gpurun_out/r04e_pytest_bn.log | cut -c1-300
timeout 300 python tools/ab_env.py "" > gpurun_out/r04e_ab.log 2>&1; cat gpurun_out/r04e_ab.log
