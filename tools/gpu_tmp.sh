#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_a_prodshape_gpu.py -x -q -m gpu -s 2>&1 | grep -E "rounding only\)|passed|failed|Error|error" | cut -c1-600 | tail -15 ) > gpurun_out/r02l_pytest.log 2>&1
cat gpurun_out/r02l_pytest.log
