#!/bin/bash
# r06c: XCD-aware placement of the weight-gradient (split, tile) pairs - parity, per-shape times (both builds), step A/B, sweep redo
export TMPDIR=/tmp
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_a_prodshape_gpu.py tests/test_ops_gpu.py -x -q -m gpu -k "wgrad or weight_grad or train_step" ) 2>&1 | grep -v amdgpu.ids | tail -4 | cut -c1-600 > gpurun_out/r06c_pytest.log; cat gpurun_out/r06c_pytest.log
{ echo "== previous build"; FS2_LIB_PATH=fastspeech2_amd/libfs2hip_prev.so timeout 300 python tools/bench_wgrad.py 2>&1 | grep -v amdgpu.ids; echo "== current build (XCD-aware pairs)"; timeout 300 python tools/bench_wgrad.py 2>&1 | grep -v amdgpu.ids; } > gpurun_out/r06c_bench_wgrad.log; cat gpurun_out/r06c_bench_wgrad.log | cut -c1-200
bash tools/ab_lib.sh 3 | tee gpurun_out/r06c_ab_step.log
timeout 900 python tools/bench_libritts_sweep.py --groups 4,16,64 2>&1 | grep -v amdgpu.ids > gpurun_out/r06c_libritts_sweep.log; cut -c1-520 gpurun_out/r06c_libritts_sweep.log
