#!/bin/bash
# r06q: attention state of the round (pipelined dK/dV, lane-constant addresses, XCD-aware block map, 32-bit prefetch offsets): parity + step A/B
export TMPDIR=/tmp
mkdir -p gpurun_out
for l in fastspeech2_amd/libfs2hip_prev.so fastspeech2_amd/libfs2hip.so; do echo $l; FS2_LIB_PATH=$l python tools/bench_attn.py 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/r06q_bench_attn.log
( timeout 1500 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py tests/test_fullsize_gpu.py tests/test_libritts_shape_gpu.py tests/test_z_bf16_budget_gpu.py tests/test_graph_gpu.py -x -q -m gpu ) 2>&1 | grep -v amdgpu.ids | tail -3 | cut -c1-600 | tee gpurun_out/r06q_pytest.log
bash tools/ab_lib.sh 3 | tee gpurun_out/r06q_ab_step.log
