#!/bin/bash
# r06xc: forward attention with small items (64 queries, 32-key tiles, 4 workgroups per CU) against the 128-query kernel - dev switch
export TMPDIR=/tmp
mkdir -p gpurun_out
( FS2_LIB_PATH=fastspeech2_amd/libfs2hip_dev.so FS2_ATTN_FWD_SMALL=1 timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "attention" ) 2>&1 | grep -v amdgpu.ids | tail -2 | cut -c1-300 | tee gpurun_out/r06xc_pytest.log
for v in 0 1 0 1; do echo "FS2_ATTN_FWD_SMALL=$v"; FS2_LIB_PATH=fastspeech2_amd/libfs2hip_dev.so FS2_ATTN_FWD_SMALL=$v python tools/bench_attn.py 2>&1 | grep fwd; done | tee gpurun_out/r06xc_bench_attn.log
