#!/bin/bash
# r06k: ablations of the eight-consumer-wave persistent kernel (dev library): what bounds it once two waves per SIMD interleave?
export TMPDIR=/tmp
mkdir -p gpurun_out
python tools/bench_p.py ab FS2_P_CW=8 FS2_P_CW=8,FS2_GEMM_ABL=1 FS2_P_CW=8,FS2_GEMM_ABL=2 FS2_P_CW=8,FS2_GEMM_ABL=3 FS2_P_CW=8,FS2_GEMM_ABL=4 FS2_P_CW=4,FS2_GEMM_ABL=2 FS2_P_CW=4,FS2_GEMM_ABL=3 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06k_bench_p_cw8_abl.log
