#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
for v in nowait; do
  echo "== variant $v"
  FORK_VARIANT=$v timeout 600 python tools/dbg_fork_capture.py fp32 8 40 poison 2>&1 | grep -v amdgpu.ids | cut -c1-260 | head -6 | tee gpurun_out/r06h_poison_fp32_$v.log
  FORK_VARIANT=$v timeout 600 python tools/dbg_fork_capture.py fp32 2>&1 | grep -v amdgpu.ids | grep -E "FORKED|replay-to" | cut -c1-200 | tee gpurun_out/r06h_compare_fp32_$v.log
  FORK_VARIANT=$v timeout 600 python tools/dbg_fork_capture.py bf16 48 128 2>&1 | grep -v amdgpu.ids | grep -E "eager|FORKED|replay-to" | cut -c1-200 | tee gpurun_out/r06h_compare_bf16_full_$v.log
done
