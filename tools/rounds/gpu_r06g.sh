#!/bin/bash
# r06g: forked hipGraph capture - read-before-write finder (NaN poison + bisection over the capture-time allocations)
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python tools/dbg_fork_capture.py fp32 8 40 poison 2>&1 | grep -v amdgpu.ids | cut -c1-500 | tee gpurun_out/r06g_poison_fp32.log
timeout 600 python tools/dbg_fork_capture.py bf16 8 40 poison 2>&1 | grep -v amdgpu.ids | cut -c1-500 | tee gpurun_out/r06g_poison_bf16.log
