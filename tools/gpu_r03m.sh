#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python tools/find_copies.py 2>&1 | grep -v amdgpu.ids | tail -45 > gpurun_out/r03s_find_copies.log; cat gpurun_out/r03s_find_copies.log
