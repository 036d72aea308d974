#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python tools/ab_synth.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r03s_ab_synth.log; cat gpurun_out/r03s_ab_synth.log
