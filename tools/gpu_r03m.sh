#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1200 python tools/ab_env.py "" FS2_WGRAD_TG1_MINUPS=16 FS2_WGRAD_TG1_MINUPS=32 FS2_WGRAD_TG1_MINUPS=4 > gpurun_out/r03t_ab_env.log 2>&1; cat gpurun_out/r03t_ab_env.log
