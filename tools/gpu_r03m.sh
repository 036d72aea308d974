#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python tools/ab_env.py "" AB_WGRAD_LENS=0 > gpurun_out/r03x_ab_env.log 2>&1; cat gpurun_out/r03x_ab_env.log
