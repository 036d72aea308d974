#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python tools/ab_env.py "" FS2_P_TKS=1 FS2_P_ORDER=0 AB_LN_DEFER=0 AB_BRANCH=0 FS2_WGRAD_TG_WGS=160 FS2_WGRAD_TG_WGS=224 FS2_WGRAD_TG1_WGS=96 FS2_WGRAD_TG1_WGS=160 > gpurun_out/r03y_ab_env.log 2>&1; cat gpurun_out/r03y_ab_env.log
