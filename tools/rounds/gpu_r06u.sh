#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
python tools/ab_env.py "" AB_FUSE_LN=1 AB_FUSE_LN=stream AB_LN_DEFER=0 AB_BRANCH=0 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06u_ab_env.log
