#!/bin/bash
# r06t: where does the step stand against its own critical chain now (attention faster)?  side-stream workgroup target re-swept; no-wgrad floor
export TMPDIR=/tmp
mkdir -p gpurun_out
python tools/ab_env.py "" AB_SKIP_WGRAD=1 FS2_WGRAD_TG_WGS=128 FS2_WGRAD_TG_WGS=160 FS2_WGRAD_TG_WGS=224 FS2_WGRAD_TG_WGS=256 AB_SIDE=0 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06t_ab_env.log
