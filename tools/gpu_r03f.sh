#!/bin/bash
TAG=${1:-r03f}
export TMPDIR=/tmp
mkdir -p gpurun_out
DEV=fastspeech2_amd/libfs2hip_dev.so
( time timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_a_prodshape_gpu.py -q -k "weight_gradient or conv_grads" ) > gpurun_out/${TAG}_pytest_wgrad.log 2>&1; tail -4 gpurun_out/${TAG}_pytest_wgrad.log | cut -c1-300
for tg1 in 1 0; do
echo "== DDP NaN trace, FS2_WGRAD_TG1=$tg1"
FS2_LIB_PATH=$DEV FS2_WGRAD_TG1=$tg1 DBG_WGRAD=1 FS2_BENCH_BACKEND=gloo FS2_BENCH_SHARE_GPU=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 2952$tg1 tools/dbg_ddp.py --workload libritts 2>&1 | grep -E "iter|^    \(|non-finite" | head -30
done > gpurun_out/${TAG}_dbg_ddp.log 2>&1; cat gpurun_out/${TAG}_dbg_ddp.log | cut -c1-500
timeout 1500 python tools/ab_env.py "" FS2_WGRAD_TG1=0 FS2_WGRAD_TG1_WGS=64 FS2_WGRAD_TG1_WGS=96 FS2_WGRAD_TG1_WGS=128 FS2_WGRAD_TG1_WGS=160 > gpurun_out/${TAG}_ab_env.log 2>&1; cat gpurun_out/${TAG}_ab_env.log
