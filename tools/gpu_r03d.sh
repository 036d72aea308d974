#!/bin/bash
TAG=${1:-r03d}
export TMPDIR=/tmp
mkdir -p gpurun_out
FS2_BENCH_BACKEND=gloo FS2_BENCH_SHARE_GPU=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29519 bench.py --gpus 2 --steps 2 --warmup 1 --workload libritts --no-roofline > gpurun_out/${TAG}_libri2.log 2>&1; tail -3 gpurun_out/${TAG}_libri2.log | cut -c1-1500
timeout 1500 python tools/ab_env.py "" FS2_WGRAD_TG_WGS=64 FS2_WGRAD_TG_WGS=96 FS2_WGRAD_TG_WGS=128 FS2_WGRAD_TG_WGS=160 FS2_WGRAD_TG_WGS=192 FS2_WGRAD_TG_WGS=384 FS2_WGRAD_TG_WGS=512 FS2_WGRAD_TG_WGS=1024 FS2_WGRAD_SLAB=0 AB_SKIP_WGRAD=1 AB_SIDE=0 > gpurun_out/${TAG}_ab_env.log 2>&1; cat gpurun_out/${TAG}_ab_env.log
