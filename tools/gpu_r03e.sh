#!/bin/bash
TAG=${1:-r03e}
export TMPDIR=/tmp
mkdir -p gpurun_out
( time timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_a_prodshape_gpu.py -q -k "weight_gradient or conv_grads" ) > gpurun_out/${TAG}_pytest_wgrad.log 2>&1; tail -6 gpurun_out/${TAG}_pytest_wgrad.log | cut -c1-300
DEV=fastspeech2_amd/libfs2hip_dev.so
for tg1 in 0 1; do echo "== FS2_WGRAD_TG1=$tg1"; FS2_LIB_PATH=$DEV FS2_WGRAD_TG1=$tg1 timeout 300 python tools/bench_wgrad.py " k1" 2>&1 | grep -v amdgpu.ids; FS2_LIB_PATH=$DEV FS2_WGRAD_TG1=$tg1 timeout 300 python tools/bench_wgrad.py "qkv" 2>&1 | grep -v amdgpu.ids;  FS2_LIB_PATH=$DEV FS2_WGRAD_TG1=$tg1 timeout 300 python tools/bench_wgrad.py " fc" 2>&1 | grep -v amdgpu.ids; done > gpurun_out/${TAG}_bench_wgrad1.log 2>&1; cat gpurun_out/${TAG}_bench_wgrad1.log
FS2_BENCH_BACKEND=gloo FS2_BENCH_SHARE_GPU=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29521 tools/dbg_ddp.py --workload libritts > gpurun_out/${TAG}_dbg_ddp.log 2>&1; grep -E "iter|^    \(" gpurun_out/${TAG}_dbg_ddp.log | head -40
timeout 1500 python tools/ab_env.py "" FS2_WGRAD_TG1=0 FS2_WGRAD_TG1_WGS=128 FS2_WGRAD_TG1_WGS=192 FS2_WGRAD_TG1_WGS=384 FS2_WGRAD_TG1_WGS=512 > gpurun_out/${TAG}_ab_env.log 2>&1; cat gpurun_out/${TAG}_ab_env.log
