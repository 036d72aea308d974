#!/bin/bash
# round 5, call A: the multi-GPU changes on hardware (self-launch, counted one-rank RCCL, DDP vs oracle), a one-rank RCCL
# kernel trace, and this box's baseline step time.
TAG=${1:-r05a}
export TMPDIR=/tmp
mkdir -p gpurun_out
( time timeout 900 python -m pytest tests/test_nccl_gpu.py tests/test_ddp_gpu.py "tests/test_bench_contract_gpu.py::test_bench_two_ranks_shared_gpu" -x -q -s ) > gpurun_out/${TAG}_pytest_ddp.log 2>&1; tail -12 gpurun_out/${TAG}_pytest_ddp.log | cut -c1-600
echo "== self-launch: python bench.py --gpus 2 (shared GPU, gloo)"
FS2_BENCH_SHARE_GPU=1 timeout 600 python bench.py --gpus 2 --steps 3 --warmup 2 --windows 2 --no-roofline > gpurun_out/${TAG}_bench_selflaunch.log 2>&1; echo "rc=$?"; tail -1 gpurun_out/${TAG}_bench_selflaunch.log | cut -c1-1500
echo "== one-rank RCCL through bench.py"
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 1 --steps 10 --warmup 3 --windows 3 --no-roofline > gpurun_out/${TAG}_bench_rccl1.log 2>&1; echo "rc=$?"; tail -1 gpurun_out/${TAG}_bench_rccl1.log | cut -c1-1800
rm -rf gpurun_out/rccl_prof; mkdir -p gpurun_out/rccl_prof
timeout 600 rocprofv3 --kernel-trace -d gpurun_out/rccl_prof -o rccl -- python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29543 bench.py --gpus 1 --steps 4 --warmup 2 --windows 1 --no-roofline > gpurun_out/rccl_prof.log 2>&1
python tools/rccl_trace.py gpurun_out/rccl_prof > gpurun_out/${TAG}_rccl_one_rank_trace.md 2>&1; head -30 gpurun_out/${TAG}_rccl_one_rank_trace.md | cut -c1-700
rm -rf gpurun_out/rccl_prof
echo "== baseline bench"
timeout 600 python bench.py --no-cpu-baseline --no-fp32 --no-synth > gpurun_out/${TAG}_bench_bf16.log 2>&1; tail -1 gpurun_out/${TAG}_bench_bf16.log | cut -c1-2500
