#!/bin/bash
# r06b: (1) the contract / RCCL tests after the shared-GPU hardware-queue fix; the 2-rank shared-GPU line with 16 queues forced, for the record
#       (2) LibriTTS sampler-window sweep over the WHOLE epoch (same utterances for every window)  (3) host-time profile of a short step
export TMPDIR=/tmp
mkdir -p gpurun_out
( time timeout 1200 python -m pytest tests/test_bench_contract_gpu.py tests/test_nccl_gpu.py -x -q -m gpu ) 2>&1 | grep -v amdgpu.ids | cut -c1-1500 > gpurun_out/r06b_pytest.log
grep -E "passed|failed|FAILED|Error|real" gpurun_out/r06b_pytest.log | tail -6
FS2_BENCH_BACKEND=gloo FS2_BENCH_SHARE_GPU=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 2 --steps 2 --warmup 1 --hw-queues 16 2>/dev/null | tail -1 > gpurun_out/r06b_shared2_hwq16.log
python -c "
import json; d=json.load(open('gpurun_out/r06b_shared2_hwq16.log')); r=d['roofline']; print('shared-GPU 2 ranks, 16 queues each:', d['ms_per_step'], r['kernel'], {k:(v['ms_per_step'],v['launches_per_step']) for k,v in r['conv_gemm_family'].items()})"
timeout 900 python tools/bench_libritts_sweep.py --groups 4,16,64 2>&1 | grep -v amdgpu.ids > gpurun_out/r06b_libritts_sweep.log; cut -c1-600 gpurun_out/r06b_libritts_sweep.log
timeout 300 python tools/host_profile.py 25 2>&1 | grep -v amdgpu.ids > gpurun_out/r06b_host_profile_L25.log; head -60 gpurun_out/r06b_host_profile_L25.log
timeout 300 python tools/host_profile.py 128 2>&1 | grep -v amdgpu.ids | head -3 > gpurun_out/r06b_host_profile_L128.log; head -3 gpurun_out/r06b_host_profile_L128.log
