#!/bin/bash
# r06a: the round-5 evidence gaps + config 4's sampler window (next 3 i), one call
#  (1) configs[3]-shape oracle parity through the lens + tile-map mode, bench contract (fp32 synthesis RTF, hw_queues, exchange_exposed_ms)
#  (2) the one-rank RCCL step with 16 hardware queues (the setting bench.py / train.py now make at EVERY world size)
#  (3) LibriTTS sampler window sweep: epoch valid fraction + measured frames/s at group_size 4 / 16 / 64
export TMPDIR=/tmp
mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests/test_libritts_shape_gpu.py tests/test_bench_contract_gpu.py -x -q -m gpu -s ) 2>&1 | grep -v amdgpu.ids > gpurun_out/r06a_pytest.log
grep -E "passed|failed|FAILED|Error|LibriTTS bucket|real" gpurun_out/r06a_pytest.log | tail -12 | cut -c1-1200
( GPU_MAX_HW_QUEUES=16 timeout 600 python -m pytest tests/test_nccl_gpu.py -x -q -m gpu ) 2>&1 | grep -v amdgpu.ids | tail -4 | cut -c1-400 > gpurun_out/r06a_pytest_nccl_hwq16.log; cat gpurun_out/r06a_pytest_nccl_hwq16.log
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-fp32 --no-synth --no-graph-line --no-roofline 2>/dev/null | tail -1 > gpurun_out/r06a_bench_rccl1_hwq16.log; cut -c1-1500 gpurun_out/r06a_bench_rccl1_hwq16.log
timeout 600 python tools/bench_libritts_sweep.py --groups 4,16,64 --nsteps 8 2>&1 | grep -v amdgpu.ids > gpurun_out/r06a_libritts_sweep.log; cut -c1-700 gpurun_out/r06a_libritts_sweep.log
for G in 4 16; do
  timeout 300 python bench.py --workload libritts --group-size $G --no-cpu-baseline --no-fp32 --no-synth --no-graph-line 2>/dev/null | tail -1 > gpurun_out/r06a_bench_libritts_g$G.log; cut -c1-900 gpurun_out/r06a_bench_libritts_g$G.log
done
