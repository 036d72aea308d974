#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_fullsize_gpu.py tests/test_a_prodshape_gpu.py tests/test_ddp_gpu.py tests/test_graph_gpu.py tests/test_cli_gpu.py tests/test_checkpoint_gpu.py tests/test_bench_contract_gpu.py -q -x -k "not contraction and not vocoder and not skinny and not polyphase" ) > gpurun_out/r03v_pytest_model.log 2>&1; tail -6 gpurun_out/r03v_pytest_model.log | cut -c1-400
timeout 600 python bench.py --no-cpu-baseline --no-fp32 --no-synth > gpurun_out/r03v_bench_bf16.log 2>&1; tail -1 gpurun_out/r03v_bench_bf16.log | cut -c1-1200
