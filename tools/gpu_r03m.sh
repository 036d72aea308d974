#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_model_gpu.py tests/test_fullsize_gpu.py tests/test_cli_gpu.py tests/test_bench_contract_gpu.py -q -x ) > gpurun_out/r04a_pytest.log 2>&1; tail -3 gpurun_out/r04a_pytest.log | cut -c1-300
timeout 600 python tools/ab_env.py "" AB_LENS_FWD=1,AB_LENS_BWD=1 AB_LENS_FWD=0,AB_LENS_BWD=0 > gpurun_out/r04a_ab_lj.log 2>&1; cat gpurun_out/r04a_ab_lj.log
AB_WORKLOAD=libritts timeout 600 python tools/ab_env.py "" AB_LENS_FWD=1,AB_LENS_BWD=1 AB_LENS_FWD=0,AB_LENS_BWD=0 > gpurun_out/r04a_ab_libri.log 2>&1; cat gpurun_out/r04a_ab_libri.log
