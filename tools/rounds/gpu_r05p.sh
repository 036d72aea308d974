#!/bin/bash
TAG=${1:-r05p}
export TMPDIR=/tmp
mkdir -p gpurun_out
( time timeout 900 python -m pytest tests/test_a_prodshape_gpu.py -x -q -k "contraction or hifigan or tail or splitk or gemm" ) > gpurun_out/${TAG}_pytest_contraction.log 2>&1; tail -4 gpurun_out/${TAG}_pytest_contraction.log | cut -c1-300
( timeout 600 python -m pytest tests/test_vocoder_stft_gpu.py tests/test_ops_gpu.py -x -q -k "conv or hifigan or resblock or gemm" ) > gpurun_out/${TAG}_pytest_voc.log 2>&1; tail -3 gpurun_out/${TAG}_pytest_voc.log | cut -c1-300
bash tools/ab_synth.sh 3 2>&1 | tee gpurun_out/${TAG}_ab_synth.log
bash tools/ab_step.sh 3 2>&1 | tee gpurun_out/${TAG}_ab_step.log | tail -12
