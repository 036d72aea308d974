#!/bin/bash
# r06f: forked hipGraph capture - where does it deviate?  + the pinned-PCM-ring pipeline test + copies per batch
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python tools/dbg_fork_capture.py fp32 2>&1 | grep -v amdgpu.ids | cut -c1-700 | tee gpurun_out/r06f_dbg_fork_capture_fp32.log
timeout 600 python tools/dbg_fork_capture.py bf16 48 128 2>&1 | grep -v amdgpu.ids | cut -c1-700 | tee gpurun_out/r06f_dbg_fork_capture_bf16_full.log
( timeout 900 python -m pytest tests/test_vocoder_stft_gpu.py -x -q -m gpu -k "pipeline" ) 2>&1 | grep -v amdgpu.ids | tail -3 | tee gpurun_out/r06f_pytest_pipeline.log
rm -rf gpurun_out/prof; mkdir -p gpurun_out/prof
timeout 300 rocprofv3 --memory-copy-trace --kernel-trace -d gpurun_out/prof -o synth -- python bench.py --mode synth --steps 6 --warmup 6 --no-roofline --no-cpu-baseline > gpurun_out/prof_synth.log 2>&1
DB=$(find gpurun_out/prof -name '*.db' | head -1); python tools/rocpd_copies.py $DB 12 2>&1 | tee gpurun_out/r06f_copies_synth.log | head -40
rm -rf gpurun_out/prof
