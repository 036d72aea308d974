#!/bin/bash
TAG=${1:-r05n}
export TMPDIR=/tmp
mkdir -p gpurun_out
( time timeout 900 python -m pytest tests/test_vocoder_stft_gpu.py -x -q -s ) > gpurun_out/${TAG}_pytest_voc.log 2>&1; grep -E "passed|failed|Error|error|assert" gpurun_out/${TAG}_pytest_voc.log | tail -12 | cut -c1-400
( timeout 600 python -m pytest tests/test_a_prodshape_gpu.py tests/test_cli_gpu.py -x -q -s -k "hifigan or synth or vocoder" ) > gpurun_out/${TAG}_pytest_voc2.log 2>&1; grep -E "stage|whole wave|passed|failed|Error" gpurun_out/${TAG}_pytest_voc2.log | tail -10 | cut -c1-300
for i in 1 2; do
  timeout 300 python bench.py --mode synth --no-cpu-baseline > gpurun_out/${TAG}_synth_$i.log 2>&1; tail -1 gpurun_out/${TAG}_synth_$i.log | cut -c1-330
done
rm -rf gpurun_out/prof; mkdir -p gpurun_out/prof
timeout 300 rocprofv3 --kernel-trace -d gpurun_out/prof -o bench -- python bench.py --mode synth --steps 4 --warmup 2 --no-roofline --no-cpu-baseline > gpurun_out/prof.log 2>&1
DB=$(find gpurun_out/prof -name '*.db' | head -1)
python tools/rocpd_summary.py $DB 6 shapes > gpurun_out/${TAG}_kernel_trace_synth.md 2>&1
rm -rf gpurun_out/prof
head -16 gpurun_out/${TAG}_kernel_trace_synth.md | cut -c1-180
