#!/bin/bash
# r05w: utils.SynthPipeline - parity tests, then a same-box A/B of the batch-synthesis loop (sequential vs two streams), alternating processes
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_vocoder_stft_gpu.py tests/test_cli_gpu.py -x -q -m gpu -k "pipeline or end_to_end or cli" 2>&1 | grep -v amdgpu.ids | tail -15 > gpurun_out/r05w_pytest.log
cat gpurun_out/r05w_pytest.log | tail -4
{
for i in 1 2 3; do
  for f in "--no-synth-pipeline" ""; do
    r=$(timeout 300 python bench.py --mode synth --no-cpu-baseline --no-roofline $f 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['config']['loop'][:30])")
    echo "round $i [${f:-pipeline}]: ms_per_step rtf = $r"
  done
done
} | tee gpurun_out/r05w_ab_synth_pipeline.log
