#!/bin/bash
# r05x: SynthPipeline with 1 / 2 / 3 vocoder streams - parity, then same-box A/B, alternating processes
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_vocoder_stft_gpu.py -x -q -m gpu -k "pipeline" 2>&1 | grep -v amdgpu.ids | tail -15 > gpurun_out/r05x_pytest.log
tail -3 gpurun_out/r05x_pytest.log
{
for i in 1 2 3; do
  for n in 1 2 3; do
    r=$(timeout 300 python bench.py --mode synth --no-cpu-baseline --no-roofline --synth-voc-streams $n 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])")
    echo "round $i [voc streams $n]: ms_per_step rtf = $r"
  done
done
} | tee gpurun_out/r05x_ab_synth_voc_streams.log
