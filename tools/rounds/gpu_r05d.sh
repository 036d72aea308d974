#!/bin/bash
TAG=${1:-r05d}
export TMPDIR=/tmp
mkdir -p gpurun_out
( time timeout 600 python -m pytest tests/test_vocoder_stft_gpu.py -x -q -k "resblock or resstage or hifigan" ) > gpurun_out/${TAG}_pytest_voc.log 2>&1; tail -3 gpurun_out/${TAG}_pytest_voc.log | cut -c1-300
( timeout 300 python -m pytest tests/test_a_prodshape_gpu.py -x -q -s -k "hifigan" ) > gpurun_out/${TAG}_pytest_voc2.log 2>&1; grep -E "whole wave|passed|failed|Error" gpurun_out/${TAG}_pytest_voc2.log | tail -4 | cut -c1-300
for i in 1 2; do
  timeout 300 python bench.py --mode synth --no-cpu-baseline --no-roofline > gpurun_out/${TAG}_synth_$i.log 2>&1; tail -1 gpurun_out/${TAG}_synth_$i.log | cut -c1-330
done
rm -rf gpurun_out/pmc_m
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_m -- python bench.py --mode synth --steps 3 --warmup 1 --no-roofline --no-cpu-baseline > gpurun_out/pmc_m.log 2>&1
python tools/pmc_mfma.py gpurun_out/pmc_m 4 > gpurun_out/${TAG}_pmc_mfma_synth.md 2>&1
rm -rf gpurun_out/pmc_m
head -14 gpurun_out/${TAG}_pmc_mfma_synth.md | cut -c1-250
rm -rf gpurun_out/prof; mkdir -p gpurun_out/prof
timeout 300 rocprofv3 --kernel-trace -d gpurun_out/prof -o bench -- python bench.py --mode synth --steps 4 --warmup 2 --no-roofline --no-cpu-baseline > gpurun_out/prof.log 2>&1
DB=$(find gpurun_out/prof -name '*.db' | head -1)
python tools/rocpd_summary.py $DB 6 shapes > gpurun_out/${TAG}_kernel_trace_synth.md 2>&1
rm -rf gpurun_out/prof
head -14 gpurun_out/${TAG}_kernel_trace_synth.md | cut -c1-180
