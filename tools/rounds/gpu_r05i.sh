#!/bin/bash
TAG=${1:-r05i}
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -rf gpurun_out/pmc_m
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_m -- python bench.py --mode synth --steps 3 --warmup 1 --no-roofline --no-cpu-baseline > gpurun_out/pmc_m.log 2>&1
python tools/pmc_mfma.py gpurun_out/pmc_m 4 > gpurun_out/${TAG}_pmc_mfma_synth.md 2>&1
rm -rf gpurun_out/pmc_m
head -9 gpurun_out/${TAG}_pmc_mfma_synth.md | cut -c1-250
rm -rf gpurun_out/pmc_l
timeout 600 rocprofv3 --pmc SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_l -- python bench.py --mode synth --steps 3 --warmup 1 --no-roofline --no-cpu-baseline > gpurun_out/pmc_l.log 2>&1
python tools/pmc_lds.py gpurun_out/pmc_l 4 > gpurun_out/${TAG}_pmc_lds_synth.md 2>&1
rm -rf gpurun_out/pmc_l
head -9 gpurun_out/${TAG}_pmc_lds_synth.md | cut -c1-250
