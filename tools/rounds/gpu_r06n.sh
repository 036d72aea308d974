#!/bin/bash
# r06n: PMC of the attention kernels alone (tools/bench_attn.py), pipelined dK/dV form
export TMPDIR=/tmp
cd /root/repo
mkdir -p gpurun_out
rm -rf gpurun_out/pmc_m gpurun_out/pmc_l
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_m -- python tools/bench_attn.py > gpurun_out/pmc_m.log 2>&1
python tools/pmc_mfma.py gpurun_out/pmc_m 1 > gpurun_out/r06n_attn_pmc_mfma.md 2>&1
timeout 600 rocprofv3 --pmc SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_l -- python tools/bench_attn.py > gpurun_out/pmc_l.log 2>&1
python tools/pmc_lds.py gpurun_out/pmc_l 1 > gpurun_out/r06n_attn_pmc_lds.md 2>&1
cat gpurun_out/r06n_attn_pmc_mfma.md gpurun_out/r06n_attn_pmc_lds.md
rm -rf gpurun_out/pmc_m gpurun_out/pmc_l
