#!/bin/bash
# Milestone evidence in one call: parity tests, full bench line (roofline + cpu baseline + fp32), synthesis bench, rocprofv3
# kernel traces (side stream on = what the bench runs; off = every kernel alone on the device), PMC HBM-traffic passes and
# two PMC passes for the MFMA / LDS picture of every kernel.
TAG=${1:-r05m}
export TMPDIR=/tmp
mkdir -p gpurun_out
# (1) HBM traffic of the contraction kernels FIRST: bench.py refuses a PMC file that was measured on other kernel sources
# (fastspeech2_amd/_lib.kernel_source_sha) and tests/test_bench_contract_gpu.py wants roofline.traffic non-null - the file of THIS
# tree has to exist under profiles/ before the suite runs (it is merged back through gpurun_out/ and committed)
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_$c
  timeout 600 rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmc_$c -- python bench.py --steps 3 --warmup 1 --windows 1 --side-stream 0 --no-cpu-baseline --no-roofline --no-fp32 --no-synth --no-graph-line > gpurun_out/pmc_$c.log 2>&1
done
python tools/pmc_traffic.py gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE 6 gpurun_out/${TAG}_pmc_traffic.json > gpurun_out/${TAG}_pmc_traffic.md 2>&1
rm -rf gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE
cp gpurun_out/${TAG}_pmc_traffic.json gpurun_out/${TAG}_pmc_traffic.md profiles/
( time timeout 1500 python -m pytest tests -m gpu -x -q -s ) > gpurun_out/${TAG}_pytest.log 2>&1; grep -E "rel-Frobenius|valid-frame|ratios|passed|failed|FAILED|eager-vs|SKIPPED" gpurun_out/${TAG}_pytest.log | tail -16 | cut -c1-900
timeout 600 python bench.py > gpurun_out/${TAG}_bench_bf16.log 2>&1; tail -1 gpurun_out/${TAG}_bench_bf16.log | cut -c1-2500
timeout 300 python bench.py --mode synth > gpurun_out/${TAG}_bench_synth.log 2>&1; tail -1 gpurun_out/${TAG}_bench_synth.log | cut -c1-1500
timeout 300 python bench.py --workload libritts --no-cpu-baseline --no-fp32 --no-synth > gpurun_out/${TAG}_bench_libritts.log 2>&1; tail -1 gpurun_out/${TAG}_bench_libritts.log | cut -c1-700
for side in 1 0; do
  rm -rf gpurun_out/prof; mkdir -p gpurun_out/prof
  timeout 300 rocprofv3 --kernel-trace -d gpurun_out/prof -o bench -- python bench.py --steps 6 --warmup 2 --windows 1 --side-stream $side --no-cpu-baseline --no-roofline --no-fp32 --no-synth --no-graph-line > gpurun_out/prof.log 2>&1
  DB=$(find gpurun_out/prof -name '*.db' | head -1)
  python tools/rocpd_summary.py $DB 10 shapes > gpurun_out/${TAG}_kernel_trace_side${side}.md 2>&1
done
rm -rf gpurun_out/prof; mkdir -p gpurun_out/prof
timeout 300 rocprofv3 --kernel-trace -d gpurun_out/prof -o bench -- python bench.py --mode synth --steps 4 --warmup 2 --no-roofline > gpurun_out/prof.log 2>&1
DB=$(find gpurun_out/prof -name '*.db' | head -1)
python tools/rocpd_summary.py $DB 6 shapes > gpurun_out/${TAG}_kernel_trace_synth.md 2>&1
rm -rf gpurun_out/prof
rm -rf gpurun_out/pmc_m
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_m -- python bench.py --steps 3 --warmup 1 --windows 1 --side-stream 0 --no-cpu-baseline --no-roofline --no-fp32 --no-synth --no-graph-line > gpurun_out/pmc_m.log 2>&1
python tools/pmc_mfma.py gpurun_out/pmc_m 6 > gpurun_out/${TAG}_pmc_mfma.md 2>&1
rm -rf gpurun_out/pmc_l
timeout 600 rocprofv3 --pmc SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_l -- python bench.py --steps 3 --warmup 1 --windows 1 --side-stream 0 --no-cpu-baseline --no-roofline --no-fp32 --no-synth --no-graph-line > gpurun_out/pmc_l.log 2>&1
python tools/pmc_lds.py gpurun_out/pmc_l 6 > gpurun_out/${TAG}_pmc_lds.md 2>&1
rm -rf gpurun_out/pmc_m gpurun_out/pmc_l
head -14 gpurun_out/${TAG}_pmc_traffic.md; head -12 gpurun_out/${TAG}_pmc_mfma.md; head -12 gpurun_out/${TAG}_pmc_lds.md
