#!/bin/bash
# Milestone evidence in one call: parity tests, full bench line (roofline + cpu baseline), rocprofv3 kernel trace
# (side stream on = what the bench runs; and off = every kernel alone on the device), PMC HBM-traffic passes.
TAG=${1:-r01x}
export TMPDIR=/tmp
mkdir -p gpurun_out
( timeout 900 python -m pytest tests -m gpu -x -q ) > gpurun_out/${TAG}_pytest.log 2>&1; tail -2 gpurun_out/${TAG}_pytest.log
timeout 600 python bench.py > gpurun_out/${TAG}_bench_bf16.log 2>&1; tail -1 gpurun_out/${TAG}_bench_bf16.log | cut -c1-300
for side in 1 0; do
  rm -rf gpurun_out/prof; mkdir -p gpurun_out/prof
  FS2_SIDE_STREAM=$side timeout 300 rocprofv3 --kernel-trace -d gpurun_out/prof -o bench -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-roofline > gpurun_out/prof.log 2>&1
  DB=$(find gpurun_out/prof -name '*.db' | head -1)
  python tools/rocpd_summary.py $DB 10 shapes > gpurun_out/${TAG}_kernel_trace_side${side}.md 2>&1
done
rm -rf gpurun_out/prof
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_$c
  FS2_SIDE_STREAM=0 timeout 600 rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmc_$c -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/pmc_$c.log 2>&1
done
python tools/pmc_traffic.py gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE 6 gpurun_out/${TAG}_pmc_traffic.json > gpurun_out/${TAG}_pmc_traffic.md 2>&1
rm -rf gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE
head -12 gpurun_out/${TAG}_pmc_traffic.md
