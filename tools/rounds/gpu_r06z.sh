#!/bin/bash
# r06z: the last tree - full GPU suite, the driver's own invocation of bench.py (no flags), smoke()
export TMPDIR=/tmp
mkdir -p gpurun_out
( time timeout 1800 python -m pytest tests -m gpu -x -q ) > gpurun_out/r06z_pytest_final.log 2>&1; tail -4 gpurun_out/r06z_pytest_final.log
( time python bench.py ) > gpurun_out/r06z_bench_default.log 2>&1; grep '^{' gpurun_out/r06z_bench_default.log | tail -1 | cut -c1-400; tail -4 gpurun_out/r06z_bench_default.log | grep real
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee gpurun_out/r06z_smoke.log
