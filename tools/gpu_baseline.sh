#!/bin/bash
# one gpurun call: parity tests, bench (graph + eager), per-shape op bench, kernel trace
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests -m gpu -x -q ) > gpurun_out/pytest.log 2>&1
tail -3 gpurun_out/pytest.log
timeout 300 python bench.py > gpurun_out/bench_graph.log 2>&1; tail -1 gpurun_out/bench_graph.log | cut -c1-400
timeout 300 python bench.py --graph 0 --no-cpu-baseline --no-roofline > gpurun_out/bench_eager.log 2>&1; tail -1 gpurun_out/bench_eager.log | cut -c1-300
timeout 300 python tools/bench_ops.py bf16 > gpurun_out/bench_ops.log 2>&1; cat gpurun_out/bench_ops.log
rm -rf gpurun_out/prof; mkdir -p gpurun_out/prof
timeout 300 rocprofv3 --kernel-trace -d gpurun_out/prof -o bench -- python bench.py --steps 6 --warmup 2 --graph 0 --no-cpu-baseline --no-roofline > gpurun_out/prof.log 2>&1
DB=$(find gpurun_out/prof -name '*.db' | head -1)
python tools/rocpd_summary.py $DB 11 shapes > gpurun_out/prof_summary.md 2>&1
find gpurun_out/prof -name '*.db' -size +30M -delete
head -50 gpurun_out/prof_summary.md
