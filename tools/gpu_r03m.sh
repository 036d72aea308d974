#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
( time timeout 400 python bench.py ) > gpurun_out/r04f_bench_default.log 2>&1; tail -5 gpurun_out/r04f_bench_default.log | cut -c1-600
