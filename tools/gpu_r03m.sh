#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests -m gpu -q ) > gpurun_out/r04b_pytest_full.log 2>&1; tail -5 gpurun_out/r04b_pytest_full.log | cut -c1-300
