#!/bin/bash
# r06v: flake hunt on the new attention kernels (asm-tracked loads, LDS-DMA staging): the attention tests 25 times, the graph tests 3 times
export TMPDIR=/tmp
mkdir -p gpurun_out
n=0; f=0
for i in $(seq 1 25); do
  if timeout 300 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "attention or attn" > /tmp/t.log 2>&1; then n=$((n+1)); else f=$((f+1)); tail -20 /tmp/t.log; fi
done
echo "attention tests: $n runs passed, $f failed" | tee gpurun_out/r06v_flake.log
for i in 1 2 3; do timeout 600 python -m pytest tests/test_graph_gpu.py tests/test_libritts_shape_gpu.py -x -q -m gpu 2>&1 | tail -1; done | tee -a gpurun_out/r06v_flake.log
