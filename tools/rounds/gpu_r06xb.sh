#!/bin/bash
# r06xb: flake hunt on the wide one-tap weight-gradient kernel (counted vmcnt with 6 pieces per tile): the weight-gradient tests 15 times
export TMPDIR=/tmp
mkdir -p gpurun_out
n=0; f=0
for i in $(seq 1 15); do
  if timeout 300 python -m pytest tests/test_a_prodshape_gpu.py tests/test_ops_gpu.py -x -q -m gpu -k "wgrad or grads or weight" > /tmp/t.log 2>&1; then n=$((n+1)); else f=$((f+1)); tail -20 /tmp/t.log; fi
done
echo "weight-gradient tests: $n runs passed, $f failed" | tee gpurun_out/r06xb_flake.log
