#!/bin/bash
# same-box A/B of the batch-synthesis step: a reference tree under _prev/ (git archive <commit> | tar -x -C _prev; make -C _prev) against
# the working tree, alternating processes.  usage: bash tools/ab_synth.sh [rounds]
export TMPDIR=/tmp
N=${1:-3}
for i in $(seq 1 $N); do
  for t in prev cur; do
    if [ $t = prev ]; then d=_prev; else d=.; fi
    r=$(cd $d && timeout 300 python bench.py --mode synth --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])")
    echo "$t round $i: ms_per_step rtf = $r"
  done
done
