#!/bin/bash
# same-box A/B of the whole train step: a reference tree (_prev/, built by `git archive <commit> | tar -x -C _prev; make -C _prev`)
# against the current tree, alternating, three rounds.
export TMPDIR=/tmp
for r in 1 2 3; do
  ( cd _prev && python bench.py --no-cpu-baseline --no-roofline --no-fp32 --no-synth --no-graph-line 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('reference tree ', d['ms_per_step'], d['value'], d['config']['window_ms_per_step'])" )
  python bench.py --no-cpu-baseline --no-roofline --no-fp32 --no-synth --no-graph-line 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('current tree   ', d['ms_per_step'], d['value'], d['config']['window_ms_per_step'])"
done
