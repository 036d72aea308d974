#!/bin/bash
# same-box A/B of the whole train step: the round-1 final tree (_prev/, built by `git archive e52fff2 | tar -x -C _prev; make`)
# against the current tree, alternating, three rounds.
export TMPDIR=/tmp
for r in 1 2 3; do
  ( cd _prev && python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('round-1 tree ', d['ms_per_step'], d['value'])" )
  python bench.py --no-cpu-baseline --no-roofline --no-fp32 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('current tree ', d['ms_per_step'], d['value'], d['config']['window_ms_per_step'])"
done
