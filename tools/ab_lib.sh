#!/bin/bash
# same-box A/B of two BUILDS of the library on the current tree's Python: fastspeech2_amd/libfs2hip_prev.so (built from an earlier
# commit: `git archive <commit> | tar -x -C /tmp/prev; make -C /tmp/prev; cp .../libfs2hip.so fastspeech2_amd/libfs2hip_prev.so`;
# *.so files travel to the GPU box) against the shipped libfs2hip.so - alternating processes, $1 rounds (default 3), extra bench flags in $2
export TMPDIR=/tmp
R=${1:-3}
for r in $(seq 1 $R); do
  FS2_LIB_PATH=fastspeech2_amd/libfs2hip_prev.so python bench.py --no-cpu-baseline --no-roofline --no-fp32 --no-synth --no-graph-line $2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('previous build ', d['ms_per_step'], d['value'], d['config']['window_ms_per_step'], d['config']['library'])"
  python bench.py --no-cpu-baseline --no-roofline --no-fp32 --no-synth --no-graph-line $2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('current build  ', d['ms_per_step'], d['value'], d['config']['window_ms_per_step'], d['config']['library'])"
done
