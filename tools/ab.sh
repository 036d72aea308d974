#!/bin/bash
# A/B of two builds of the library on the SAME box: prev (fastspeech2_amd/libfs2hip_prev.so) vs current
for lib in prev cur; do
  if [ $lib = prev ]; then export FS2_LIB_PATH=$PWD/fastspeech2_amd/libfs2hip_prev.so; else unset FS2_LIB_PATH; fi
  echo "== $lib"
  python bench.py --no-cpu-baseline --no-roofline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['config']['final_loss'])"
  if [ -n "$1" ]; then python tools/bench_ops.py bf16 2>&1 | grep -E "$1"; fi
done
