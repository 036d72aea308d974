#!/bin/bash
# r06e: what are the ~99 small copies per synthesis batch (and the 32 per train step)?  memory-copy trace of both
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -rf gpurun_out/prof; mkdir -p gpurun_out/prof
timeout 300 rocprofv3 --memory-copy-trace --kernel-trace -d gpurun_out/prof -o synth -- python bench.py --mode synth --steps 6 --warmup 6 --no-roofline --no-cpu-baseline > gpurun_out/prof_synth.log 2>&1
DB=$(find gpurun_out/prof -name '*.db' | head -1); python tools/rocpd_copies.py $DB 12 2>&1 | tee gpurun_out/r06e_copies_synth.log | head -60
rm -rf gpurun_out/prof; mkdir -p gpurun_out/prof
timeout 300 rocprofv3 --memory-copy-trace --hip-runtime-trace --kernel-trace -d gpurun_out/prof -o synth -- python bench.py --mode synth --steps 3 --warmup 3 --no-roofline --no-cpu-baseline --no-synth-pipeline > gpurun_out/prof_synth2.log 2>&1
DB=$(find gpurun_out/prof -name '*.db' | head -1)
python - $DB <<'PY' 2>&1 | tee gpurun_out/r06e_hip_calls_synth.log | head -60
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
print([t for t in tabs if 'region' in t.lower() or 'api' in t.lower() or 'hip' in t.lower()][:20])
for t in tabs:
    if t.lower() in ('regions', 'hip_api', 'top') or 'regions' == t.lower():
        cols = [r[1] for r in c.execute(f"pragma table_info({t})")]
        print(t, cols)
try:
    rows = c.execute("select name, count(*), sum(end-start) from regions group by name order by 2 desc limit 30").fetchall()
    for r in rows: print(f"  {r[0]!s:50s} x {r[1]:6d}  {r[2]/1e6:9.3f} ms")
except Exception as e:
    print('regions query failed:', e)
PY
rm -rf gpurun_out/prof
