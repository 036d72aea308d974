"""Summarise a rocprofv3 rocpd sqlite database (kernel trace) into a per-kernel stats table (markdown/CSV-ish).
usage: python tools/rocpd_summary.py gpurun_out/prof/bench_results.db [n_steps] > profiles/xxx.md"""
import sqlite3
import sys
import re

db = sys.argv[1]
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
c = sqlite3.connect(db)
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else cols[0]
rows = c.execute(f"select {name_col}, count(*), sum(end-start), min(end-start), max(end-start) from kernels group by {name_col} order by 3 desc").fetchall()
total = sum(r[2] for r in rows)
print(f"# rocprofv3 --kernel-trace summary ({db})\n")
print(f"total kernel time {total/1e6:.3f} ms over {steps:g} profiled steps -> {total/1e6/steps:.3f} ms/step\n")
print("| kernel | calls | total ms | avg us | min us | max us | % |")
print("|---|---|---|---|---|---|---|")
for n, cnt, tot, mn, mx in rows[:40]:
    short = re.sub(r"\(.*", "", n)[:90]
    print(f"| {short} | {cnt} | {tot/1e6:.3f} | {tot/cnt/1e3:.1f} | {mn/1e3:.1f} | {mx/1e3:.1f} | {100*tot/total:.1f} |")
