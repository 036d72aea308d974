"""Dev tool: what the memory copies of a rocprofv3 run were (run with --memory-copy-trace --kernel-trace): count / bytes per
(direction, size) and per step.  usage: python tools/rocpd_copies.py <results.db> [n_steps]"""
import sqlite3
import sys

db = sys.argv[1]
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
cand = [t for t in tabs if t.lower() == "memory_copies"] or [t for t in tabs if "memory_cop" in t.lower() or "memcpy" in t.lower()]
print("tables/views with copies:", cand)
for t in cand:
    cols = [r[1] for r in c.execute(f"pragma table_info({t})")]
    print(t, cols)
    n = c.execute(f"select count(*) from {t}").fetchone()[0]
    print("rows:", n)
    if not n:
        continue
    size_col = next((x for x in cols if x.lower() in ("size", "bytes")), None)
    name_col = next((x for x in cols if x.lower() in ("name", "kind", "direction")), None)
    if size_col and not name_col:
        rows = c.execute(f"select {size_col}, count(*) from {t} group by {size_col} order by 2 desc limit 40").fetchall()
        for r in rows:
            print(f"  {r[0]:>12} B  x {r[1] / steps:.1f} per step")
    if size_col and name_col:
        rows = c.execute(f"select {name_col}, {size_col}, count(*) from {t} group by {name_col}, {size_col} order by 3 desc limit 40").fetchall()
        for r in rows:
            print(f"  {r[0]!s:40s} {r[1]:>12} B  x {r[2] / steps:.1f} per step")
    break
