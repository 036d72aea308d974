"""Dev tool: for every dispatch of kernels matching <pattern> in a rocprofv3 kernel-trace database, the kernels dispatched right
before and after it ON THE SAME QUEUE, aggregated - tells where anonymous runtime kernels (__amd_rocclr_copyBuffer, fillBuffer,
torch elementwise launches) come from.  usage: python tools/rocpd_neighbours.py <results.db> <pattern> [n_steps]"""
import collections
import re
import sqlite3
import sys

db, pat = sys.argv[1], sys.argv[2]
steps = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
c = sqlite3.connect(db)
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
qcol = next((x for x in ("queue_id", "stream_id", "queue") if x in cols), None)
gcols = [x for x in cols if "grid" in x.lower()]
rows = c.execute(f"select name, start, end, {qcol or '0'}, {gcols[0] if gcols else '0'} from kernels order by start").fetchall()
short = lambda n: re.sub(r"\(.*", "", n)[:60]
byq = collections.defaultdict(list)
for r in rows:
    byq[r[3]].append(r)
ctx = collections.Counter()
perq = collections.Counter()
for q, lst in byq.items():
    for i, r in enumerate(lst):
        if pat in r[0]:
            prev = short(lst[i - 1][0]) if i > 0 else "-"
            nxt = short(lst[i + 1][0]) if i + 1 < len(lst) else "-"
            ctx[(prev, nxt, r[4])] += 1
            perq[q] += 1
print(f"{sum(perq.values()) / steps:.1f} dispatches of *{pat}* per step; per queue: {dict(perq)}")
for (prev, nxt, g), n in ctx.most_common(30):
    print(f"  {n / steps:6.1f}/step  grid {g:>8}  after [{prev}]  before [{nxt}]")
