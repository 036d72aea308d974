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

# ---- per-shape breakdown: the same kernel grouped by launch geometry (grid), so each layer shape shows up separately
gcols = [c_ for c_ in cols if "grid" in c_.lower()]
if gcols and len(sys.argv) > 3 and sys.argv[3] == "shapes":
    gsel = ", ".join(gcols)
    rows = c.execute(f"select {name_col}, {gsel}, count(*), sum(end-start), min(end-start) from kernels group by {name_col}, {gsel} order by sum(end-start) desc").fetchall()
    print("\n## per launch geometry (top 60)\n")
    print("| kernel | grid | calls/step | avg us | min us | ms/step |")
    print("|---|---|---|---|---|---|")
    for r in rows[:60]:
        n, g, cnt, tot, mn = r[0], r[1:1 + len(gcols)], r[-3], r[-2], r[-1]
        short = re.sub(r"\(.*", "", n)[:70]
        print(f"| {short} | {'x'.join(str(v) for v in g)} | {cnt/steps:.1f} | {tot/cnt/1e3:.1f} | {mn/1e3:.1f} | {tot/1e6/steps:.3f} |")

# ---- device occupancy over time: union of all kernel intervals vs the span (idle = launch gaps / host-bound stretches)
if len(sys.argv) > 3:
    iv = c.execute("select start, end from kernels order by start").fetchall()
    if iv:
        span = max(e for _, e in iv) - iv[0][0]
        busy, cur_s, cur_e = 0, iv[0][0], iv[0][1]
        gaps = []
        for s0, e0 in iv[1:]:
            if s0 > cur_e:
                busy += cur_e - cur_s
                gaps.append(s0 - cur_e)
                cur_s, cur_e = s0, e0
            else:
                cur_e = max(cur_e, e0)
        busy += cur_e - cur_s
        big = sorted(gaps, reverse=True)
        print(f"\n## timeline\n\nspan {span/1e6:.3f} ms, some kernel running {busy/1e6:.3f} ms ({100*busy/span:.1f} %), idle {(span-busy)/1e6:.3f} ms in {len(gaps)} gaps "
              f"(gaps > 20 us: {sum(1 for g in gaps if g > 20000)}, their sum {sum(g for g in gaps if g > 20000)/1e6:.3f} ms; "
              f"gaps 2-20 us: {sum(1 for g in gaps if 2000 < g <= 20000)}, sum {sum(g for g in gaps if 2000 < g <= 20000)/1e6:.3f} ms)")

# ---- steady state only: between the end of the 3rd and of the last adam_kernel launch (whole train steps, no start-up)
if len(sys.argv) > 3:
    ad = c.execute(f"select end from kernels where {name_col} like '%adam_kernel%' order by end").fetchall()
    if len(ad) >= 5:
        w0, w1 = ad[2][0], ad[-1][0]
        nst = len(ad) - 3
        iv = c.execute("select start, end from kernels where end > ? and start < ? order by start", (w0, w1)).fetchall()
        busy, cur_s, cur_e, ngap, gsum = 0, max(iv[0][0], w0), iv[0][1], 0, 0
        for s0, e0 in iv[1:]:
            s0 = max(s0, w0); e0 = min(e0, w1)
            if s0 > cur_e:
                busy += cur_e - cur_s; ngap += 1; gsum += s0 - cur_e
                cur_s, cur_e = s0, e0
            else:
                cur_e = max(cur_e, e0)
        busy += cur_e - cur_s
        ksum = c.execute("select sum(end-start) from kernels where start >= ? and end <= ?", (w0, w1)).fetchone()[0]
        print(f"\n## steady state ({nst} steps)\n\n{(w1-w0)/1e6/nst:.3f} ms/step wall; device busy {busy/1e6/nst:.3f} ms/step, idle {gsum/1e6/nst:.3f} ms/step in {ngap/nst:.0f} gaps; "
              f"sum of kernel durations {ksum/1e6/nst:.3f} ms/step (overlap factor {ksum/busy:.2f})")

# ---- where the idle time sits (steady state): the largest gaps with the kernels on either side, and gap time summed by the
# kernel that FOLLOWS the gap (= what the host was late to launch)
if len(sys.argv) > 3:
    ad = c.execute(f"select end from kernels where {name_col} like '%adam_kernel%' order by end").fetchall()
    if len(ad) >= 5:
        w0, w1 = ad[2][0], ad[-1][0]
        nst = len(ad) - 3
        iv = c.execute(f"select start, end, {name_col} from kernels where end > ? and start < ? order by start", (w0, w1)).fetchall()
        gaps, cur_e, cur_n = [], iv[0][1], iv[0][2]
        for s0, e0, n in iv[1:]:
            if s0 > cur_e:
                gaps.append((s0 - cur_e, cur_n, n))
            if e0 > cur_e:
                cur_e, cur_n = e0, n
        short = lambda n: re.sub(r"\(.*", "", n).replace("void ", "")[:46]
        by_next = {}
        for g, p, n in gaps:
            d = by_next.setdefault(short(n), [0, 0]); d[0] += g; d[1] += 1
        print("\n## idle time by the kernel that follows the gap (steady state)\n\n| next kernel | gaps/step | idle us/step | avg gap us |\n|---|---|---|---|")
        for k, (g, cnt) in sorted(by_next.items(), key=lambda kv: -kv[1][0])[:18]:
            print(f"| {k} | {cnt/nst:.1f} | {g/1e3/nst:.1f} | {g/cnt/1e3:.1f} |")
        print("\n## largest gaps\n\n| gap us | after | before |\n|---|---|---|")
        for g, p, n in sorted(gaps, reverse=True)[:14]:
            print(f"| {g/1e3:.1f} | {short(p)} | {short(n)} |")
