"""Where the gradient exchange sits in a data-parallel step: post-processes the rocprofv3 rocpd databases of a 2-rank run
(kernel trace + memory-copy trace; on the 1-GPU test box the two ranks share the device and talk over gloo, so a "collective" is
what gloo does with a device tensor: the bucket's div_ kernel on the exchange stream, a D2H copy, the host reduction, an H2D copy).
For rank 0's last profiled steps it prints, relative to the first backward kernel of the step: when backward's last kernel ends,
when each bucket starts (div_ kernel) and when its data is back (H2D copy end), and when the clip pass (sumsq_partial_kernel)
starts - i.e. which buckets travel UNDER backward and how long the step waits between backward's end and clip + Adam.

    rocprofv3 --kernel-trace --memory-copy-trace -d gpurun_out/ddp_prof -- python -m torch.distributed.run ... bench.py --gpus 2 ...
    python tools/ddp_trace.py gpurun_out/ddp_prof > profiles/r04_ddp_trace.md"""
import glob
import sqlite3
import sys


def cols(c, table):
    return [r[1] for r in c.execute(f"pragma table_info({table})")]


def tables(c):
    return [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]


def main():
    dbs = sorted(glob.glob(sys.argv[1] + "/**/*.db", recursive=True))
    print(f"# data-parallel step timeline from rocprofv3 ({len(dbs)} databases under {sys.argv[1]})\n")
    for db in dbs:
        c = sqlite3.connect(db)
        tb = tables(c)
        if "kernels" not in tb:
            continue
        kc = cols(c, "kernels")
        name = "name" if "name" in kc else kc[0]
        extra = [x for x in ("stream_id", "queue_id", "pid", "tid", "stream") if x in kc]
        n_adam = c.execute(f"select count(*) from kernels where {name} like '%adam_kernel%'").fetchone()[0]
        if n_adam < 3:
            continue                                            # the launcher process / a rank that did not train
        print(f"## {db.split('/')[-1]}: {n_adam} optimiser steps; kernel columns {kc}\n")
        ad = [r[0] for r in c.execute(f"select start from kernels where {name} like '%adam_kernel%' order by start")]
        mc = None
        for t in tb:
            if "memory_cop" in t.lower() and "start" in cols(c, t):
                mc = t
                break
        mcols = cols(c, mc) if mc else []
        print(f"memory-copy table: {mc}\n")
        if "--names" in sys.argv:
            for r in c.execute(f"select {name}, count(*), stream_id from kernels where {name} like '%elementwise%' or {name} like '%Functor%' group by {name}, stream_id order by 2 desc limit 12"):
                print("   ", r[1], "stream", r[2], r[0][:150])
            if mc:
                print("    copies:", c.execute(f"select count(*), min(size), max(size), sum(size) from {mc}").fetchone())
                for r in c.execute(f"select size, count(*), stream_id from {mc} group by size, stream_id order by 2 desc limit 10"):
                    print("    copy size", r[0], "x", r[1], "stream", r[2])
            for r in c.execute("select stream_id, count(*) from kernels group by stream_id"):
                print("    stream", r[0], "kernels", r[1])
        # streams: main = the one with most kernels; side (weight gradients) = the runner-up; exchange = where the buckets' scaling
        # kernels (div_ by the world size = MulFunctor) run
        st = c.execute("select stream_id, count(*) from kernels group by stream_id order by 2 desc").fetchall()
        main_s, side_s = st[0][0], st[1][0]
        ex = c.execute(f"select stream_id, count(*) from kernels where {name} like '%MulFunctor%' and stream_id != ? group by stream_id order by 2 desc", (main_s,)).fetchone()
        ex_s = ex[0] if ex else None
        print(f"streams: main {main_s} ({st[0][1]} kernels), weight-gradient side stream {side_s} ({st[1][1]}), exchange stream {ex_s} ({ex[1] if ex else 0} bucket scalings)\n")
        print("| step | backward ends (us after its first kernel; main / side stream) | buckets: scaling kernel starts (us) | bucket copies (gloo: D2H + H2D of 8 MB pieces): first start .. last end (us) | clip starts (us) | exposed = clip start - backward end (us) |")
        print("|---|---|---|---|---|---|")
        for i in range(1, len(ad)):
            w0, w1 = ad[i - 1], ad[i]
            ks = c.execute(f"select {name}, start, end, stream_id from kernels where start > ? and start <= ? order by start", (w0, w1)).fetchall()
            b0 = next((s for n, s, e, sid in ks if "loss_bwd_kernel" in n), None)
            clip = next((s for n, s, e, sid in ks if "sumsq_partial_kernel" in n), None)
            if b0 is None or clip is None:
                continue
            div = [s for n, s, e, sid in ks if sid == ex_s and "MulFunctor" in n and s >= b0]
            if not div:
                continue                                        # a local (no-exchange) step
            # backward's last kernel on the main stream: the embedding gradient (engine.py: the last launch of Engine._backward)
            mend = max([e for n, s, e, sid in ks if sid == main_s and b0 <= s < clip and "embed_bwd_kernel" in n] or
                       [max(e for n, s, e, sid in ks if sid == main_s and b0 <= s < clip)])
            send = max([e for n, s, e, sid in ks if sid == side_s and b0 <= s < clip] or [b0])
            bend = max(mend, send)
            cp = ""
            if mc:
                rows = c.execute(f"select start, end from {mc} where start > ? and start <= ? and size >= 1000000 and stream_id != ? order by start", (b0, clip, main_s)).fetchall()
                if rows:
                    cp = f"{len(rows)} copies, {(rows[0][0] - b0) / 1e3:.0f} .. {(max(r[1] for r in rows) - b0) / 1e3:.0f}"
            under = sum(1 for s in div if s < bend)
            print(f"| {i} | {(mend - b0) / 1e3:.0f} / {(send - b0) / 1e3:.0f} | {len(div)} buckets, {under} start under backward: {[round((s - b0) / 1e3) for s in div]} | {cp} | "
                  f"{(clip - b0) / 1e3:.0f} | {(clip - bend) / 1e3:.0f} |")
        print()


if __name__ == "__main__":
    main()
