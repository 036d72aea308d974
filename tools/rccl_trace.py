"""RCCL kernels of a data-parallel train step in a rocprofv3 kernel trace (rocpd database): which stream they run on and where
they sit relative to backward.  Written for the ONE-rank RCCL run a one-GPU box allows,

    rocprofv3 --kernel-trace -d gpurun_out/rccl_prof -o rccl -- python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 \
        --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 1 --steps 4 --warmup 2 --windows 1 --no-roofline
    python tools/rccl_trace.py gpurun_out/rccl_prof > profiles/r05_rccl_one_rank_trace.md

(bench.py under a launcher builds a process group also for one rank, and fastspeech2_amd/ddp.py then issues every bucket's
all_reduce), but it reads an N-rank trace the same way.  Per optimiser step of each rank: the RCCL kernels between the first
backward kernel (loss_bwd_kernel) and the clip pass (sumsq_partial_kernel): count, stream, start offsets, total duration, how
many start before backward's last kernel ends."""
import glob
import sqlite3
import sys


def main():
    dbs = sorted(glob.glob(sys.argv[1] + "/**/*.db", recursive=True))
    print(f"# RCCL kernels in the data-parallel step ({len(dbs)} rocpd databases under {sys.argv[1]})\n")
    for db in dbs:
        c = sqlite3.connect(db)
        tb = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
        if "kernels" not in tb:
            continue
        kc = [r[1] for r in c.execute("pragma table_info(kernels)")]
        name = "name" if "name" in kc else kc[0]
        ad = [r[0] for r in c.execute(f"select start from kernels where {name} like '%adam_kernel%' order by start")]
        if len(ad) < 3:
            continue
        cc = c.execute(f"select {name}, count(*), stream_id, avg(end - start) from kernels where lower({name}) like '%nccl%' or lower({name}) like '%rccl%' "
                       f"group by {name}, stream_id order by 2 desc").fetchall()
        print(f"## {db.split('/')[-1]}: {len(ad)} optimiser steps\n")
        if not cc:
            # a ONE-rank communicator reduces in place over one rank: RCCL has nothing to move and launches no device kernel for
            # it (the calls are still made and counted - tests/test_nccl_gpu.py); what the trace does show is each bucket's scaling
            # kernel (view.div_(world): MulFunctor) on the communication stream, i.e. WHEN every bucket was handed to RCCL
            print("no RCCL device kernel in this trace (one-rank communicator: in-place all_reduce over one rank moves nothing); "
                  "showing the buckets' scaling kernels on the communication stream instead\n")
            pat = "mulfunctor"
        else:
            pat = None
        hit = (lambda n: pat in n.lower()) if pat else (lambda n: "nccl" in n.lower() or "rccl" in n.lower())
        for n, k, sid, avg in cc:
            print(f"- `{n[:110]}`: {k} launches on stream {sid}, {avg / 1e3:.1f} us average")
        st = c.execute("select stream_id, count(*) from kernels group by stream_id order by 2 desc").fetchall()
        print(f"\nstreams by kernel count: {st} (first = the step's stream, second = the weight-gradient side stream)\n")
        main_s = st[0][0]
        if pat:
            ex = c.execute(f"select stream_id, count(*) from kernels where lower({name}) like '%mulfunctor%' and stream_id != ? group by stream_id order by 2 desc", (main_s,)).fetchone()
            print(f"communication stream: {ex[0] if ex else None} ({ex[1] if ex else 0} bucket scalings)\n")
            comm_s = ex[0] if ex else None
        print("| step | backward: loss_bwd start -> last main-stream kernel before clip (us) | RCCL kernels in that window: count, stream(s) | start offsets (us) | started before backward's end | sum of durations (us) | clip start - backward end (us) |")
        print("|---|---|---|---|---|---|---|")
        for i in range(1, len(ad)):
            ks = c.execute(f"select {name}, start, end, stream_id from kernels where start > ? and start <= ? order by start", (ad[i - 1], ad[i])).fetchall()
            b0 = next((s for n, s, e, sid in ks if "loss_bwd_kernel" in n), None)
            clip = next((s for n, s, e, sid in ks if "sumsq_partial_kernel" in n), None)
            if b0 is None or clip is None:
                continue
            mend = max(e for n, s, e, sid in ks if sid == main_s and b0 <= s < clip)
            rc = [(s, e, sid) for n, s, e, sid in ks if hit(n) and (pat is None or sid == comm_s) and b0 <= s <= clip + 1]
            if not rc:
                continue
            under = sum(1 for s, e, sid in rc if s < mend)
            print(f"| {i} | {(mend - b0) / 1e3:.0f} | {len(rc)} on {sorted({sid for _, _, sid in rc})} | {[round((s - b0) / 1e3) for s, _, _ in rc]} | {under} | "
                  f"{sum(e - s for s, e, _ in rc) / 1e3:.0f} | {(clip - mend) / 1e3:.0f} |")
        print()


if __name__ == "__main__":
    main()
