"""Per-kernel HBM traffic from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs as the guide prescribes).
Units/corrections (MI355X_MICROARCH.md §HBM): FETCH_SIZE/WRITE_SIZE are reported in KB (1024 B); on gfx950 FETCH_SIZE
counts 128-B requests as 64 B for wide coalesced reads -> doubled here; WRITE_SIZE is uncalibrated and used as reported.
usage: python tools/pmc_traffic.py <fetch_dir> <write_dir> <steps> [out.json]"""
import collections, csv, glob, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fastspeech2_amd._lib import kernel_source_sha

fetch_dir, write_dir, steps = sys.argv[1], sys.argv[2], float(sys.argv[3])
out_path = sys.argv[4] if len(sys.argv) > 4 else None


def load(d, counter):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")[:60]
            agg[k][0] += 1
            agg[k][1] += float(r["Counter_Value"])
    return agg


fe, wr = load(fetch_dir, "FETCH_SIZE"), load(write_dir, "WRITE_SIZE")
rows = []
for k in fe:
    n = fe[k][0]
    fetch_b = fe[k][1] * 1024 * 2           # KB -> B, x2 gfx950 correction
    write_b = wr.get(k, [0, 0.0])[1] * 1024
    rows.append((k, n, fetch_b, write_b))
rows.sort(key=lambda r: -(r[2] + r[3]))
tot = sum(r[2] + r[3] for r in rows)
print(f"| kernel | launches/step | fetch MB/launch | write MB/launch | total GB/step |")
print("|---|---|---|---|---|")
res = {}
for k, n, fb, wb in rows[:25]:
    print(f"| {k} | {n/steps:.1f} | {fb/n/1e6:.2f} | {wb/n/1e6:.2f} | {(fb+wb)/steps/1e9:.3f} |")
    res[k] = {"launches_per_step": n / steps, "fetch_bytes_per_launch": fb / n, "write_bytes_per_launch": wb / n}
print(f"\ntotal HBM traffic {tot/steps/1e9:.2f} GB/step")
if out_path:
    json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), FETCH x2 gfx950 correction", "kernels": res,
               "total_bytes_per_step": tot / steps, "kernel_source_sha": kernel_source_sha()}, open(out_path, "w"), indent=1)
