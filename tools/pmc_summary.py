"""Summarise a rocprofv3 --pmc counter_collection.csv: per kernel, mean counter values (+ durations from the kernel trace).
usage: python tools/pmc_summary.py <dir-with-csvs> [kernel-substring]"""
import csv, collections, glob, sys
d = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else ""
cc = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
kt = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in cc:
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-60:]
        if sub in r["Kernel_Name"]:
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = collections.defaultdict(list)
for f in kt:
    for r in csv.DictReader(open(f)):
        if sub in r["Kernel_Name"]:
            dur[r["Kernel_Name"].split("(")[0][-60:]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, c in agg.items():
    ds = dur.get(k, [0])
    print(f"{k}: launches={len(ds)} avg_us={sum(ds)/len(ds):.1f}")
    for n, v in sorted(c.items()):
        print(f"    {n:30s} {sum(v)/len(v):.4e}")
