"""LDS pipe vs MFMA pipe per kernel from one rocprofv3 PMC pass (SQ_INSTS_LDS, SQ_WAIT_INST_LDS, SQ_ACTIVE_INST_LDS,
SQ_LDS_ADDR_CONFLICT, SQ_LDS_IDX_ACTIVE, SQ_INSTS_VALU, SQ_INSTS_MFMA, SQ_WAVE_CYCLES; GRBM_GUI_ACTIVE).
LDS busy % = SQ_LDS_IDX_ACTIVE (LDS-array cycles, summed over the chip's 256 CUs) / (GRBM_GUI_ACTIVE / 8 XCDs x 256 CUs);
MFMA busy % (from the instruction count) = SQ_INSTS_MFMA x 32 cycles / (GUI / 8 x 1024 SIMDs)  [32x32x16 bf16 = 32 cycles].
usage: python tools/pmc_lds.py <dir> <steps> > profiles/xxx_pmc_lds.md"""
import collections, csv, glob, sys

d, steps = sys.argv[1], float(sys.argv[2])
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")[:48]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (r.get("Dispatch_Id"), k)
        if key not in seen:
            seen.add(key); cnt[k] += 1
print("| kernel | launches/step | LDS-array busy % | MFMA busy % (insts x 32) | LDS insts / MFMA inst | VALU insts / MFMA inst | LDS-issue stall % of wave cycles | LDS addr-conflict % of LDS cycles |")
print("|---|---|---|---|---|---|---|---|")
rows = sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_INSTS_MFMA", 0))
for k, c in rows[:14]:
    gui = c.get("GRBM_GUI_ACTIVE", 0)
    if gui <= 0 or c.get("SQ_INSTS_MFMA", 0) <= 0:
        continue
    cyc = gui / 8.0
    lds = c.get("SQ_LDS_IDX_ACTIVE", 0)
    mf = c["SQ_INSTS_MFMA"]
    wc = c.get("SQ_WAVE_CYCLES", 0) or 1
    print(f"| {k} | {cnt[k]/steps:.0f} | {100*lds/(cyc*256):.1f} | {100*mf*32/(cyc*1024):.1f} | {c.get('SQ_INSTS_LDS',0)/mf:.2f} | "
          f"{c.get('SQ_INSTS_VALU',0)/mf:.2f} | {100*c.get('SQ_WAIT_INST_LDS',0)/wc:.1f} | {100*c.get('SQ_LDS_ADDR_CONFLICT',0)/(lds or 1):.1f} |")
