"""MFMA-pipe utilisation per kernel from one rocprofv3 PMC pass (SQ_VALU_MFMA_BUSY_CYCLES, SQ_INSTS_MFMA, SQ_WAVE_CYCLES,
SQ_WAIT_ANY, SQ_WAIT_INST_ANY, SQ_ACTIVE_INST_ANY, SQ_LDS_BANK_CONFLICT, SQ_LDS_IDX_ACTIVE; GRBM_GUI_ACTIVE).
MI355X_MICROARCH.md: MFMA_BUSY counts cycles summed over the chip's 1024 SIMDs (32 per v_mfma_f32_32x32x16_bf16);
SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves.
usage: python tools/pmc_mfma.py <dir> <steps> > profiles/xxx_pmc_mfma.md"""
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
NSIMD = 1024
NXCD = 8   # GRBM_GUI_ACTIVE comes back summed over the 8 XCDs (cross-check: SQ_INSTS_MFMA x 32 cycles / 1024 SIMDs / kernel duration)
print("| kernel | launches/step | MFMA busy % of GPU-active cycles | MFMA insts/launch | wave time: active / issue-stalled / parked % | LDS bank-conflict % of LDS cycles |")
print("|---|---|---|---|---|---|")
rows = sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_VALU_MFMA_BUSY_CYCLES", 0))
for k, c in rows[:14]:
    gui = c.get("GRBM_GUI_ACTIVE", 0)
    if gui <= 0 or c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) <= 0:
        continue
    util = 100.0 * c["SQ_VALU_MFMA_BUSY_CYCLES"] / (gui / NXCD * NSIMD)
    wc = c.get("SQ_WAVE_CYCLES", 0) or 1
    lds = c.get("SQ_LDS_IDX_ACTIVE", 0) or 1
    print(f"| {k} | {cnt[k]/steps:.0f} | {util:.1f} | {c.get('SQ_INSTS_MFMA',0)/cnt[k]:.3g} | {100*c.get('SQ_ACTIVE_INST_ANY',0)/wc:.0f} / "
          f"{100*c.get('SQ_WAIT_INST_ANY',0)/wc:.0f} / {100*c.get('SQ_WAIT_ANY',0)/wc:.0f} | {100*c.get('SQ_LDS_BANK_CONFLICT',0)/lds:.1f} |")
