"""Per-kernel MFMA evidence for the projection GEMMs (tools/mfma_counters.sh): kernel-trace durations joined with the
MFMA counters of a separate --pmc pass, per kernel name (launch-averaged).
  mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs)   -- share of SIMD-cycles the matrix pipe is busy
  flops     = SQ_INSTS_VALU_MFMA_MOPS_BF16 x 512                                    -- MFMA flops issued (counter unit: 512 flops)
usage: mfma_summary.py kernel_trace.csv counter_collection.csv"""
import collections
import csv
import sys

trace, pmc = sys.argv[1], sys.argv[2]
dur = collections.defaultdict(list)
for r in csv.DictReader(open(trace)):
    dur[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(set))
for r in csv.DictReader(open(pmc)):
    agg[r["Kernel_Name"]][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[r["Kernel_Name"]][r["Counter_Name"]].add(r["Dispatch_Id"])
print("| kernel | launches | avg us | MFMA flops/launch (counter) | TFLOP/s | frac of 2.5 PF | SQ_VALU_MFMA_BUSY_CYCLES | GRBM_GUI_ACTIVE | mfma_busy |")
print("|---|---|---|---|---|---|---|---|---|")
rows = []
for k, c in agg.items():
    n = {x: max(len(cnt[k][x]), 1) for x in c}
    mops = c.get("SQ_INSTS_VALU_MFMA_MOPS_BF16", 0) / n.get("SQ_INSTS_VALU_MFMA_MOPS_BF16", 1)
    if mops == 0:
        continue
    busy = c["SQ_VALU_MFMA_BUSY_CYCLES"] / n["SQ_VALU_MFMA_BUSY_CYCLES"]
    grbm = c["GRBM_GUI_ACTIVE"] / n["GRBM_GUI_ACTIVE"]
    us = sum(dur.get(k, [0])) / max(len(dur.get(k, [])), 1)
    flops = mops * 512
    tf = flops / (us * 1e-6) / 1e12 if us else 0
    rows.append((us * len(dur.get(k, [])), f"| `{k[:70]}` | {len(dur.get(k, []))} | {us:.1f} | {flops:.3e} | {tf:.0f} | {tf / 2500:.3f} | {busy:,.0f} | {grbm:,.0f} | {busy / (grbm / 8 * 1024):.3f} |"))
for _, line in sorted(rows, reverse=True):
    print(line)
