"""Per-launch averages of rocprofv3 --pmc counter CSVs for the scan kernels, as a markdown table with the ratios
DESIGN.md argues from (tools/sq_profile.sh).  usage: sq_summary.py a.csv b.csv ...
SQ_* cycle counters are in quad-cycles (4 shader clocks) summed over all waves / SIMDs (MI355X_MICROARCH.md)."""
import collections
import csv
import sys

KEYS = ("scan_bwd_sp_kernel", "scan_fwd_pair_kernel", "scan_bwd_pair4_kernel", "scan_bwd_pair_kernel", "scan_fwd_lds_kernel", "conv_fwd_kernel", "conv_bwd_kernel")


def main():
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(lambda: collections.defaultdict(set))
    meta = {}
    for path in sys.argv[1:]:
        for r in csv.DictReader(open(path)):
            name = r["Kernel_Name"]
            k = next((x for x in KEYS if x in name), None)
            if k is None:
                continue
            c = r["Counter_Name"]
            agg[k][c] += float(r["Counter_Value"])
            disp[k][c].add((path, r["Dispatch_Id"]))
            meta[k] = {x: r.get(x) for x in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Workgroup_Size", "Grid_Size")}
    for k in KEYS:
        if k not in agg:
            continue
        v = {c: agg[k][c] / max(len(disp[k][c]), 1) for c in agg[k]}
        print(f"### {k}  ({meta[k]})\n")
        print("| counter | per launch |")
        print("|---|---|")
        for c in sorted(v):
            print(f"| {c} | {v[c]:,.0f} |")
        g = v.get
        print()
        if g("SQ_WAVE_CYCLES"):
            wc = g("SQ_WAVE_CYCLES")
            for c in ("SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_LDS",
                      "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_SCA"):
                if g(c) is not None:
                    print(f"* {c} / SQ_WAVE_CYCLES = {g(c) / wc:.3f}")
        if g("SQ_INSTS_VALU") and g("SQ_WAVES"):
            print(f"* VALU wave-instructions per wave = {g('SQ_INSTS_VALU') / g('SQ_WAVES'):,.0f}")
        if g("SQ_INSTS_VALU") and g("SQ_BUSY_CYCLES"):
            print(f"* SQ_INSTS_VALU / SQ_BUSY_CYCLES = {g('SQ_INSTS_VALU') / g('SQ_BUSY_CYCLES'):.3f}")
        if g("SQ_INST_CYCLES_VALU") and g("SQ_INSTS_VALU"):
            print(f"* SQ_INST_CYCLES_VALU / SQ_INSTS_VALU = {g('SQ_INST_CYCLES_VALU') / g('SQ_INSTS_VALU'):.2f} (quad-cycles of issue per VALU instruction)")
        if g("SQ_LDS_BANK_CONFLICT") is not None and g("SQ_LDS_IDX_ACTIVE"):
            print(f"* SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = {g('SQ_LDS_BANK_CONFLICT') / g('SQ_LDS_IDX_ACTIVE'):.3f}")
        if g("GRBM_GUI_ACTIVE"):
            print(f"* GRBM_GUI_ACTIVE = {g('GRBM_GUI_ACTIVE'):,.0f} cycles per launch")
        print()


main()
