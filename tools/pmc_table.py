"""Per-launch averages of rocprofv3 --pmc counter CSVs for every vms:: kernel (main and carry kernels separately), as a
markdown table, plus HBM traffic (FETCH_SIZE x2 + WRITE_SIZE, KB; gfx950 correction of MI355X_MICROARCH.md) and the SQ
ratios DESIGN.md argues from.  usage: pmc_table.py [--json out.json] a.csv b.csv ...
SQ_* cycle counters are quad-cycles summed over waves / SIMDs."""
import collections
import csv
import json
import re
import sys


def short(name):
    m = re.search(r"vms::(\w+)", name)
    if not m:
        mm = re.search(r"_ZN3vms(\d+)", name)   # a name rocprofv3 left mangled
        if not mm:
            return None
        n = int(mm.group(1))
        rest = name[mm.end():]
        return rest[:n] + "<" + rest[n:n + 24] + ">"
    k = m.group(1)
    t = re.search(r"<(.*)>", name)
    if t:
        args = re.sub(r"__hip_bfloat16|hip_bfloat16|bool_Accum", "bf16", t.group(1).replace(" ", "")).replace("(bool)", "").replace("(int)", "")
        k += "<" + args.split(">")[0][:40] + ">"
    return k


def main():
    args = sys.argv[1:]
    jpath = None
    if args and args[0] == "--json":
        jpath, args = args[1], args[2:]
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(lambda: collections.defaultdict(set))
    meta = {}
    for path in args:
        for r in csv.DictReader(open(path)):
            k = short(r["Kernel_Name"])
            if k is None:
                continue
            c = r["Counter_Name"]
            agg[k][c] += float(r["Counter_Value"])
            disp[k][c].add((path, r["Dispatch_Id"]))
            meta[k] = f"grid {r['Grid_Size']} wg {r['Workgroup_Size']} vgpr {r['VGPR_Count']} lds {r['LDS_Block_Size']}"
    out = {}
    for k in sorted(agg):
        v = {c: agg[k][c] / max(len(disp[k][c]), 1) for c in agg[k]}
        g = v.get
        print(f"### {k}  ({meta[k]})\n\n| counter | per launch |\n|---|---|")
        for c in sorted(v):
            print(f"| {c} | {v[c]:,.0f} |")
        print()
        e = {}
        if g("FETCH_SIZE") is not None and g("WRITE_SIZE") is not None:
            e = {"fetch_bytes": 2048.0 * g("FETCH_SIZE"), "write_bytes": 1024.0 * g("WRITE_SIZE")}
            e["hbm_bytes"] = e["fetch_bytes"] + e["write_bytes"]
            print(f"* HBM traffic per launch = {e['hbm_bytes'] / 1e6:,.1f} MB (read {e['fetch_bytes'] / 1e6:,.1f}, written {e['write_bytes'] / 1e6:,.1f})")
        if g("SQ_WAVE_CYCLES"):
            wc = g("SQ_WAVE_CYCLES")
            for c in ("SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_LDS",
                      "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_SCA"):
                if g(c) is not None:
                    print(f"* {c} / SQ_WAVE_CYCLES = {g(c) / wc:.3f}")
            if g("SQ_BUSY_CYCLES") and g("SQ_WAVES"):
                # SQ_BUSY_CYCLES counts per SE (32 on the chip / 8 XCD x 4): mean resident waves per SIMD = wave-cycles / (1024 SIMD x kernel cycles)
                print(f"* SQ_WAVE_CYCLES / SQ_WAVES = {wc / g('SQ_WAVES'):,.0f} quad-cycles resident per wave")
        if g("SQ_INSTS_VALU") and g("SQ_WAVES"):
            print(f"* VALU wave-instructions per wave = {g('SQ_INSTS_VALU') / g('SQ_WAVES'):,.0f}")
            e["insts_valu"] = g("SQ_INSTS_VALU")
        if g("GRBM_GUI_ACTIVE"):
            # GRBM_GUI_ACTIVE comes back summed over the 8 XCDs (checked against the kernel's duration: 12.04 M / 8 = 1.5 M
            # cycles = 0.68 ms at 2.2 GHz for a 0.68 ms launch)
            cyc = g("GRBM_GUI_ACTIVE") / 8.0
            print(f"* GRBM_GUI_ACTIVE / 8 XCDs = {cyc:,.0f} cycles per launch")
            e["gui_cycles"] = cyc
            if g("SQ_WAVE_CYCLES"):
                print(f"* mean waves per SIMD while running (4 x SQ_WAVE_CYCLES / 1024 SIMDs / cycles) = {4 * g('SQ_WAVE_CYCLES') / (1024 * cyc):.2f}")
            if g("SQ_ACTIVE_INST_VALU"):
                print(f"* VALU pipe busy (4 x SQ_ACTIVE_INST_VALU / 1024 SIMDs / cycles) = {4 * g('SQ_ACTIVE_INST_VALU') / (1024 * cyc):.3f}")
        print()
        if e:
            out[k] = e
    if jpath:
        # per C-ABI entry point: the kernels one call launches (main + carry), per-launch averages added up
        entries = {}
        for key, entry in (("scan_bwd_pair4_dual", "vms_selective_scan_bwd_dual"), ("scan_bwd", "vms_selective_scan_bwd"),
                           ("scan_fwd", "vms_selective_scan_fwd")):
            ks = [k for k in out if key in k and "hbm_bytes" in out[k] and not any(k in v["kernels"] for v in entries.values())]
            if ks:
                # template variants of one kernel (the two directions) are the same work: mean; different kernels of one call add up
                bases = collections.defaultdict(list)
                for k in ks:
                    bases[k.split("<")[0]].append(out[k])
                tot = lambda f: sum(sum(v[f] for v in vs) / len(vs) for vs in bases.values())
                entries[entry] = {"hbm_bytes": tot("hbm_bytes"), "fetch_bytes": tot("fetch_bytes"), "write_bytes": tot("write_bytes"),
                                  "kernels": ks}
        json.dump(dict(entries, kernels=out), open(jpath, "w"), indent=1, sort_keys=True)


main()
