"""HBM traffic per launch of the four hot kernels from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE in
separate runs, as MI355X_MICROARCH.md prescribes), written as JSON for bench.py's `roofline.traffic`.
  tools/pmc.sh f vms FETCH_SIZE -- python tools/kbench.py fwd bwd conv
  tools/pmc.sh w vms WRITE_SIZE -- python tools/kbench.py fwd bwd conv
  python tools/traffic.py gpurun_out/pmc_f/*counter_collection.csv gpurun_out/pmc_w/*counter_collection.csv > profiles/rNN_traffic.json
gfx950 corrections: FETCH_SIZE (KB) reports half of the bytes of wide coalesced reads -> x2; WRITE_SIZE (KB) matches
the byte count of plain stores (checked on causal_conv1d_fwd: 131072 KB = the 134.2 MB of `out`)."""
import collections, csv, json, sys
ENTRY = {"scan_fwd": "vms_selective_scan_fwd", "scan_bwd": "vms_selective_scan_bwd",
         "conv_fwd": "vms_causal_conv1d_fwd", "conv_bwd": "vms_causal_conv1d_bwd"}
out = collections.defaultdict(dict)
for path in sys.argv[1:]:
    agg = collections.defaultdict(float); disp = collections.defaultdict(set)
    for r in csv.DictReader(open(path)):
        for key, entry in ENTRY.items():
            if key in r["Kernel_Name"]:
                agg[(entry, r["Counter_Name"])] += float(r["Counter_Value"]); disp[(entry, r["Counter_Name"])].add(r["Dispatch_Id"])
    for (entry, ctr), v in agg.items():
        per = v / len(disp[(entry, ctr)]) * 1024.0
        if ctr == "FETCH_SIZE": out[entry]["fetch_bytes"] = 2.0 * per
        if ctr == "WRITE_SIZE": out[entry]["write_bytes"] = per
for e in out.values():
    if "fetch_bytes" in e and "write_bytes" in e: e["hbm_bytes"] = e["fetch_bytes"] + e["write_bytes"]
json.dump(out, sys.stdout, indent=1, sort_keys=True); print()
