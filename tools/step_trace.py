"""Per-step kernel timeline from a rocprofv3 --kernel-trace csv of bench.py: python tools/step_trace.py <csv> [steps]
Prints the kernels of the LAST step (the trace holds warmup + timed steps of equal kernel count) and the sums."""
import csv, sys, collections

def main():
    path, steps = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 13
    rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
    per = len(rows) // steps
    last = rows[-per:]
    # rotate so that the step starts at the first in_proj-side kernel after the longest idle gap
    t0 = int(last[0]["Start_Timestamp"])
    tot = 0.0
    small = 0
    small_t = 0.0
    groups = collections.OrderedDict()
    for r in last:
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        tot += d
        if d < 10:
            small += 1
            small_t += d
        name = r["Kernel_Name"]
        key = name.split("<")[0].split("(")[0][:60] if name.startswith("void") else name[:40]
        groups[key] = groups.get(key, 0.0) + d
        if "-v" in sys.argv:
            print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} {d:8.1f}  {name[:120]}")
    span = (int(last[-1]["End_Timestamp"]) - t0) / 1e3
    print(f"kernels per step {per}; busy {tot:.1f} us; span {span:.1f} us; kernels < 10 us: {small} ({small_t:.1f} us)")
    for k, v in sorted(groups.items(), key=lambda kv: -kv[1])[:25]:
        print(f"  {v:8.1f} us  {k}")

if __name__ == "__main__":
    main()
