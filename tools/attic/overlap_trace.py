"""Which kernels of a rocprofv3 --kernel-trace csv ran at the same time?  python tools/overlap_trace.py <kernel_trace.csv> [min_us]
Prints every kernel longer than min_us (default 100) with start / end relative to the first one, its stream (queue) and,
for each, the kernels whose interval intersects it with the length of the intersection."""
import csv, sys


def short(name):
    n = name.replace("void ", "").replace("vms::", "")
    return n.split("<")[0].split("(")[0][:44]


def main():
    path = sys.argv[1]
    min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 100.0
    rows = [r for r in csv.DictReader(open(path))]
    ks = []
    for r in rows:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if (e - s) / 1e3 >= min_us:
            ks.append((s, e, short(r["Kernel_Name"]), r.get("Queue_Id", "?"), r.get("VGPR_Count", "?"), r.get("LDS_Block_Size", "?"),
                       r.get("Workgroup_Size", "?"), r.get("Grid_Size", "?")))
    ks.sort()
    if not ks:
        print("no kernels >= %.0f us" % min_us); return
    t0 = ks[0][0]
    print(f"{'start us':>10} {'end us':>10} {'dur us':>8}  queue vgpr   lds   wg     grid  kernel | overlaps")
    for i, (s, e, n, q, v, l, w, g) in enumerate(ks):
        ov = []
        for j, (s2, e2, n2, *_rest) in enumerate(ks):
            if j == i or e2 <= s or s2 >= e:
                continue
            ov.append(f"{n2[:24]}:{(min(e, e2) - max(s, s2)) / 1e3:.0f}us")
        print(f"{(s - t0) / 1e3:10.1f} {(e - t0) / 1e3:10.1f} {(e - s) / 1e3:8.1f}  {q:>5} {v:>4} {l:>6} {w:>4} {g:>8}  {n} | {' '.join(ov) if ov else '-'}")


if __name__ == "__main__":
    main()
