"""Condense a rocprofv3 --kernel-trace --stats kernel_stats.csv into a short table for profiles/."""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
def short(n):
    n = re.sub(r"\(.*", "", n)
    n = re.sub(r"Cijk_(\w+?)_BBS.*?_(MT\d+x\d+x\d+)_.*", r"hipBLASLt GEMM \1 \2", n)
    n = re.sub(r"Custom_Cijk.*?(MT\d+x\d+x\d+).*", r"hipBLASLt GEMM custom \1", n)
    n = n.replace("void ", "").replace("at::native::", "")
    return n[:90]
print(f"| kernel | calls | total ms | avg us | % |\n|---|---|---|---|---|")
for r in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 25]:
    print(f"| {short(r['Name'])} | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.2f} | {float(r['AverageNs'])/1e3:.1f} | {float(r['TotalDurationNs'])/tot*100:.1f} |")
print(f"\ntotal GPU kernel time {tot/1e6:.2f} ms")
