#!/bin/bash
# usage: tools/pmc.sh <tag> <kernel-substr> <counters...> -- <command...>
# runs rocprofv3 --pmc (own pass, no tracing), prints per-launch averages for kernels matching substr
tag=$1; sub=$2; shift 2
ctrs=(); while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done; shift
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag; mkdir -p $GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc "${ctrs[@]}" -d $out -o p --output-format csv -- "$@" > $out.log 2>&1 || tail -5 $out.log
python - <<PY
import csv, glob, collections
f = glob.glob("$out/*counter_collection.csv")[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    if "$sub" not in k: continue
    agg[k[:50]][r["Counter_Name"]] += float(r["Counter_Value"]); disp[k[:50]].add(r["Dispatch_Id"])
for k, v in agg.items():
    n = len(disp[k])
    print(k, "launches", n)
    for c, x in sorted(v.items()): print(f"   {c:32s} {x/n:16.0f}")
PY
