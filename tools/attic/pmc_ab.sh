#!/bin/bash
# tools/pmc_ab.sh <outtag> "<counters>" <what> <libtag>... : one --pmc pass per A/B library, per-kernel averages
out=$1; ctrs=$2; what=$3; shift 3
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for t in "$@"; do
  VMS_HIP_LIB=$R/tools/build/libvms_$t.so rocprofv3 --pmc $ctrs -d $O/$t -o p --output-format csv -- python $R/tools/kbench.py $what > $O/$t.log 2>&1
  python - <<PY >> $O/summary.txt
import csv, glob, collections
f = glob.glob("$O/$t/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    if "scan_" not in k: continue
    k = k.split("<")[0].replace("void vms::", "")
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
for k in acc:
    print("$t", k, {c: round(v / len(n[k])) for c, v in acc[k].items()})
PY
  rm -rf $O/$t
done
cat $O/summary.txt
