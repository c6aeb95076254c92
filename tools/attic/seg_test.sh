cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02i; mkdir -p $O
for s in 1 2 4; do
  VMS_BWD_SEGMENTS=$s rocprofv3 --kernel-trace --stats -d $O/seg$s -o p --output-format csv -- python $R/tools/kbench.py bwd > $O/seg$s.log 2>&1
  echo "== segments $s" >> $O/seg.txt
  grep scan_bwd $O/seg$s.log >> $O/seg.txt
  python - <<PY >> $O/seg.txt
import csv,glob
f=glob.glob("$O/seg$s/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'scan' in r['Name']: print(r['Name'][:60], r['Calls'], r['AverageNs'])
PY
  rm -rf $O/seg$s
done
cat $O/seg.txt
