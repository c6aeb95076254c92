#!/bin/bash
# tools/traffic_ab.sh <what> <libtag>... : FETCH_SIZE / WRITE_SIZE (separate passes) of the scan kernels per A/B library
what=$1; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/traffic_ab; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for t in "$@"; do
  for c in FETCH_SIZE WRITE_SIZE; do
    VMS_HIP_LIB=$R/tools/build/libvms_$t.so rocprofv3 --pmc $c -d $O/$t$c -o p --output-format csv -- python $R/tools/kbench.py $what > $O/$t$c.log 2>&1
  done
  python $R/tools/traffic.py $O/${t}FETCH_SIZE/*counter_collection.csv $O/${t}WRITE_SIZE/*counter_collection.csv | python -c "
import json,sys
d=json.load(sys.stdin)
for k,v in d.items(): print('$t', k, {a: round(b/1e6,1) for a,b in v.items()})"
  rm -rf $O/${t}FETCH_SIZE $O/${t}WRITE_SIZE
done
