#!/bin/bash
# per-kernel times of the rows forward for a given library build: tools/kb_rows.sh <tag> [lib]
tag=$1; lib=$2
cd /tmp && export TMPDIR=/tmp
[ -n "$lib" ] && export VMS_HIP_LIB=$lib
VMS_SCAN_IMPL=rows rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/kb_$tag -o p --output-format csv -- python $GRAFT_REPO_ROOT/tools/kbench.py fwd > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
echo "== $tag"; python tools/prof_summary.py $(ls gpurun_out/kb_$tag/*kernel_stats.csv | head -1) 4 | grep rows
