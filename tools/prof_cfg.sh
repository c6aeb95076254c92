#!/bin/bash
# tools/prof_cfg.sh <config> <tag>: rocprofv3 kernel stats of `bench.py --config <config>` (10 steps) -> gpurun_out/<tag>/kernel_stats.md
cfg=$1; tag=$2
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/prof -o p --output-format csv -- python $R/bench.py --config $cfg --steps 10 --warmup 20 --no-projections --no-cpu-baseline > $O/prof.log 2>&1
cd $R
python tools/prof_summary.py $O/prof/p_kernel_stats.csv 40 > $O/kernel_stats.md
python tools/step_trace.py $O/prof/p_kernel_trace.csv 10 > $O/step_trace.txt 2>/dev/null
rm -rf $O/prof
cat $O/kernel_stats.md; head -20 $O/step_trace.txt
