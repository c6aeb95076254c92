#!/bin/bash
# tools/trace_cfg.sh <config> <tag>: rocprofv3 kernel trace of `bench.py --config <config>` (10 steps); the kernels of the last step in launch
# order (start offset, duration, name) -> gpurun_out/<tag>/<config>_trace.txt
cfg=$1; tag=$2
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/prof_$cfg -o p --output-format csv -- python $R/bench.py --config $cfg --steps 10 --warmup 20 --no-projections --no-cpu-baseline --no-graph --no-extra-configs > $O/prof_$cfg.log 2>&1
cd $R
python tools/step_trace.py $O/prof_$cfg/p_kernel_trace.csv 30 -v > $O/${cfg}_trace.txt 2>/dev/null
rm -rf $O/prof_$cfg
