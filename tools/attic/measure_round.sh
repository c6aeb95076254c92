#!/bin/bash
# One GPU call that produces everything profiles/ holds for a build: tools/measure_round.sh <tag>
#   gpurun_out/<tag>/bench.json          python bench.py   (defaults: 50 timed steps after 20 warm-up steps)
#   gpurun_out/<tag>/kernel_stats.csv    rocprofv3 --kernel-trace --stats of python bench.py --steps 10 --warmup 20 --no-projections --no-cpu-baseline
#   gpurun_out/<tag>/kernel_trace.csv    (same run; tools/step_trace.py reads it)
#   gpurun_out/<tag>/kbench.txt          python tools/kbench.py fwd bwd conv norm
#   gpurun_out/<tag>/pmc_{f,w}           FETCH_SIZE / WRITE_SIZE passes (separate runs, no tracing) -> traffic.json
tag=$1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
python bench.py 2>/dev/null | tail -1 > $O/bench.json
python tools/kbench.py fwd bwd conv norm 2>&1 | grep -v amdgpu.ids > $O/kbench.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/prof -o p --output-format csv -- python $R/bench.py --steps 10 --warmup 20 --no-projections --no-cpu-baseline > $O/prof.log 2>&1
cp $O/prof/p_kernel_stats.csv $O/kernel_stats.csv; cp $O/prof/p_kernel_trace.csv $O/kernel_trace.csv
tail -1 $O/prof.log | grep '^{' > $O/bench_under_rocprof.json
for c in f:FETCH_SIZE w:WRITE_SIZE; do
  k=${c%%:*}; ctr=${c##*:}
  rocprofv3 --pmc $ctr -d $O/pmc_$k -o p --output-format csv -- python $R/tools/kbench.py fwd bwd conv > $O/pmc_$k.log 2>&1
done
cd $R
python tools/traffic.py $O/pmc_f/*counter_collection.csv $O/pmc_w/*counter_collection.csv > $O/traffic.json
python tools/prof_summary.py $O/kernel_stats.csv 30 > $O/kernel_stats.md
python tools/step_trace.py $O/kernel_trace.csv 30 > $O/step_trace.txt
rm -rf $O/prof $O/pmc_f/*agent_info* $O/pmc_w/*agent_info*
cat $O/bench.json | cut -c1-250; cat $O/kbench.txt; cat $O/traffic.json; head -3 $O/step_trace.txt
