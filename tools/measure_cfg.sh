#!/bin/bash
# tools/measure_cfg.sh <tag> [configs...]: per BASELINE config other than the headline -- the evidence set profiles/ holds per shape:
#   gpurun_out/<tag>/<cfg>_bench.json          python bench.py --config <cfg>
#   gpurun_out/<tag>/<cfg>_kernel_stats.md     rocprofv3 --kernel-trace --stats of bench.py --config <cfg> --steps 10
#   gpurun_out/<tag>/<cfg>_pmc.md / _traffic.json   FETCH_SIZE / WRITE_SIZE / SQ counter passes over the scans at the config's shape
#                                              (tools/kbench.py with KB_SHAPE; each --pmc pass on its own, no tracing)
tag=$1; shift
if [ $# -eq 0 ]; then set -- stack long dbm; fi
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
# block_coarse (round 6): the headline block with the 128-element checkpoint layout (VMS_X_LAYOUT=1: what a memory-filling training job
# gets from the "auto" policy) -- the traffic profile of bench.py's `block_coarse_checkpoints` extra line
declare -A DUAL=( [block]=dual [block_coarse]=dual [stack]=dual [long]=dual [dbm]= )
declare -A SHAPE=( [block]=8,1024,8192,16 [block_coarse]=8,1024,8192,16 [stack]=8,768,3136,16 [long]=1,768,65536,16 [dbm]=4,512,2304,16 )
A="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY"
B="SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
D="GRBM_GUI_ACTIVE GRBM_COUNT"
for cfg0 in "$@"; do
  cd $R
  cfg=$cfg0; unset VMS_X_LAYOUT
  if [ $cfg0 = block_coarse ]; then export VMS_X_LAYOUT=1; fi
  bcfg=${cfg%_coarse}
  python bench.py --config $bcfg --no-cpu-baseline --no-extra-configs 2>/dev/null | tail -1 > $O/${cfg}_bench.json
  KB_SHAPE=${SHAPE[$cfg]} python tools/kbench.py fwd bwd ${DUAL[$cfg]} 2>&1 | grep -v amdgpu.ids > $O/${cfg}_kbench.txt
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --stats -d $O/prof -o p --output-format csv -- python $R/bench.py --config $bcfg --steps 10 --warmup 20 --no-projections --no-cpu-baseline --no-extra-configs > $O/${cfg}_prof.log 2>&1
  python $R/tools/prof_summary.py $O/prof/p_kernel_stats.csv 30 > $O/${cfg}_kernel_stats.md
  python $R/tools/step_trace.py $O/prof/p_kernel_trace.csv 10 > $O/${cfg}_step_trace.txt 2>/dev/null
  rm -rf $O/prof
  csvs=""
  for k in f:FETCH_SIZE w:WRITE_SIZE a:"$A" b:"$B" d:"$D"; do
    n=${k%%:*}; ctrs=${k#*:}
    KB_SHAPE=${SHAPE[$cfg]} rocprofv3 --pmc $ctrs -d $O/pmc_${cfg}_$n -o p --output-format csv -- python $R/tools/kbench.py fwd bwd ${DUAL[$cfg]} > $O/pmc_${cfg}_$n.log 2>&1 || tail -3 $O/pmc_${cfg}_$n.log
    csvs="$csvs $O/pmc_${cfg}_$n/*counter_collection.csv"
  done
  python $R/tools/pmc_table.py --json $O/${cfg}_traffic.json $csvs > $O/${cfg}_pmc.md
  rm -rf $O/pmc_${cfg}_*
  cut -c1-200 $O/${cfg}_bench.json; cat $O/${cfg}_kbench.txt; grep -E "^###|HBM traffic|VALU pipe|waves per SIMD" $O/${cfg}_pmc.md; rm -f $O/${cfg}_prof.log
done
