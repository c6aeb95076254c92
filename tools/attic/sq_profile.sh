#!/bin/bash
# SQ counter passes over the scan kernels at the benchmark shape: tools/sq_profile.sh <tag> [kbench args...]
# Three rocprofv3 --pmc passes (<= 8 SQ counters each, no tracing), each over `python tools/kbench.py fwd bwd`;
# gpurun_out/<tag>/sq_{a,b,c}/ hold the raw CSVs, tools/sq_summary.py turns them into profiles/<tag>_sq_scan.md.
tag=$1; shift
if [ $# -eq 0 ]; then set -- fwd bwd; fi
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
A="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY"
B="SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
C="SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INST_CYCLES_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA"
D="GRBM_GUI_ACTIVE GRBM_COUNT"
for k in a:"$A" b:"$B" c:"$C" d:"$D"; do
  n=${k%%:*}; ctrs=${k#*:}
  rocprofv3 --pmc $ctrs -d $O/sq_$n -o p --output-format csv -- python $R/tools/kbench.py "$@" > $O/sq_$n.log 2>&1 || tail -3 $O/sq_$n.log
  rm -rf $O/sq_$n/*agent_info*
done
cd $R
python tools/sq_summary.py $O/sq_a/*counter_collection.csv $O/sq_b/*counter_collection.csv $O/sq_c/*counter_collection.csv $O/sq_d/*counter_collection.csv > $O/sq_scan.md
cat $O/sq_scan.md
