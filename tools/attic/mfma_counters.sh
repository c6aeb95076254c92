#!/bin/bash
# MFMA counters of the projection GEMMs: tools/mfma_counters.sh <tag>
#   pass 1: rocprofv3 --kernel-trace --stats  (durations)      pass 2: --pmc MFMA counters (own run, no tracing)
tag=$1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/trace -o p --output-format csv -- python $R/tools/gemm_proj.py 5 > $O/trace.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE -d $O/pmc -o p --output-format csv -- python $R/tools/gemm_proj.py 5 > $O/pmc.log 2>&1
cd $R
python tools/mfma_summary.py $O/trace/p_kernel_trace.csv $O/pmc/p_counter_collection.csv > $O/mfma.md
cat $O/mfma.md
rm -rf $O/trace/*agent_info* $O/pmc/*agent_info*
