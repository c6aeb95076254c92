#!/bin/bash
# tools/pmc_kb.sh <tag> <KB_SHAPE> <kbench args...>: SQ + GRBM + traffic counter passes over tools/kbench.py at one shape -> gpurun_out/<tag>/pmc.md
tag=$1; shape=$2; shift 2
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
A="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY"
B="SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
D="GRBM_GUI_ACTIVE GRBM_COUNT"
cd /tmp && export TMPDIR=/tmp
csvs=""
for k in f:FETCH_SIZE w:WRITE_SIZE a:"$A" b:"$B" d:"$D"; do
  n=${k%%:*}; ctrs=${k#*:}
  KB_SHAPE=$shape rocprofv3 --pmc $ctrs -d $O/pmc_$n -o p --output-format csv -- python $R/tools/kb_short.py > $O/pmc_$n.log 2>&1 || tail -3 $O/pmc_$n.log
  csvs="$csvs $O/pmc_$n/*counter_collection.csv"
done
python $R/tools/pmc_table.py --json $O/traffic.json $csvs > $O/pmc.md
rm -rf $O/pmc_? $O/pmc_?.log
grep -v "^$" $O/pmc.md | grep -E "^###|^\*"
