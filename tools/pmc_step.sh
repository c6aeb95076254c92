#!/bin/bash
# tools/pmc_step.sh <tag> [configs...]: rocprofv3 --pmc passes (each on its own, no tracing) over a few steps of bench.py --config <cfg>:
# FETCH_SIZE, WRITE_SIZE, two SQ sets, GRBM for EVERY kernel of the step as the step runs it -- the fused head / tail / projection
# kernels included (tools/measure_cfg.sh covers the scans at the config's shape through tools/kbench.py) -> gpurun_out/<tag>/<cfg>_step_pmc.md
tag=$1; shift
if [ $# -eq 0 ]; then set -- block stack; fi
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
A="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY"
B="SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
C="SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES"
D="GRBM_GUI_ACTIVE GRBM_COUNT"
cd /tmp && export TMPDIR=/tmp
for cfg in "$@"; do
  csvs=""
  for k in f:FETCH_SIZE w:WRITE_SIZE a:"$A" b:"$B" c:"$C" d:"$D"; do
    n=${k%%:*}; ctrs=${k#*:}
    rocprofv3 --pmc $ctrs -d $O/pmc_${cfg}_$n -o p --output-format csv -- python $R/bench.py --config $cfg --steps 3 --warmup 2 --no-cpu-baseline --no-projections --no-extra-configs --no-graph > $O/pmc_${cfg}_$n.log 2>&1 || tail -3 $O/pmc_${cfg}_$n.log
    csvs="$csvs $O/pmc_${cfg}_$n/*counter_collection.csv"
  done
  python $R/tools/pmc_table.py --json $O/${cfg}_step_traffic.json $csvs > $O/${cfg}_step_pmc.md
  rm -rf $O/pmc_${cfg}_*
  grep -E "^###|HBM traffic|VALU pipe|waves per SIMD" $O/${cfg}_step_pmc.md
done
