#!/bin/bash
# One command for the 1 -> 8 GPU weak-scaling record of the headline block and the two DDP configs of BASELINE.json
# (configs[1] block, configs[3] dbm, configs[4] long; add "stack" to CONFIGS for configs[2]):
#   tools/scale.sh [out_dir]            ->  out_dir/scale_<config>.jsonl (one bench line per N), out_dir/scale_summary.txt
# Every N > 1 run is the driver's own launch line (one rank per GPU over RCCL, rendezvous on 127.0.0.1); ranks pin
# themselves to disjoint host cores (bench.py pin_rank_to_cores).  GPUS="1 2 4 8" CONFIGS="block dbm long" override.
set -u
root=$(cd "$(dirname "$0")/.." && pwd)
out=${1:-$root/gpurun_out/scale}
mkdir -p "$out"
GPUS=${GPUS:-"1 2 4 8"}
CONFIGS=${CONFIGS:-"block dbm long"}
STEPS=${STEPS:-50}; WARMUP=${WARMUP:-20}
export HSA_ENABLE_IPC_MODE_LEGACY=0
have=$(python -c 'import torch; print(torch.cuda.device_count())' 2>/dev/null || echo 0)
port=29511
: > "$out/scale_summary.txt"
for cfg in $CONFIGS; do
  : > "$out/scale_$cfg.jsonl"
  for n in $GPUS; do
    if [ "$n" -gt "$have" ]; then echo "$cfg n=$n: only $have GPU(s) visible, skipped" | tee -a "$out/scale_summary.txt"; continue; fi
    if [ "$n" -eq 1 ]; then
      line=$(cd "$root" && python bench.py --gpus 1 --steps $STEPS --warmup $WARMUP --config $cfg --no-cpu-baseline --no-projections 2>"$out/err_${cfg}_$n.log" | tail -1)
    else
      port=$((port + 1))
      line=$(cd "$root" && python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $port \
             bench.py --gpus $n --steps $STEPS --warmup $WARMUP --config $cfg --no-cpu-baseline --no-projections 2>"$out/err_${cfg}_$n.log" | grep '^{' | tail -1)
    fi
    echo "$line" >> "$out/scale_$cfg.jsonl"
    python - "$cfg" "$n" <<PY | tee -a "$out/scale_summary.txt"
import json, sys
cfg, n = sys.argv[1], int(sys.argv[2])
try:
    r = json.loads('''$line''')
    print(f"{cfg:6s} n={n}: {r['value'] / 1e6:8.3f} M tokens/s  {r['ms_per_step']:8.3f} ms/step  per GPU {r['value'] / n / 1e6:7.3f} M")
except Exception as e:
    print(f"{cfg:6s} n={n}: no result ({e})")
PY
  done
done
