#!/bin/bash
# tools/ab_env.sh <out> <ENVVAR=value> [configs...]: bench.py lines of each config with and without one environment switch,
# alternating on the same box (box-to-box spread is larger than most single changes)
out=$1; sw=$2; shift 2
if [ $# -eq 0 ]; then set -- block stack; fi
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$out; mkdir -p $O; cd $R
for cfg in "$@"; do
  for rep in 1 2; do
    python bench.py --config $cfg --no-cpu-baseline --no-projections 2>/dev/null | tail -1 > $O/${cfg}_default_$rep.json
    env $sw python bench.py --config $cfg --no-cpu-baseline --no-projections 2>/dev/null | tail -1 > $O/${cfg}_switch_$rep.json
  done
  python - $O $cfg "$sw" <<'PY'
import json, sys
o, cfg, sw = sys.argv[1:4]
for tag in ("default", "switch"):
    ms = [json.load(open(f"{o}/{cfg}_{tag}_{r}.json"))["ms_per_step"] for r in (1, 2)]
    k = json.load(open(f"{o}/{cfg}_{tag}_1.json"))["kernels"]
    scan = {n: round(v["ms_per_step"], 3) for n, v in k.items() if "scan" in n}
    print(f"{cfg:6s} {tag if tag == 'default' else sw:24s} ms/step {ms[0]:.3f} {ms[1]:.3f}   scans/step {scan}")
PY
done
