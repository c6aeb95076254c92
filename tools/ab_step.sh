#!/bin/bash
# tools/ab_step.sh <out> <config> <variant tags...>: A/B of LIBRARY builds inside a step.  `VMS_HIP_LIB=` selects another library only for the ctypes
# binding (and with it the unfused Python nodes: another step); the compiled binding loads vms_hip/libvms_hip.so by path, so the in-tree library itself is
# swapped between alternating bench.py runs on one box (three rounds) and restored at the end.  Variants: tools/variant.sh <tag> -D... -> tools/build/libvms_<tag>.so
# (include a plain `tools/variant.sh base` build: the same compiler flags as the variants).  Prints ms per step and us per step of every entry point.
out=$1; cfg=$2; shift 2
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$out; mkdir -p $O; cd $R
cp video-mamba-suite_amd/vms_hip/libvms_hip.so /tmp/libvms_orig.so
for rep in 1 2 3; do
  for v in "$@"; do
    cp tools/build/libvms_$v.so video-mamba-suite_amd/vms_hip/libvms_hip.so
    python bench.py --config $cfg --no-cpu-baseline --no-projections --no-extra-configs 2>/dev/null | tail -1 > $O/${v}_$rep.json
  done
done
cp /tmp/libvms_orig.so video-mamba-suite_amd/vms_hip/libvms_hip.so
python - $O "$@" <<'PY'
import json, sys
o, tags = sys.argv[1], sys.argv[2:]
for v in tags:
    ms, ks = [], {}
    for r in (1, 2, 3):
        d = json.load(open(f"{o}/{v}_{r}.json"))
        ms.append(round(d["ms_per_step"], 3))
        for k, x in d["kernels"].items():
            ks.setdefault(k, []).append(round(x["ms_per_step"] * 1e3))
    print(f"{v:14s} {ms}", {k.replace("vms_", ""): v2 for k, v2 in ks.items() if k != "vms_param_prep"})
PY
