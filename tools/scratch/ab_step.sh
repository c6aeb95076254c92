mkdir -p gpurun_out/r06h
cp video-mamba-suite_amd/vms_hip/libvms_hip.so /tmp/libvms_orig.so
for rep in 1 2 3; do
for v in v_base v_x8aux0 v_bwdld v_fwdld; do
  cp tools/build/libvms_$v.so video-mamba-suite_amd/vms_hip/libvms_hip.so
  python bench.py --config block --no-cpu-baseline --no-projections --no-extra-configs 2>/dev/null | tail -1 > gpurun_out/r06h/${v}_$rep.json
done; done
cp /tmp/libvms_orig.so video-mamba-suite_amd/vms_hip/libvms_hip.so
python - <<'PY'
import json
for v in ("v_base","v_x8aux0","v_bwdld","v_fwdld"):
    ms=[]; ks={}
    for r in (1,2,3):
        d=json.load(open(f"gpurun_out/r06h/{v}_{r}.json"))
        ms.append(round(d["ms_per_step"],3))
        for k,x in d["kernels"].items(): ks.setdefault(k,[]).append(round(x["ms_per_step"]*1e3))
    print(v, ms, {k.replace("vms_",""):v2 for k,v2 in ks.items() if k!="vms_param_prep"})
PY
