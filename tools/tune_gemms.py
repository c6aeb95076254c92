"""Tune the library GEMMs of the block configs with torch's TunableOp (hipBLASLt / rocBLAS solution search per GEMM shape) and write
the results file bench.py / the modules can load (tools/tune_gemms.py [configs...] -> profiles/tunableop_results.csv).
The in_proj / out_proj products of a block step are plain library GEMMs (mamba_ssm/ops/projections.py); torch picks hipBLASLt's
first heuristic answer for each, which runs them at 39-41 % of the dense bf16 MFMA peak at the benchmark shape."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "video-mamba-suite_amd"))
import torch
import torch.cuda.tunable as tun
import bench

out = os.path.join(ROOT, "profiles", "tunableop_results.csv")
cfgs = sys.argv[1:] or ["block", "stack", "long", "dbm"]
res = {}
for cfg in cfgs:   # untuned first
    r = bench.run(cfg, steps=20, warmup=10, cpu_base=False, projections=(cfg == "block"), graph=False)
    res[cfg] = {"untuned_ms": r["ms_per_step"], "untuned_proj": r.get("projections")}
tun.enable(True); tun.tuning_enable(True); tun.set_filename(out)
tun.set_max_tuning_duration(200); tun.set_max_tuning_iterations(50)
for cfg in cfgs:
    t0 = time.time()
    bench.run(cfg, steps=2, warmup=2, cpu_base=False, projections=(cfg == "block"), graph=False)   # tunes every new GEMM shape it meets
    res[cfg]["tune_s"] = time.time() - t0
tun.tuning_enable(False)
for cfg in cfgs:
    r = bench.run(cfg, steps=20, warmup=10, cpu_base=False, projections=(cfg == "block"), graph=False)
    res[cfg]["tuned_ms"] = r["ms_per_step"]; res[cfg]["tuned_proj"] = r.get("projections")
tun.write_file() if hasattr(tun, "write_file") else None
print(json.dumps(res, indent=1))
print("results:", len(tun.get_results()), "entries ->", tun.get_filename())
