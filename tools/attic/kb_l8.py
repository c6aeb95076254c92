import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(ROOT, "video-mamba-suite_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch, selective_scan_cuda, vms_hip
from kb_dual import problem, timeit
for L in (3144, 3152):
    p = problem(0, b=8, d=768, L=L)
    f = lambda: selective_scan_cuda.fwd(p["u"], p["delta"], p["A"], p["B"], p["C"], p["D"], p["z"], p["bias"], True)
    t = timeit(f, 20, 5); kf = vms_hip.last_kernel()
    out, x, oz = f()
    bw = lambda: selective_scan_cuda.bwd(p["u"], p["delta"], p["A"], p["B"], p["C"], p["D"], p["z"], p["bias"], p["dout"], x, out, p["dz"], True, False)
    tb = timeit(bw, 20, 5); kb = vms_hip.last_kernel()
    print(L, "x pitch", x.stride(2), "fwd %.1f us (%s)  bwd %.1f us (%s)" % (t, kf, tb, kb))
