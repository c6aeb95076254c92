import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(ROOT, "video-mamba-suite_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch, selective_scan_cuda
from kb_dual import problem, timeit
p = problem(0)
a = torch.randn(4096, 4096, device="cuda", dtype=torch.bfloat16)
for _ in range(30): a @ a
for rev in (False, True, False, True):
    f = lambda: selective_scan_cuda.fwd(p["u"], p["delta"], p["A"], p["B"], p["C"], p["D"], p["z"], p["bias"], True, reverse=rev)
    print(f"rev={rev!s:5} fwd {timeit(f, 30, 10):7.1f} us", end="   ")
print()
