"""The backward scan alone at the bench shape, both directions, with and without dz accumulation, twice (A/B of kernel builds:
VMS_HIP_LIB=tools/build/libvms_<tag>.so, VMS_X_LAYOUT=1).  usage: python tools/kb_bwd_only.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-mamba-suite_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch, selective_scan_cuda
from kb_dual import problem, timeit
p = problem(0)
a = torch.randn(4096, 4096, device="cuda", dtype=torch.bfloat16)
for _ in range(30): a @ a
for rev in (False, True, False, True):
    bw = lambda acc: selective_scan_cuda.bwd(p["u"], p["delta"], p["A"], p["B"], p["C"], p["D"], p["z"], p["bias"], p["dout"], p["x"], p["out"],
                                             p["dz"], True, False, reverse=rev, accumulate_dz=acc)
    print(f"rev={rev!s:5} bwd {timeit(lambda: bw(False), 20, 5):7.1f} / acc {timeit(lambda: bw(True), 20, 5):7.1f} us", end="   ")
print()
