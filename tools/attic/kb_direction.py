"""Forward / backward scan left-to-right vs right-to-left (reverse=True), with and without the accumulate flags, standalone:
is the 4-5 % gap between the two scans of a block step (rocprofv3 trace) the direction or the accumulation?
usage: python tools/kb_direction.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-mamba-suite_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import selective_scan_cuda
from kb_dual import problem, timeit

p = problem(0)
a = torch.randn(4096, 4096, device="cuda", dtype=torch.bfloat16)
for _ in range(30): a @ a
outz = torch.zeros_like(p["dout"])
for rev in (False, True):
    f = lambda: selective_scan_cuda.fwd(p["u"], p["delta"], p["A"], p["B"], p["C"], p["D"], p["z"], p["bias"], True, reverse=rev)
    fa = lambda: selective_scan_cuda.fwd(p["u"], p["delta"], p["A"], p["B"], p["C"], p["D"], p["z"], p["bias"], True, reverse=rev, out_z_into=outz)
    bw = lambda acc: selective_scan_cuda.bwd(p["u"], p["delta"], p["A"], p["B"], p["C"], p["D"], p["z"], p["bias"], p["dout"], p["x"], p["out"],
                                             p["dz"], True, False, reverse=rev, accumulate_dz=acc)
    print(f"reverse={rev!s:5}:  fwd {timeit(f, 20, 5):7.1f} us   fwd + out_z accumulate {timeit(fa, 20, 5):7.1f} us   "
          f"bwd {timeit(lambda: bw(False), 20, 5):7.1f} us   bwd + dz accumulate {timeit(lambda: bw(True), 20, 5):7.1f} us", flush=True)
