"""One backward launch through an A/B library (VMS_HIP_LIB) -- for builds that print their own in-kernel timers."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-mamba-suite_amd")); sys.path.insert(0, ROOT)
import torch
sys.path.insert(0, os.path.join(ROOT, "tools"))
from kb_dual import problem, bwd
p = problem(0)
for _ in range(3): bwd(p, False)
torch.cuda.synchronize()
