"""Time vms_param_prep on the (8, 8192, 1024) ViM block's eight jobs against the kernels it replaces (torch copies)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "video-mamba-suite_amd"))
import torch
import vms_hip
dev = "cuda"
mk = lambda *s: torch.randn(*s, device=dev)
ws = [mk(2048, 1024), mk(96, 1024), mk(1024, 64), mk(96, 1024), mk(1024, 64), mk(1024, 1024)]
al = [mk(1024, 16), mk(1024, 16)]
lows = [torch.empty(1024, 2048, device=dev, dtype=torch.bfloat16)] + [torch.empty_like(w, dtype=torch.bfloat16) for w in ws[1:]]
A2 = torch.empty(2, 1024, 16, device=dev)
jobs = [(ws[0], lows[0], vms_hip.PREP_CAST_T)] + [(w, l, vms_hip.PREP_CAST) for w, l in zip(ws[1:], lows[1:])] + \
       [(al[0], A2[0], vms_hip.PREP_NEG_EXP), (al[1], A2[1], vms_hip.PREP_NEG_EXP)]
def t(f, n=200):
    for _ in range(20): f()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1000
def old():
    lows[0].copy_(ws[0].t())
    torch._foreach_copy_(lows[1:5], ws[1:5])
    lows[5].copy_(ws[5]); lows[5].copy_(ws[5])
    o = torch._foreach_exp(al); torch._foreach_neg_(o)
print("vms_param_prep (one launch)      %.1f us" % t(lambda: vms_hip.param_prep(jobs)))
print("the copies it replaces (7 launches, back to back) %.1f us" % t(old))
