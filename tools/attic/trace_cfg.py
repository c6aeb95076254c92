"""13 block steps at a given (B, L, d_model) for a rocprofv3 kernel trace: python tools/trace_cfg.py B L d_model"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "video-mamba-suite_amd"))
from mamba_ssm.modules.mamba_simple import Mamba as ViM
B, L, dm = [int(v) for v in sys.argv[1:4]]
block = ViM(dm, expand=1, bimamba_type="v2").cuda()
x = torch.randn(B, L, dm, device="cuda", dtype=torch.bfloat16, requires_grad=True)
g = torch.randn(B, L, dm, device="cuda", dtype=torch.bfloat16)
for _ in range(13):
    block.zero_grad(set_to_none=True); x.grad = None
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = block(x)
    y.backward(g)
torch.cuda.synchronize()
