"""A few launches of the fused backward tail (vms_proj_conv_bwd) at the block shape, for counter passes (tools/pmc.sh).
usage: python tools/kb_tail_once.py [acc]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-mamba-suite_amd"))
import torch
import vms_hip

b, d, L, R, N = (int(v) for v in os.environ.get("KB_SHAPE", "8,1024,8192,64,16").split(","))
dev, dt, K2 = "cuda", torch.bfloat16, R + 2 * N
torch.manual_seed(0)
xz = torch.randn(b, 2 * d, L, device=dev, dtype=dt); dxz = torch.randn(b, 2 * d, L, device=dev, dtype=dt)
du = torch.randn(b, d, L, device=dev, dtype=dt); dx_dbl = torch.randn(b, K2, L, device=dev, dtype=dt)
w_x = (torch.randn(K2, d, device=dev) * 0.03).to(dt)
cw, cb = torch.randn(d, 4, device=dev) * 0.3, torch.randn(d, device=dev) * 0.1
zw, zb, zx = torch.zeros(d, 4, device=dev), torch.zeros(d, device=dev), torch.zeros(K2, d, device=dev)
for _ in range(6):
    vms_hip.proj_conv_bwd(xz[:, :d], du, dx_dbl, w_x, cw, cb, dxz[:, :d], zw, zb, zx, dx_accumulate="acc" in sys.argv)
torch.cuda.synchronize()
print("done")
