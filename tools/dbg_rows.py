import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-mamba-suite_amd")); sys.path.insert(0, ROOT)
import torch
import selective_scan_cuda
torch.manual_seed(0)
b, d, L, N = 1, 64, 128, 16
dt = torch.float32
dev = "cuda"
u = torch.randn(b, d, L, device=dev, dtype=dt); delta = 0.5 * torch.rand(b, d, L, device=dev, dtype=dt)
A = -0.5 * torch.rand(d, N, device=dev); B = torch.randn(b, 1, N, L, device=dev, dtype=dt); C = torch.randn(b, 1, N, L, device=dev, dtype=dt)
D = torch.randn(d, device=dev); z = torch.randn(b, d, L, device=dev, dtype=dt); bias = 0.5 * torch.rand(d, device=dev)
os.environ["VMS_SCAN_IMPL"] = "rows"
o1, x1, oz1 = selective_scan_cuda.fwd(u, delta, A, B, C, D, z, bias, True)
os.environ["VMS_SCAN_IMPL"] = "generic"
o2, x2, oz2 = selective_scan_cuda.fwd(u, delta, A, B, C, D, z, bias, True)
torch.cuda.synchronize()
print("out  max err", (o1 - o2).abs().max().item(), "ref max", o2.abs().max().item())
print("x    max err", (x1 - x2).abs().max().item())
e = (o1 - o2).abs()
print("err by position (first 16):", e.amax(dim=(0, 1))[:16].tolist())
print("err by row (first 8):", e.amax(dim=(0, 2))[:8].tolist())
print("o1[0,0,:8]", o1[0, 0, :8].tolist()); print("o2[0,0,:8]", o2[0, 0, :8].tolist())
# no-C check: out - D*u
print("out err by row:", [round(v, 3) for v in e.amax(dim=(0, 2)).tolist()])
ex = (x1 - x2).abs()
print("x err by row:", [round(v, 3) for v in ex.amax(dim=(0, 2, 3)).tolist()])
print("x err by slot (row0):", [round(v, 3) for v in ex[0, 0, 0].tolist()])
