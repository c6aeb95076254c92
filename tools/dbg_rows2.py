import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-mamba-suite_amd")); sys.path.insert(0, ROOT)
import torch
import selective_scan_cuda
torch.manual_seed(0)
b, d, L, N = 2, 64, 128, 16
dt = torch.bfloat16
dev = "cuda"
u = torch.randn(b, d, L, device=dev).to(dt); delta = (0.5 * torch.rand(b, d, L, device=dev)).to(dt)
A = -0.5 * torch.rand(d, N, device=dev); B = torch.randn(b, 1, N, L, device=dev).to(dt); C = torch.randn(b, 1, N, L, device=dev).to(dt)
D = torch.randn(d, device=dev); bias = 0.5 * torch.rand(d, device=dev)
for hz in (False, True):
    z = torch.randn(b, d, L, device=dev).to(dt) if hz else None
    os.environ["VMS_SCAN_IMPL"] = "rows"
    r1 = selective_scan_cuda.fwd(u, delta, A, B, C, D, z, bias, True)
    os.environ["VMS_SCAN_IMPL"] = "generic"
    r2 = selective_scan_cuda.fwd(u, delta, A, B, C, D, z, bias, True)
    torch.cuda.synchronize()
    o1, o2 = r1[0].float(), r2[0].float()
    nan = torch.isnan(o1)
    print("hz", hz, "nans", int(nan.sum()), "of", o1.numel(), "max err (non-nan)", (o1 - o2)[~nan].abs().max().item())
    if nan.any():
        idx = nan.nonzero()
        print(" first nan idx", idx[:5].tolist(), "nan by batch", nan.sum(dim=(1, 2)).tolist())
        print(" nan by position", nan.sum(dim=(0, 1)).tolist())
        print(" nan by row(b0)", nan[0].sum(dim=1).tolist())
    print(" x err", (r1[1] - r2[1]).abs().max().item())
