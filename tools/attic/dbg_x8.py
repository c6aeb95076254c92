import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-mamba-suite_amd"))
import torch, selective_scan_cuda
torch.manual_seed(0)
b, d, N, L = 2, 64, 16, int(sys.argv[1]) if len(sys.argv) > 1 else 1024
u = torch.randn(b, d, L, device="cuda"); delta = 0.5 * torch.rand(b, d, L, device="cuda"); A = -0.5 * torch.rand(d, N, device="cuda")
B = torch.randn(b, 1, N, L, device="cuda"); C = torch.randn(b, 1, N, L, device="cuda"); D = torch.randn(d, device="cuda"); z = torch.randn(b, d, L, device="cuda")
bias = 0.5 * torch.rand(d, device="cuda")
def run(dt):
    c = lambda t: t.to(dt)
    out, x, oz = selective_scan_cuda.fwd(c(u), c(delta), A, c(B), c(C), D, c(z), bias, True)
    nch = x.shape[2]
    return x.as_strided((b, d, nch, 258 * N), (d * nch * 258 * N, nch * 258 * N, 258 * N, 1))[..., 2 * N:].reshape(b, d, nch, 4, 256, 4).clone()
x32 = run(torch.float32)
for dt in (torch.bfloat16, torch.float16):
    xx = run(dt)
    err = (xx - x32).abs() / (x32.abs().amax() + 1e-9)
    n8 = L // 8
    for c in range(x32.shape[2]):
        for g in range(4):
            e = err[:, :, c, g, :min(256, n8 - c * 256)]
            bad = (e > 0.05).nonzero()
            print(dt, "chunk", c, "group", g, "max err %.3g" % e.max().item(), "bad", bad.shape[0], bad[:4].tolist())
