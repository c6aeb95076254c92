"""the lane-per-row scan kernels at a TimeMamba shape (batch x 196 rows per channel, a few frames long), in the block's layouts.
usage: KB_SHAPE=b,d,L python tools/kb_short.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-mamba-suite_amd")); sys.path.insert(0, ROOT)
import torch
import selective_scan_cuda, vms_hip

def timeit(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

b, d, L = [int(v) for v in os.environ.get("KB_SHAPE", "1568,768,16").split(",")]
dt, N, dev = torch.bfloat16, 16, "cuda"
torch.manual_seed(0)
xz = torch.randn(2 * d, b, L, device=dev, dtype=dt).permute(1, 0, 2)           # channel-slowest, as the block produces it
u, z = xz[:, :d], xz[:, d:]
delta = (0.5 * torch.rand(d, b, L, device=dev)).to(dt).permute(1, 0, 2)
A = -torch.arange(1, N + 1, device=dev, dtype=torch.float32).repeat(d, 1).contiguous()
B = torch.randn(b, 1, N, L, device=dev, dtype=dt); C = torch.randn(b, 1, N, L, device=dev, dtype=dt)
D = torch.ones(d, device=dev); bias = torch.randn(d, device=dev) - 4.0
dout = torch.randn(d, b, L, device=dev, dtype=dt).permute(1, 0, 2)
out, x, out_z = selective_scan_cuda.fwd(u, delta, A, B, C, D, z, bias, True)
t = timeit(lambda: selective_scan_cuda.fwd(u, delta, A, B, C, D, z, bias, True))
print(f"scan_fwd ({b}, {d}, {L})  {t:8.1f} us  [{vms_hip.last_kernel()}]")
dxz = torch.empty_like(xz); dz = dxz[:, d:]
t = timeit(lambda: selective_scan_cuda.bwd(u, delta, A, B, C, D, z, bias, dout, x, out, dz, True, False))
print(f"scan_bwd ({b}, {d}, {L})  {t:8.1f} us  [{vms_hip.last_kernel()}]")
