"""Does the memory-bound tail of one direction's backward (small projection GEMMs + conv backward) overlap with the other
direction's VALU-bound backward scan when they run on two streams?  (8, 8192, 1024, 16) bf16."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-mamba-suite_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import causal_conv1d_cuda
from kb_dual import problem, bwd

def main():
    p = problem(0)
    b, d, L, R, N = 8, 1024, 8192, 64, 16
    dev, dt = "cuda", torch.bfloat16
    ddelta = torch.randn(b, d, L, device=dev, dtype=dt); x_dbl = torch.randn(b, R + 2 * N, L, device=dev, dtype=dt)
    conv_out = torch.randn(b, d, L, device=dev, dtype=dt); dconv = torch.randn(b, d, L, device=dev, dtype=dt)
    Wdt = torch.randn(d, R, device=dev, dtype=dt); Wx = torch.randn(R + 2 * N, d, device=dev, dtype=dt)
    dx_dbl = torch.empty_like(x_dbl)
    xz = torch.randn(b, 2 * d, L, device=dev, dtype=dt); dxz = torch.empty_like(xz)
    w = torch.randn(d, 4, device=dev); cb = torch.randn(d, device=dev)
    def tail():   # what follows a direction's scan backward in _inner_backward
        g1 = torch.matmul(ddelta, x_dbl[:, :R].transpose(1, 2)).sum(0)
        dx_dbl[:, :R] = torch.matmul(Wdt.t(), ddelta)
        g2 = torch.matmul(dx_dbl, conv_out.transpose(1, 2)).sum(0)
        dconv.baddbmm_(Wx.t().expand(b, -1, -1), dx_dbl)
        causal_conv1d_cuda.causal_conv1d_bwd(xz[:, :d], w, cb, dconv, dxz[:, :d], True)
        return g1, g2
    def timeit(fn, n=20):
        for _ in range(5): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3
    s1 = torch.cuda.Stream()
    def serial():
        bwd(p, False); tail()
    def overlapped():
        cur = torch.cuda.current_stream()
        s1.wait_stream(cur)
        with torch.cuda.stream(s1):
            tail()
        bwd(p, False)
        cur.wait_stream(s1)
    print(f"scan_bwd alone {timeit(lambda: bwd(p, False)):8.1f} us   tail alone {timeit(tail):8.1f} us   serial {timeit(serial):8.1f} us   two streams {timeit(overlapped):8.1f} us")

if __name__ == "__main__":
    main()
