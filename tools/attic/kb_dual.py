"""Two independent backward (and forward) scans -- the two directions of a ViM block -- back to back on one stream
vs concurrently on two streams: how much of a second launch fits beside the first (2 waves/SIMD per launch).
usage: python tools/kb_dual.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-mamba-suite_amd")); sys.path.insert(0, ROOT)
import torch
import selective_scan_cuda

def problem(seed, b=8, d=1024, L=8192, N=16, dt=torch.bfloat16, dev="cuda"):
    torch.manual_seed(seed)
    xz = torch.randn(b, 2 * d, L, device=dev, dtype=dt)
    u, z = xz[:, :d], xz[:, d:]
    delta = (0.5 * torch.rand(d, b, L, device=dev)).to(dt).permute(1, 0, 2)
    A = -torch.arange(1, N + 1, device=dev, dtype=torch.float32).repeat(d, 1).contiguous()
    B = torch.randn(b, 1, N, L, device=dev, dtype=dt); C = torch.randn(b, 1, N, L, device=dev, dtype=dt)
    D = torch.ones(d, device=dev); bias = torch.randn(d, device=dev) - 4.0
    out, x, out_z = selective_scan_cuda.fwd(u, delta, A, B, C, D, z, bias, True)
    dout = torch.randn(b, d, L, device=dev, dtype=dt)
    dxz = torch.empty_like(xz); dz = dxz[:, d:]
    return dict(u=u, delta=delta, A=A, B=B, C=C, D=D, z=z, bias=bias, dout=dout, x=x, out=out, dz=dz)

def bwd(p, rev):
    return selective_scan_cuda.bwd(p["u"], p["delta"], p["A"], p["B"], p["C"], p["D"], p["z"], p["bias"], p["dout"], p["x"], p["out"],
                                   p["dz"], True, True, reverse=rev)
def fwd(p, rev):
    return selective_scan_cuda.fwd(p["u"], p["delta"], p["A"], p["B"], p["C"], p["D"], p["z"], p["bias"], True, reverse=rev)

def timeit(fn, n=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

def main():
    p0, p1 = problem(0), problem(1)
    s1 = torch.cuda.Stream()
    for name, op in (("bwd", bwd), ("fwd", fwd)):
        def serial():
            op(p0, False); op(p1, True)
        def dual():
            cur = torch.cuda.current_stream()
            s1.wait_stream(cur)
            op(p0, False)
            with torch.cuda.stream(s1):
                op(p1, True)
            cur.wait_stream(s1)
        print(f"{name}: one launch {timeit(lambda: op(p0, False)):8.1f} us   two serial {timeit(serial):8.1f} us   two streams {timeit(dual):8.1f} us")

if __name__ == "__main__":
    main()
