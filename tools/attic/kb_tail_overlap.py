"""The fused backward tail of one direction (vms_proj_conv_bwd: 2 waves per SIMD, 54 KB LDS, latency-bound) beside the OTHER
direction's VALU-bound backward scan (2 waves per SIMD x 112 VGPRs, 57 KB LDS, one workgroup per CU) on two streams: do they
share the CUs, and what does the pair cost?  (8, 8192, 1024, 16) bf16.  usage: python tools/kb_tail_overlap.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-mamba-suite_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import vms_hip
from kb_dual import problem, bwd


def main():
    p = problem(0)
    b, d, L, R, N = 8, 1024, 8192, 64, 16
    dev, dt = "cuda", torch.bfloat16
    K2 = R + 2 * N
    xz = torch.randn(b, 2 * d, L, device=dev, dtype=dt); dxz = torch.empty_like(xz)
    du = torch.randn(b, d, L, device=dev, dtype=dt); dx_dbl = torch.randn(b, K2, L, device=dev, dtype=dt)
    w_x = (torch.randn(K2, d, device=dev) * 0.03).to(dt)
    cw, cb = torch.randn(d, 4, device=dev) * 0.3, torch.randn(d, device=dev) * 0.1
    zw, zb, zx = torch.zeros(d, 4, device=dev), torch.zeros(d, device=dev), torch.zeros(K2, d, device=dev)
    ddelta = torch.randn(b, d, L, device=dev, dtype=dt); x_dbl = torch.randn(b, K2, L, device=dev, dtype=dt)
    Wdt = (torch.randn(d, R, device=dev) * 0.1).to(dt)

    def tail(acc=False):
        vms_hip.proj_conv_bwd(xz[:, :d], du, dx_dbl, w_x, cw, cb, dxz[:, :d], zw, zb, zx, dx_accumulate=acc)

    def small_gemms():
        g1 = torch.matmul(ddelta, x_dbl[:, :R].transpose(1, 2)).sum(0)
        torch.bmm(Wdt.t().unsqueeze(0).expand(b, -1, -1), ddelta, out=dx_dbl[:, :R])
        return g1

    def timeit(fn, n=20):
        for _ in range(5): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3

    s1 = torch.cuda.Stream()

    def two_streams(side, first):
        cur = torch.cuda.current_stream()
        s1.wait_stream(cur)
        if first == "side":
            with torch.cuda.stream(s1):
                side()
            bwd(p, False)
        else:
            bwd(p, False)
            with torch.cuda.stream(s1):
                side()
        cur.wait_stream(s1)

    a = torch.randn(4096, 4096, device=dev, dtype=dt)
    for _ in range(20): a @ a
    t_scan, t_tail, t_gemm = timeit(lambda: bwd(p, False)), timeit(tail), timeit(small_gemms)
    print(f"alone: scan_bwd {t_scan:7.1f} us   fused tail {t_tail:7.1f} us   d_dt + dW_dt GEMMs {t_gemm:7.1f} us")
    for name, side, t_side in (("fused tail", tail, t_tail), ("small GEMMs + fused tail", lambda: (small_gemms(), tail()), t_tail + t_gemm)):
        ser = timeit(lambda: (bwd(p, False), side()))
        for first in ("side", "scan"):
            t = timeit(lambda: two_streams(side, first))
            print(f"scan_bwd || {name:26s} launched {'before' if first == 'side' else 'after '} the scan: {t:7.1f} us   serial {ser:7.1f} us   ratio {t / ser:.3f}", flush=True)


if __name__ == "__main__":
    main()
