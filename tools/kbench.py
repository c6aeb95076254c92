"""Kernel-level timing of the four hot kernels at BASELINE config 2 (8, 8192, 1024, 16) bf16.
usage: python tools/kbench.py [fwd] [bwd] [conv]   (env VMS_DEBUG passes profiling knobs)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-mamba-suite_amd")); sys.path.insert(0, ROOT)
import torch
import selective_scan_cuda, causal_conv1d_cuda
from bench import algorithmic_bytes

def timeit(fn, n=20, warm=3):
    # clocks ramp over the first tens of milliseconds of load: run >= 60 ms before timing (a cold 13-launch sample reads 10-15 % slow)
    t0 = time.time()
    while True:
        for _ in range(warm): fn()
        torch.cuda.synchronize()
        if time.time() - t0 > 0.06: break
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

def main():
    which = [a for a in sys.argv[1:] if a != "norm"] or ["fwd", "bwd", "conv"]
    dt = torch.bfloat16 if os.environ.get("KB_DTYPE", "bf16") == "bf16" else torch.float32
    b, d, L, N = [int(x) for x in os.environ.get("KB_SHAPE", "8,1024,8192,16").split(",")]
    dev = "cuda"
    torch.manual_seed(0)
    xz = torch.randn(b, 2 * d, L, device=dev, dtype=dt)
    u, z = xz[:, :d], xz[:, d:]
    delta = (0.5 * torch.rand(d, b, L, device=dev)).to(dt).permute(1, 0, 2)
    A = -torch.arange(1, N + 1, device=dev, dtype=torch.float32).repeat(d, 1).contiguous()
    B = torch.randn(b, 1, N, L, device=dev, dtype=dt); C = torch.randn(b, 1, N, L, device=dev, dtype=dt)
    D = torch.ones(d, device=dev); bias = torch.randn(d, device=dev) - 4.0
    ab = algorithmic_bytes(b, d, L, N, 2 if dt == torch.bfloat16 else 4)   # backward without the out_z recompute (the blocks' nodes)
    out, x, out_z = selective_scan_cuda.fwd(u, delta, A, B, C, D, z, bias, True)
    if "fwd" in which:
        t = timeit(lambda: selective_scan_cuda.fwd(u, delta, A, B, C, D, z, bias, True))
        print(f"scan_fwd  {t*1e3:9.1f} us  {ab['vms_selective_scan_fwd']/t/1e6:8.1f} GB/s  {ab['vms_selective_scan_fwd']/t/1e6/8000*100:5.1f}% of 8 TB/s")
    if "bwd" in which:
        dout = torch.randn(b, d, L, device=dev, dtype=dt)
        dxz = torch.empty_like(xz); dz = dxz[:, d:]
        t = timeit(lambda: selective_scan_cuda.bwd(u, delta, A, B, C, D, z, bias, dout, x, out, dz, True, False))
        print(f"scan_bwd  {t*1e3:9.1f} us  {ab['vms_selective_scan_bwd']/t/1e6:8.1f} GB/s  {ab['vms_selective_scan_bwd']/t/1e6/8000*100:5.1f}% of 8 TB/s")
    if "dual" in which:   # both directions' backward scans as one call (vms_selective_scan_bwd_dual)
        dout = torch.randn(b, d, L, device=dev, dtype=dt)
        dxz = torch.empty_like(xz); dz = dxz[:, d:]
        A2 = A * 1.1
        out2, x2, _ = selective_scan_cuda.fwd(u, delta, A2, B, C, D, z, bias, True, reverse=True)
        da, db_ = (u, delta, A, B, C, D, bias, x, out), (u, delta, A2, B, C, D, bias, x2, out2)
        t = timeit(lambda: selective_scan_cuda.bwd_dual(da, db_, z, dout, dz, True, keep_fp32=True))
        import vms_hip
        print(f"scan_bwd_dual {t*1e3:9.1f} us  {ab['vms_selective_scan_bwd_dual']/t/1e6:8.1f} GB/s  {ab['vms_selective_scan_bwd_dual']/t/1e6/8000*100:5.1f}% of 8 TB/s  [{vms_hip.lib().vms_last_kernel().decode()}]")
    if "conv" in which:
        w = torch.randn(d, 4, device=dev); cb = torch.randn(d, device=dev)
        t = timeit(lambda: causal_conv1d_cuda.causal_conv1d_fwd(u, w, cb, True))
        print(f"conv_fwd  {t*1e3:9.1f} us  {ab['vms_causal_conv1d_fwd']/t/1e6:8.1f} GB/s  {ab['vms_causal_conv1d_fwd']/t/1e6/8000*100:5.1f}% of 8 TB/s")
        dout = torch.randn(b, d, L, device=dev, dtype=dt); dx = torch.empty_like(xz)[:, :d]
        t = timeit(lambda: causal_conv1d_cuda.causal_conv1d_bwd(u, w, cb, dout, dx, True))
        print(f"conv_bwd  {t*1e3:9.1f} us  {ab['vms_causal_conv1d_bwd']/t/1e6:8.1f} GB/s  {ab['vms_causal_conv1d_bwd']/t/1e6/8000*100:5.1f}% of 8 TB/s")

def norm_bench():
    import layer_norm_cuda
    M, N = [int(v) for v in os.environ.get("KB_NORM_SHAPE", "65536,1024").split(",")]   # rows, columns
    x = torch.randn(M, N, device="cuda", dtype=torch.bfloat16); res = torch.randn(M, N, device="cuda", dtype=torch.float32)
    w = torch.ones(N, device="cuda"); dy = torch.randn_like(x); dres = torch.randn_like(res)
    y, mean, rstd, ro = layer_norm_cuda.fwd(x, w, None, 1e-5, res, is_rms_norm=True)
    t = timeit(lambda: layer_norm_cuda.fwd(x, w, None, 1e-5, res, is_rms_norm=True))
    by = M * N * (2 + 4 + 2 + 4)
    print(f"rmsnorm_fwd (bf16 x + fp32 residual -> bf16 y + fp32 sum) {t*1e3:7.1f} us  {by/t/1e6:8.1f} GB/s  {by/t/1e6/8000*100:5.1f}% of 8 TB/s")
    t = timeit(lambda: layer_norm_cuda.bwd(dy, ro, w, None, 1e-5, mean, rstd, dres, True, True, x_dtype=torch.bfloat16))
    by = M * N * (4 + 2 + 4 + 2 + 4)
    print(f"rmsnorm_bwd (fp32 sum, bf16 dy, fp32 dres -> bf16 dx, fp32 dres_in) {t*1e3:7.1f} us  {by/t/1e6:8.1f} GB/s  {by/t/1e6/8000*100:5.1f}% of 8 TB/s")
    x2 = torch.randn(M, N, device="cuda", dtype=torch.bfloat16)
    t = timeit(lambda: layer_norm_cuda.fwd(x2, w, w, 1e-5, None, is_rms_norm=False))
    by = M * N * 4
    print(f"layernorm_fwd (bf16 -> bf16) {t*1e3:7.1f} us  {by/t/1e6:8.1f} GB/s  {by/t/1e6/8000*100:5.1f}% of 8 TB/s")


if __name__ == "__main__":
    args = sys.argv[1:]
    if not args or any(a in ("fwd", "bwd", "conv", "dual") for a in args):
        main()
    if "norm" in args:
        norm_bench()
