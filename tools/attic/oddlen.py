"""Kernel times for sequence lengths with and without the fast paths' alignment (L % 8 == 0, 16-byte rows)."""
import sys, torch
sys.path.insert(0, "video-mamba-suite_amd")
import selective_scan_cuda, causal_conv1d_cuda

def timeit(fn, n=10, w=3):
    for _ in range(w): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

for (b, d, L) in [(8, 768, 1568), (8, 768, 1569), (8, 768, 3136), (8, 768, 3137)]:
    dev, dt = "cuda", torch.bfloat16
    u = torch.randn(b, d, L, device=dev).to(dt); delta = (0.5 * torch.rand(b, d, L, device=dev)).to(dt)
    z = torch.randn(b, d, L, device=dev).to(dt); dout = torch.randn(b, d, L, device=dev).to(dt)
    A = -torch.rand(d, 16, device=dev); B = torch.randn(b, 1, 16, L, device=dev).to(dt); C = torch.randn(b, 1, 16, L, device=dev).to(dt)
    D = torch.randn(d, device=dev); bias = torch.rand(d, device=dev)
    out, x, oz = selective_scan_cuda.fwd(u, delta, A, B, C, D, z, bias, True)
    tf = timeit(lambda: selective_scan_cuda.fwd(u, delta, A, B, C, D, z, bias, True))
    tb = timeit(lambda: selective_scan_cuda.bwd(u, delta, A, B, C, D, z, bias, dout, x, out, None, True, True))
    w = torch.randn(d, 4, device=dev); cb = torch.randn(d, device=dev)
    tcf = timeit(lambda: causal_conv1d_cuda.causal_conv1d_fwd(u, w, cb, True))
    tcb = timeit(lambda: causal_conv1d_cuda.causal_conv1d_bwd(u, w, cb, dout, None, True))
    print(f"(B, D, L) = ({b}, {d}, {L}): scan fwd {tf:8.1f} us  scan bwd {tb:8.1f} us  conv fwd {tcf:7.1f} us  conv bwd {tcb:7.1f} us", flush=True)
