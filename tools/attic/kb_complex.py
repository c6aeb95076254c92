"""Time of the complex-A scan kernels (csrc/selective_scan_complex.hip; built for completeness, not tuned) next to the real ones."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "video-mamba-suite_amd"))
import torch
import selective_scan_cuda as ssc
dev, dt = "cuda", torch.bfloat16
def t(f, n=5):
    for _ in range(2): f()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n
for (b, d, L, n) in ((8, 1024, 8192, 8), (8, 768, 3136, 8)):
    u = torch.randn(b, d, L, device=dev, dtype=dt); delta = (0.5 * torch.rand(b, d, L, device=dev)).to(dt)
    z = torch.randn_like(u); dout = torch.randn_like(u)
    D, bias = torch.randn(d, device=dev), torch.rand(d, device=dev)
    Ac = torch.complex(-torch.rand(d, n, device=dev), torch.randn(d, n, device=dev))
    Bc = torch.randn(b, 1, n, 2 * L, device=dev, dtype=dt); Cc = torch.randn_like(Bc)
    out, x, oz = ssc.fwd(u, delta, Ac, Bc, Cc, D, z, bias, True)
    tf = t(lambda: ssc.fwd(u, delta, Ac, Bc, Cc, D, z, bias, True))
    tb = t(lambda: ssc.bwd(u, delta, Ac, Bc, Cc, D, z, bias, dout, x, out, None, True, False))
    Ar = -torch.rand(d, 2 * n, device=dev); Br = torch.randn(b, 1, 2 * n, L, device=dev, dtype=dt); Cr = torch.randn_like(Br)
    outr, xr, ozr = ssc.fwd(u, delta, Ar, Br, Cr, D, z, bias, True)
    tfr = t(lambda: ssc.fwd(u, delta, Ar, Br, Cr, D, z, bias, True))
    tbr = t(lambda: ssc.bwd(u, delta, Ar, Br, Cr, D, z, bias, dout, xr, outr, None, True, False))
    print(f"(B, D, L) = ({b}, {d}, {L}): complex A, {n} complex states: fwd {tf:7.3f} ms  bwd {tb:7.3f} ms | real A, {2 * n} states: fwd {tfr:6.3f} ms  bwd {tbr:6.3f} ms")
