import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "video-mamba-suite_amd")); sys.path.insert(0, ROOT)
import torch, vms_hip
import selective_scan_cuda as ssc
DEV="cuda"
def rel(a, b):
    a=a.float(); b=b.float()
    return ((a-b).abs().max()/b.abs().max().clamp_min(1e-6)).item()
for N in (4, 8, 16):
  for layout in ("3", "1"):
    os.environ["VMS_X_LAYOUT"] = layout
    b, d, L = 2, 64, 1040
    torch.manual_seed(N)
    dt = torch.bfloat16
    u = torch.randn(b, d, L, device=DEV).to(dt)
    delta = (0.5 * torch.rand(b, d, L, device=DEV)).to(dt)
    A = -0.5 * torch.rand(d, N, device=DEV)
    B = torch.randn(b, 1, N, L, device=DEV).to(dt)
    C = torch.randn(b, 1, N, L, device=DEV).to(dt)
    D = torch.randn(d, device=DEV)
    z = torch.randn(b, d, L, device=DEV).to(dt)
    bias = 0.5 * torch.rand(d, device=DEV)
    dout = torch.randn(b, d, L, device=DEV).to(dt)
    res = {}
    for impl in ("pair", "generic"):
        vms_hip.debug.scan_impl = impl
        out, x, out_z = ssc.fwd(u, delta, A, B, C, D, z, bias, True)
        kf = vms_hip.last_kernel()
        grads = ssc.bwd(u, delta, A, B, C, D, z, bias, dout, x, out, None, True, False, keep_fp32=True)
        kb = vms_hip.last_kernel()
        res[impl] = (out, out_z, x, grads, kf, kb)
    torch.cuda.synchronize()
    p, g = res["pair"], res["generic"]
    print(f"N={N} layout={layout} {p[4]} {p[5]} | out {rel(p[0], g[0]):.2e} out_z {rel(p[1], g[1]):.2e} last {rel(p[2][:, :, -1, 1::2], g[2][:, :, -1, 1::2]):.2e} xshape {tuple(p[2].shape)} {p[2].stride()}")
    print("   ", " ".join(f"{n} {rel(p[3][k], g[3][k]):.2e}" for k, n in enumerate(["du", "ddelta", "dA", "dB", "dC", "dD", "dbias", "dz"])))
    # backward of the pair kernels fed by the GENERIC forward's x (coarse layout only makes sense when layouts agree)
