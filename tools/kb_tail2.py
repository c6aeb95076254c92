"""Needs profiles/r06_tail_dual.patch applied (vms_hip.proj_conv_bwd_dual): the one-pass two-direction backward tail against the two launches,
results compared and both timed back to back.  usage: python tools/kb_tail2.py [n_shapes]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-mamba-suite_amd"))
import torch, vms_hip
dev, bf = "cuda", torch.bfloat16
def timeit(fn, n=30, w=10):
    for _ in range(w): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
def rel(a, b):
    a, b = a.float(), b.float()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-6)).item()
def one(b, d, L, k, W=4):
    torch.manual_seed(0)
    xz = torch.randn(b, 2 * d, L, device=dev, dtype=bf)
    x = xz[:, :d]
    mk = lambda: (torch.randn(d, b, L, device=dev, dtype=bf).permute(1, 0, 2), (torch.randn(b, k, L, device=dev) * 0.3).to(bf),
                  (torch.randn(k, d, device=dev) * d ** -0.5).to(bf), torch.randn(d, W, device=dev) * 0.4, torch.randn(d, device=dev) * 0.2)
    A, B = mk(), mk()
    def zeros(): return (torch.zeros(d, W, device=dev), torch.zeros(d, device=dev), torch.zeros(k, d, device=dev))
    # reference: two launches
    dxz0 = torch.zeros_like(xz); dx0 = dxz0[:, :d]
    za, zb = zeros(), zeros()
    vms_hip.proj_conv_bwd(x, A[0], A[1], A[2], A[3], A[4], dx0, za[0], za[1], za[2])
    vms_hip.proj_conv_bwd(x, B[0], B[1], B[2], B[3], B[4], dx0, zb[0], zb[1], zb[2], reverse=True, dx_accumulate=True)
    dxz1 = torch.full_like(xz, float("nan")); dx1 = dxz1[:, :d]
    ya, yb = zeros(), zeros()
    fused = vms_hip.proj_conv_bwd_dual(x, dx1, (A[0], A[1], A[2], A[3], A[4], ya[0], ya[1], ya[2]), (B[0], B[1], B[2], B[3], B[4], yb[0], yb[1], yb[2]))
    torch.cuda.synchronize()
    errs = dict(dx=rel(dx1, dx0), dcw_a=rel(ya[0], za[0]), dcb_a=rel(ya[1], za[1]), dwx_a=rel(ya[2], za[2]), dcw_b=rel(yb[0], zb[0]), dcb_b=rel(yb[1], zb[1]), dwx_b=rel(yb[2], zb[2]))
    # where does dx differ most
    diff = (dx1.float() - dx0.float()).abs()
    idx = torch.nonzero(diff == diff.max())[0].tolist()
    t2 = timeit(lambda: (vms_hip.proj_conv_bwd(x, A[0], A[1], A[2], A[3], A[4], dx0, za[0], za[1], za[2]),
                         vms_hip.proj_conv_bwd(x, B[0], B[1], B[2], B[3], B[4], dx0, zb[0], zb[1], zb[2], reverse=True, dx_accumulate=True)))
    t1 = timeit(lambda: vms_hip.proj_conv_bwd_dual(x, dx1, (A[0], A[1], A[2], A[3], A[4], ya[0], ya[1], ya[2]), (B[0], B[1], B[2], B[3], B[4], yb[0], yb[1], yb[2])))
    print(f"({b}, {d}, {L}, k={k}) fused={fused} [{vms_hip.last_kernel()}]  two launches {t2:7.1f} us   one pass {t1:7.1f} us   " + " ".join(f"{n} {e:.1e}" for n, e in errs.items()) + f"  worst dx at {idx}", flush=True)
shapes = [(2, 64, 256, 20), (3, 200, 1032, 40), (8, 1024, 8192, 96), (8, 768, 3136, 80), (1, 768, 65536, 80), (8, 384, 3152, 56)]
if len(sys.argv) > 1: shapes = shapes[:int(sys.argv[1])]
for s in shapes: one(*s)
