"""The block's four large dense GEMMs at (8, 8192, 1024) with the WEIGHT operand as the parameter stores it vs as a transposed copy
(the activations' layouts are dictated by the kernels): python tools/gemm_weight_layout_ab.py"""
import sys
import time
import torch

dev, dt = "cuda", torch.bfloat16
BL, dm, ch = [int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (65536, 1024, 2048))]   # rows, d_model, in_proj channels (2 d_inner)
di = ch // 2
hidden = torch.randn(BL, dm, device=dev, dtype=dt)
W = torch.randn(ch, dm, device=dev, dtype=dt); Wt = W.t().contiguous()
Wo = torch.randn(dm, di, device=dev, dtype=dt); Wot = Wo.t().contiguous()   # out_proj.weight (d_model, d_inner)
y = torch.randn(di, BL, device=dev, dtype=dt)             # (channels, B L)
g = torch.randn(ch, BL, device=dev, dtype=dt)
dout = torch.randn(BL, dm, device=dev, dtype=dt)

forms = [
    ("in_proj fwd   (ch, BL) = W hidden^T         weight as stored", lambda: torch.mm(W, hidden.t())),
    ("in_proj fwd                                  transposed copy ", lambda: torch.mm(Wt.t(), hidden.t())),
    ("in_proj dgrad (BL, dm) = g^T W              weight as stored", lambda: torch.mm(g.t(), W)),
    ("in_proj dgrad                                transposed copy ", lambda: torch.mm(g.t(), Wt.t())),
    ("out_proj fwd  (BL, dm) = y^T Wo^T           weight as stored", lambda: torch.mm(y.t(), Wo.t())),
    ("out_proj fwd                                 transposed copy ", lambda: torch.mm(y.t(), Wot)),
    ("out_proj dgrad (dm, BL) = Wo^T dout^T       weight as stored", lambda: torch.mm(Wo.t(), dout.t())),
    ("out_proj dgrad                               transposed copy ", lambda: torch.mm(Wot, dout.t())),
]


def timeit(fn, n=60):
    t0 = time.time()
    while time.time() - t0 < 0.15:
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


print(f'rows {BL}, d_model {dm}, in_proj channels {ch}')
for rep in range(2):
    for k, f in forms:
        t = timeit(f)
        fl = 2 * BL * dm * (ch if "in_proj" in k else di)
        print(f"{k}   {t:7.1f} us   {fl / t / 1e9:6.3f} PFLOP/s", flush=True)
    print()
