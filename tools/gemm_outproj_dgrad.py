"""out_proj's input gradient dy (B, C, L) = W^T dout^T: per-batch batched GEMM (what OutProjFn runs) vs ONE GEMM over the flattened
rows whose (C, B L) result is the channel-slowest (B, C, L) view the scan backward reads anyway; same for the forward."""
import torch
from gemm_wgrad import timeit

dev = "cuda"
for b, L, C, dm in ((8, 3136, 768, 768), (8, 8192, 1024, 1024), (1, 65536, 768, 768), (8, 3152, 768, 384)):
    W = (torch.randn(dm, C, device=dev) * 0.02).to(torch.bfloat16)
    dout = torch.randn(b, L, dm, device=dev, dtype=torch.bfloat16)
    y_cs = torch.randn(C, b, L, device=dev, dtype=torch.bfloat16).permute(1, 0, 2)        # channel-slowest (B, C, L)
    fl = 2 * b * L * C * dm
    t1 = timeit(lambda: torch.matmul(W.t(), dout.transpose(1, 2)))
    t2 = timeit(lambda: torch.matmul(W.t(), dout.reshape(b * L, dm).t()))
    t3 = timeit(lambda: torch.nn.functional.linear(y_cs.transpose(1, 2), W))
    y2 = y_cs.permute(1, 0, 2).reshape(C, b * L)
    t4 = timeit(lambda: (y2.t() @ W.t()))
    print(f"({b}, {L}, C {C}, d_model {dm}): dgrad batched {t1:6.1f} us {fl / t1 / 1e9:5.2f} PF | flattened {t2:6.1f} us {fl / t2 / 1e9:5.2f} PF"
          f" || fwd F.linear(y^T) {t3:6.1f} us | flattened (B L, C) @ (C, d) {t4:6.1f} us", flush=True)
