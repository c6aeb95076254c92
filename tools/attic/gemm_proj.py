"""The block's projection GEMMs on their own, at the benchmark shape (B, L, d_model, d_inner) = (8, 8192, 1024, 1024), bf16:
in_proj / out_proj forward + backward through mamba_ssm/ops/projections.py and the x_proj / dt_proj pair as the fused
node runs them (per-batch row-major GEMMs on (d, l) matrices).  For tools/mfma_counters.sh (rocprofv3 --pmc / --stats)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-mamba-suite_amd")); sys.path.insert(0, ROOT)
import torch
from mamba_ssm.ops.projections import in_proj_fn, out_proj_fn

B, L, DM, DI, R, N = 8, 8192, 1024, 1024, 64, 16
dev, bf = "cuda", torch.bfloat16
torch.manual_seed(0)
h = torch.randn(B, L, DM, device=dev, dtype=bf, requires_grad=True)
w_in = torch.randn(2 * DI, DM, device=dev, dtype=bf, requires_grad=True)
w_out = torch.randn(DM, DI, device=dev, dtype=bf, requires_grad=True)
y = torch.randn(B, DI, L, device=dev, dtype=bf, requires_grad=True)
w_x = torch.randn(R + 2 * N, DI, device=dev, dtype=bf)
w_dt = torch.randn(DI, R, device=dev, dtype=bf)
conv_out = torch.randn(B, DI, L, device=dev, dtype=bf)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
for _ in range(n):
    xz = in_proj_fn(h, w_in, None)
    xz.backward(torch.randn_like(xz))
    o = out_proj_fn(y, w_out, None)
    o.backward(torch.randn_like(o))
    x_dblT = torch.matmul(w_x, conv_out)            # (B, R + 2N, L)
    delta = torch.matmul(w_dt, x_dblT[:, :R])       # (B, DI, L)
torch.cuda.synchronize()
