"""The block's six large GEMMs in the layouts the product runs them in, under torch's two BLAS backends
(preferred_blas_library: hipblaslt = default for bf16 on ROCm, cublas = rocBLAS)."""
import torch
B, L, Dm, C = 8, 8192, 1024, 2048
dev, dt = "cuda", torch.bfloat16
torch.manual_seed(0)
hidden = torch.randn(B * L, Dm, device=dev, dtype=dt)
wt = torch.randn(Dm, C, device=dev, dtype=dt)                 # in_proj weight, K-contiguous copy (d_model, channels)
g2 = torch.randn(C, B * L, device=dev, dtype=dt)              # dxz as (channels, rows)
y = torch.randn(B, Dm, L, device=dev, dtype=dt)               # scan output (B, d_inner, L)
wo = torch.randn(Dm, Dm, device=dev, dtype=dt)                # out_proj weight (d_model, d_inner)
dout = torch.randn(B, L, Dm, device=dev, dtype=dt)

def t(f, n=20):
    for _ in range(5): f()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1000

ops = {
    "in_proj fwd   wt^T @ x^T": lambda: wt.t() @ hidden.t(),
    "in_proj dgrad g2^T @ wt^T": lambda: g2.t() @ wt.t(),
    "in_proj wgrad bmm(8 K-slices)": lambda: torch.bmm(g2.view(C, 8, L).permute(1, 0, 2), hidden.view(8, L, Dm)),
    "out_proj fwd  linear(y^T, w)": lambda: torch.nn.functional.linear(y.transpose(1, 2), wo),
    "out_proj dgrad w^T @ dout^T": lambda: torch.matmul(wo.t(), dout.transpose(1, 2)),
    "out_proj wgrad bmm(dout^T, y^T)": lambda: torch.bmm(dout.transpose(1, 2), y.transpose(1, 2)),
}
for _ in range(30): ops["in_proj fwd   wt^T @ x^T"]()   # clocks
for lib in ("hipblaslt", "cublas", "hipblaslt", "cublas"):
    torch.backends.cuda.preferred_blas_library(lib)
    print(lib, "  ".join(f"{k.split()[0]} {k.split()[1]} {t(f):6.1f}" for k, f in ops.items()), flush=True)
