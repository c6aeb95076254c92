"""Timing of formulations of the out_proj GEMMs of the bench block (y kept as (B, d, L), library GEMMs only)."""
import torch
from gemm_wgrad import timeit

def main():
    dev = "cuda"
    B, L, dm, d = 8, 8192, 1024, 1024
    yt = torch.randn(B, d, L, device=dev, dtype=torch.bfloat16)
    W = torch.randn(dm, d, device=dev, dtype=torch.bfloat16) * 0.02
    dout = torch.randn(B, L, dm, device=dev, dtype=torch.bfloat16)
    fl = 2 * B * L * dm * d
    def rep(name, fn):
        t = timeit(fn)
        print(f"{name:64s} {t:8.1f} us  {fl / t / 1e9:7.2f} PFLOP/s", flush=True)
    rep("fwd   F.linear(yt.transpose(1,2), W)", lambda: torch.nn.functional.linear(yt.transpose(1, 2), W))
    rep("fwd   bmm(yt^T, W^T)", lambda: torch.bmm(yt.transpose(1, 2), W.t().expand(B, -1, -1)))
    rep("fwd   matmul(W, yt) -> (B, dm, L)", lambda: torch.matmul(W, yt))
    rep("dgrad matmul(W^T, dout^T) -> (B, d, L)", lambda: torch.matmul(W.t(), dout.transpose(1, 2)))
    rep("dgrad matmul(dout, W)^T  -> (B, L, d)", lambda: torch.matmul(dout, W))
    rep("wgrad bmm(yt, dout).sum(0)      (B x [d x L] @ [L x dm])", lambda: torch.bmm(yt, dout).sum(0))
    rep("wgrad bmm(dout^T, yt^T).sum(0)  (B x [dm x L] @ [L x d])", lambda: torch.bmm(dout.transpose(1, 2), yt.transpose(1, 2)).sum(0))
    d2 = dout.reshape(B * L, dm)
    y2 = yt.transpose(1, 2).reshape(B * L, d)   # copy
    rep("wgrad single GEMM d2^T @ y2 (y2 = contiguous copy, not timed)", lambda: d2.t() @ y2)
    for S in (2, 4):
        ys = yt.view(B, d, S, L // S).permute(0, 2, 1, 3).reshape(B * S, d, L // S)      # copy! (not a view)
        ds = dout.view(B * S, L // S, dm)
        rep(f"wgrad bmm split L by {S} (copy of yt not timed)", lambda: torch.bmm(ys, ds).sum(0))

if __name__ == "__main__":
    main()
