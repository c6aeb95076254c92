"""Timing of formulations of the in_proj / out_proj GEMMs of the bench block on the GPU (library GEMMs only)."""
import torch, sys

def timeit(fn, n=20, w=5):
    for _ in range(w): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

def main():
    dev = "cuda"
    B, L, dm, d2 = 8, 8192, 1024, 2048
    X = torch.randn(B * L, dm, device=dev, dtype=torch.bfloat16)
    W = torch.randn(d2, dm, device=dev, dtype=torch.bfloat16) * 0.02
    G = torch.randn(d2, B * L, device=dev, dtype=torch.bfloat16)
    fl = 2 * B * L * dm * d2
    def rep(name, fn):
        t = timeit(fn)
        print(f"{name:58s} {t:8.1f} us  {fl / t / 1e9:7.2f} PFLOP/s" if fl else f"{name} {t:.1f} us", flush=True)
    rep("fwd   W @ X^T                       (2048x65536, K=1024)", lambda: W @ X.t())
    rep("fwd   F.linear(X, W)                (65536x2048, K=1024)", lambda: torch.nn.functional.linear(X, W))
    rep("dgrad W^T @ G                       (1024x65536, K=2048)", lambda: W.t() @ G)
    rep("dgrad (G^T @ W)                     (65536x1024, K=2048)", lambda: G.t() @ W)
    rep("wgrad G @ X                         (2048x1024, K=65536)", lambda: G @ X)
    rep("wgrad (X^T @ G^T)^T                 (1024x2048, K=65536)", lambda: (X.t() @ G.t()).t())
    for S in (4, 8, 16, 32):
        Gs = G.view(d2, S, B * L // S).permute(1, 0, 2)          # (S, 2048, K/S) strided
        Xs = X.view(S, B * L // S, dm)
        rep(f"wgrad bmm split-K S={S:2d} + sum        ", lambda: torch.bmm(Gs, Xs).sum(0))
        rep(f"wgrad bmm split-K S={S:2d} (no sum)     ", lambda: torch.bmm(Gs, Xs))
    Gc = G.t().contiguous()  # (BL, 2048): channel-last gradient
    rep("wgrad Gc^T @ X  (Gc = (BL, 2048) contiguous)            ", lambda: Gc.t() @ X)

if __name__ == "__main__":
    main()
