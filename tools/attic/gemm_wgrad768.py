"""K-split counts for the weight-gradient GEMMs of the d_model = 768 configs (library GEMMs): in_proj (C = 1536) and out_proj
(C = 768) at rows = B L = 65536 (configs[4], batch 1) and 25088 (configs[2], 8 x 3136)."""
import torch
from gemm_wgrad import timeit


def main():
    dev = "cuda"
    for rows, b in ((65536, 1), (25088, 8)):
        L = rows // b
        for name, C, dm in (("in_proj", 1536, 768), ("out_proj", 768, 768)):
            G = torch.randn(b, C, L, device=dev, dtype=torch.bfloat16)            # (B, C, L): the scans' layout
            X = torch.randn(b, L, dm, device=dev, dtype=torch.bfloat16)           # (B, L, d_model)
            fl = 2 * rows * C * dm
            for S in (1, 2, 4, 7, 8, 14, 16, 28, 32, 56, 64):
                if L % S:
                    continue
                Ls = L // S
                Gs = G.view(b, C, S, Ls).permute(0, 2, 1, 3).reshape(b * S, C, Ls) if b == 1 else None
                if b == 1:
                    Xs = X.view(b * S, Ls, dm)
                    fn = lambda: torch.bmm(Gs, Xs).sum(0, dtype=torch.float32)
                    fn2 = lambda: torch.bmm(Xs.transpose(1, 2), Gs.transpose(1, 2)).sum(0, dtype=torch.float32)
                else:
                    G4 = G.view(b, C, S, Ls).permute(0, 2, 1, 3)                  # (b, S, C, Ls) strided
                    X4 = X.view(b, S, Ls, dm)
                    fn = lambda: torch.matmul(G4, X4).sum((0, 1), dtype=torch.float32)
                    fn2 = lambda: torch.matmul(X4.transpose(2, 3), G4.transpose(2, 3)).sum((0, 1), dtype=torch.float32)
                t, t2 = timeit(fn), timeit(fn2)
                print(f"rows {rows:6d} {name:8s} S={S:3d} (b*S = {b * S:3d})  G X: {t:7.1f} us {fl / t / 1e9:5.2f} PF   X^T G^T: {t2:7.1f} us {fl / t2 / 1e9:5.2f} PF", flush=True)


if __name__ == "__main__":
    main()
