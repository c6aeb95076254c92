"""in_proj / out_proj weight gradients as bmm over S K-slices of the flattened rows (the form mamba_ssm/ops/projections.py runs)."""
import torch
from gemm_wgrad import timeit

dev = "cuda"
for rows, C, dm in ((65536, 2048, 1024), (65536, 1024, 1024), (65536, 1536, 768), (65536, 768, 768), (25088, 1536, 768), (25088, 768, 768),
                    (25216, 1536, 384), (4608, 2048, 512)):
    G = torch.randn(C, rows, device=dev, dtype=torch.bfloat16)        # (C, B L): g2 of InProjFn.backward
    X = torch.randn(rows, dm, device=dev, dtype=torch.bfloat16)
    fl = 2 * rows * C * dm
    for S in (1, 2, 3, 4, 6, 7, 8, 9, 12, 14, 16, 18, 24, 28, 32):
        if rows % S:
            continue
        fn = lambda: torch.bmm(G.view(C, S, rows // S).permute(1, 0, 2), X.view(S, rows // S, dm)).sum(0, dtype=torch.float32)
        t = timeit(fn)
        print(f"rows {rows:6d} C {C:5d} dm {dm:5d}  S={S:3d} ({rows // S:6d} per slice)  {t:7.1f} us {fl / t / 1e9:5.2f} PF", flush=True)
