"""What does the library (hipBLASLt through torch) reach for the block's large projection GEMMs in its BEST plain layout?
(is the (B, 2D, L)-direct layout of ops/projections.py costing anything?)

For every logical GEMM C[M, N] = A[M, K] @ B[K, N] of the (8, 8192, 1024) bf16 block, all 8 storage combinations are timed
on contiguous, pre-laid-out operands: A as (M, K) or (K, M), B as (K, N) or (N, K), C written as (M, N) or (N, M).
"here" marks the combination mamba_ssm/ops/projections.py runs.  usage: python tools/gemm_layouts.py
"""
import itertools
import sys
import torch

dev, bf = "cuda", torch.bfloat16


def timeit(fn, n=12, w=4):
    for _ in range(w):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def probe(name, M, N, K, here):
    """here = (a_km, b_nk, c_nm) flags of the formulation the product uses"""
    flop = 2.0 * M * N * K
    print(f"== {name}: C[{M}, {N}] = A[{M}, {K}] @ B[{K}, {N}]  ({flop / 1e9:.0f} GFLOP; compute floor {flop / 2.5e15 * 1e6:.0f} us, "
          f"memory floor {(M * K + K * N + M * N) * 2 / 8e12 * 1e6:.0f} us)")
    best = None
    for a_km, b_nk, c_nm in itertools.product((False, True), repeat=3):
        A = (torch.randn(K, M, device=dev, dtype=bf).t() if a_km else torch.randn(M, K, device=dev, dtype=bf))
        Bm = (torch.randn(N, K, device=dev, dtype=bf).t() if b_nk else torch.randn(K, N, device=dev, dtype=bf)) * 0.03
        if c_nm:   # C^T = B^T @ A^T written as a contiguous (N, M)
            out = torch.empty(N, M, device=dev, dtype=bf)
            fn = lambda: torch.matmul(Bm.t(), A.t(), out=out)
        else:
            out = torch.empty(M, N, device=dev, dtype=bf)
            fn = lambda: torch.matmul(A, Bm, out=out)
        t = timeit(fn)
        tag = "  <- here" if (a_km, b_nk, c_nm) == here else ""
        print(f"   A {'(K,M)' if a_km else '(M,K)'}  B {'(N,K)' if b_nk else '(K,N)'}  C {'(N,M)' if c_nm else '(M,N)'} : "
              f"{t:7.1f} us  {flop / t / 1e9:6.3f} PFLOP/s{tag}", flush=True)
        if best is None or t < best[0]:
            best = (t, a_km, b_nk, c_nm)
        del A, Bm, out
    print(f"   best {best[0]:.1f} us = {flop / best[0] / 1e9:.3f} PFLOP/s = {flop / best[0] / 1e9 / 2.5 * 100:.1f} % of the dense bf16 MFMA peak")


def main():
    BL, dm, d2, di = 65536, 1024, 2048, 1024
    torch.manual_seed(0)
    # in_proj forward: xz^T (d2, BL) = W (d2, dm) @ X^T; X stored (BL, dm) -> as C[M=d2, N=BL]: A = W (M,K), B = X^T stored (N,K)
    probe("in_proj fwd", d2, BL, dm, (False, True, False))
    # in_proj dgrad: dH (BL, dm) = G^T (BL, d2) @ W (d2, dm); G stored (d2, BL) -> A stored (K,M); B (K,N)
    probe("in_proj dgrad", BL, dm, d2, (True, False, False))
    # out_proj forward, per batch entry (8 of them, batched): O (L, dm) = Y^T (L, di) @ W^T (di, dm); Y stored (di, L) -> A (K,M); W stored (dm, di) -> B (N,K)
    probe("out_proj fwd (one launch over the batch, shown as M = B L)", BL, dm, di, (True, True, False))
    # out_proj dgrad: dY (di, L) = W^T (di, dm) @ dout^T (dm, L); W stored (dm, di) -> A (K,M); dout stored (L, dm) -> B (N,K)
    probe("out_proj dgrad (M = d_inner, N = B L)", di, BL, dm, (True, True, False))
    if "wgrad" in sys.argv:
        probe("in_proj wgrad (K = B L, unsplit)", d2, dm, BL, (False, False, False))


if __name__ == "__main__":
    main()
