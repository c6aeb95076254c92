"""Why does vms_selective_scan_bwd_dual take ~8 % longer inside the block step than in tools/kbench.py?
The kbench call (same shapes, layouts of the block's node) timed per launch
  (a) back to back,  (b) with the step's GEMM load between the launches,  (c) after a 256 MB fill (operands out of L2 / MALL),
  (d) after 2 ms idle.      usage: python tools/exp_instep_dual.py"""
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "video-mamba-suite_amd"))
import torch  # noqa: E402
import selective_scan_cuda  # noqa: E402

dev, dt = "cuda", torch.bfloat16
b, d, L, N = [int(x) for x in os.environ.get("KB_SHAPE", "8,1024,8192,16").split(",")]
torch.manual_seed(0)
xz = torch.randn(b, 2 * d, L, device=dev, dtype=dt)
u, z = xz[:, :d], xz[:, d:]
delta = (0.5 * torch.rand(d, b, L, device=dev)).to(dt).permute(1, 0, 2)
A = -torch.arange(1, N + 1, device=dev, dtype=torch.float32).repeat(d, 1).contiguous()
B = torch.randn(b, 1, N, L, device=dev, dtype=dt)
C = torch.randn(b, 1, N, L, device=dev, dtype=dt)
D = torch.ones(d, device=dev)
bias = torch.randn(d, device=dev) - 4.0
out, x, _ = selective_scan_cuda.fwd(u, delta, A, B, C, D, z, bias, True)
A2 = A * 1.1
out2, x2, _ = selective_scan_cuda.fwd(u, delta, A2, B, C, D, z, bias, True, reverse=True)
dout = torch.randn(b, d, L, device=dev, dtype=dt)
dxz = torch.empty_like(xz)
dz = dxz[:, d:]
da, db_ = (u, delta, A, B, C, D, bias, x, out), (u, delta, A2, B, C, D, bias, x2, out2)


def call():
    selective_scan_cuda.bwd_dual(da, db_, z, dout, dz, True, keep_fp32=True)


def replay(between=None, n=25):
    ev = []
    for _ in range(n):
        if between is not None:
            between()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        call()
        e1.record()
        ev.append((e0, e1))
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b_) for a, b_ in ev[5:])
    return t[len(t) // 2] * 1e3, t[0] * 1e3


ga = torch.randn(b * L, d, device=dev, dtype=dt)
gw = torch.randn(d, 2 * d, device=dev, dtype=dt)
go = torch.empty(b * L, 2 * d, device=dev, dtype=dt)
big = torch.empty(1 << 28, device=dev, dtype=torch.uint8)


def gemms():
    torch.mm(ga, gw, out=go)
    torch.mm(ga, gw, out=go)


def flush():
    big.zero_()


def idle():
    torch.cuda._sleep(int(2e-3 * 2.4e9))


oa = torch.randn(b * L, d, device=dev, dtype=dt)
ow = torch.randn(d, d, device=dev, dtype=dt)
oo = torch.empty(b * L, d, device=dev, dtype=dt)


def small(n):   # n out_proj-sized GEMMs (137 GFLOP, ~120 us each): what precedes the dual call in the block's backward is two of them
    def f():
        for _ in range(n):
            torch.mm(oa, ow, out=oo)
    return f


for _ in range(30):
    call()
for name, fn in (("back to back", None), ("after 1 out_proj-sized GEMM", small(1)), ("after 2 out_proj-sized GEMMs", small(2)),
                 ("after 4 out_proj-sized GEMMs", small(4)), ("after two GEMMs", gemms), ("after a 256 MB fill", flush), ("after 2 ms idle", idle),
                 ("after GEMMs + fill", lambda: (gemms(), flush())), ("back to back again", None)):
    m, lo = replay(fn)
    print(f"{name:22s}: median {m:8.1f} us   min {lo:8.1f}", flush=True)
