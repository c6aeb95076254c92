"""vms_conv_xproj_dual (both directions' conv1d + SiLU and x_proj in one pass over x) vs the two launches it replaces
(vms_causal_conv1d_fwd_dual + vms_proj_kred), at the BASELINE configs' shapes, bf16.  usage: python tools/kb_cxp.py"""
import os
import sys
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "video-mamba-suite_amd"))
import vms_hip  # noqa: E402

dev, bf = "cuda", torch.bfloat16


def timeit(fn, n=40, w=10):
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


def one(b, d, L, R, N):
    m = R + 2 * N
    torch.manual_seed(0)
    xz = torch.randn(b, 2 * d, L, device=dev, dtype=bf)
    x = xz[:, :d, :]
    cw, cwb = torch.randn(d, 4, device=dev) * 0.4, torch.randn(d, 4, device=dev) * 0.4
    cb, cbb = torch.randn(d, device=dev) * 0.2, torch.randn(d, device=dev) * 0.2
    wx, wxb = [(torch.randn(m, d, device=dev) * d ** -0.5).to(bf) for _ in range(2)]
    o, ob = torch.empty(b, d, L, device=dev, dtype=bf), torch.empty(b, d, L, device=dev, dtype=bf)
    xd, xdb = torch.empty(b, m, L, device=dev, dtype=bf), torch.empty(b, m, L, device=dev, dtype=bf)
    act = b * d * L * 2 / 1e6
    t_conv = timeit(lambda: vms_hip.conv_fwd_dual(x, cw, cb, o, cwb, cbb, ob, True))
    t_kred = timeit(lambda: vms_hip.proj_kred(wx, o, xd, wxb, ob, xdb))
    t_both = timeit(lambda: (vms_hip.conv_fwd_dual(x, cw, cb, o, cwb, cbb, ob, True), vms_hip.proj_kred(wx, o, xd, wxb, ob, xdb)))
    ts = {t: timeit(lambda: vms_hip.conv_xproj_dual(x, cw, cb, o, cwb, cbb, ob, wx, wxb, xd, xdb, tile=t)) for t in (0, 64, 128)}
    print(f"(b, d, L, m) = ({b}, {d}, {L}, {m}): conv_fwd_dual {t_conv:6.1f} + proj_kred {t_kred:6.1f} = {t_both:6.1f} us back to back | fused auto {ts[0]:6.1f}  "
          f"tile 64 {ts[64]:6.1f}  128 {ts[128]:6.1f} us -> {3 * act / min(ts.values()):.2f} TB/s of its 3 passes ({3 * act:.0f} MB)", flush=True)


def main():
    a = torch.randn(4096, 4096, device=dev, dtype=bf)
    for _ in range(20):
        a @ a
    if "KB_SHAPE" in os.environ:
        shapes = [tuple(int(v) for v in os.environ["KB_SHAPE"].split(","))]
    else:
        shapes = [(8, 1024, 8192, 64, 16), (8, 768, 3136, 48, 16), (1, 768, 65536, 48, 16), (4, 512, 2304, 32, 16), (8, 384, 3152, 24, 16)]
    for s in shapes:
        one(*s)


if __name__ == "__main__":
    main()
