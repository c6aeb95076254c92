"""vms_proj_kred (the inner node's two products that contract over the channels: x_dbl = W_x @ conv_out, d_dt = W_dt^T @ ddelta)
vs the library GEMMs the node ran until round 4, per tile width, at the BASELINE configs' shapes (or KB_SHAPE=b,d,L,R,N), bf16.
usage: python tools/kb_kred.py"""
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
    K2 = R + 2 * N
    torch.manual_seed(0)
    conv_out = [torch.randn(b, d, L, device=dev, dtype=bf) for _ in range(2)]
    ddelta = torch.randn(d, b, L, device=dev, dtype=bf).permute(1, 0, 2)      # the scan's channel-slowest layout
    w_x = [(torch.randn(K2, d, device=dev) * d ** -0.5).to(bf) for _ in range(2)]
    w_dt = (torch.randn(d, R, device=dev) * d ** -0.5).to(bf)
    xp = [torch.empty(b, K2, L, device=dev, dtype=bf) for _ in range(2)]
    dx_dbl = torch.empty(b, K2, L, device=dev, dtype=bf)
    d_dt = dx_dbl[:, :R, :]
    wdtt = w_dt.t()
    act = b * d * L * 2 / 1e6
    print(f"== (b, d, L, R, N) = ({b}, {d}, {L}, {R}, {N}): activation {act:.0f} MB, floor at 8 TB/s {act / 8:.1f} us", flush=True)
    t_lib = timeit(lambda: torch.matmul(w_x[0], conv_out[0], out=xp[0]))
    ts = {t: timeit(lambda: vms_hip.proj_kred(w_x[0], conv_out[0], xp[0], tile=t)) for t in (0, 64, 128, 256)}
    print(f"x_dbl = W_x @ conv_out   (m = {K2:2d}): library {t_lib:6.1f} us | kred auto {ts[0]:6.1f}  tile 64 {ts[64]:6.1f}  128 {ts[128]:6.1f}  256 {ts[256]:6.1f} us"
          f" -> {act / min(ts.values()):.2f} TB/s", flush=True)
    t_lib2 = timeit(lambda: (torch.matmul(w_x[0], conv_out[0], out=xp[0]), torch.matmul(w_x[1], conv_out[1], out=xp[1])))
    td = {t: timeit(lambda: vms_hip.proj_kred(w_x[0], conv_out[0], xp[0], w_x[1], conv_out[1], xp[1], tile=t)) for t in (0, 64, 128, 256)}
    print(f"both directions' x_dbl  (m = {K2:2d}): library {t_lib2:6.1f} us | kred auto {td[0]:6.1f}  tile 64 {td[64]:6.1f}  128 {td[128]:6.1f}  256 {td[256]:6.1f} us"
          f" -> {2 * act / min(td.values()):.2f} TB/s", flush=True)
    t_lib = timeit(lambda: torch.bmm(wdtt.unsqueeze(0).expand(b, -1, -1), ddelta, out=d_dt))
    if vms_hip.proj_kred_eligible(wdtt, ddelta, d_dt):
        tt = {t: timeit(lambda: vms_hip.proj_kred(wdtt, ddelta, d_dt, tile=t)) for t in (0, 64, 128, 256)}
        print(f"d_dt = W_dt^T @ ddelta   (m = {R:2d}): library {t_lib:6.1f} us | kred auto {tt[0]:6.1f}  tile 64 {tt[64]:6.1f}  128 {tt[128]:6.1f}  256 {tt[256]:6.1f} us"
              f" -> {act / min(tt.values()):.2f} TB/s", flush=True)
    else:
        print(f"d_dt = W_dt^T @ ddelta   (m = {R:2d}): library {t_lib:6.1f} us | kred: not eligible", flush=True)


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
