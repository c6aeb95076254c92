"""The inner node's small projections: hand-written MFMA kernels (csrc/inner_proj.hip) vs the library GEMMs the node used
until round 3, at the block shape (KB_SHAPE=b,d,L,R,N; default 8,1024,8192,64,16), bf16.  usage: python tools/kb_proj.py"""
import os
import sys
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "video-mamba-suite_amd"))
import vms_hip  # noqa: E402

dev, bf = "cuda", torch.bfloat16


def timeit(fn, n=30, w=10):
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


def main():
    b, d, L, R, N = (int(v) for v in os.environ.get("KB_SHAPE", "8,1024,8192,64,16").split(","))
    K2 = R + 2 * N
    torch.manual_seed(0)
    # warm the clocks
    a = torch.randn(4096, 4096, device=dev, dtype=bf)
    for _ in range(20):
        a @ a
    conv_out = torch.randn(b, d, L, device=dev, dtype=bf)
    x_dbl = torch.randn(b, K2, L, device=dev, dtype=bf)
    ddelta = torch.randn(b, d, L, device=dev, dtype=bf)
    dconv = torch.randn(b, d, L, device=dev, dtype=bf)
    w_dt = (torch.randn(d, R, device=dev) * 0.1).to(bf)
    w_x = (torch.randn(K2, d, device=dev) * 0.1).to(bf)
    delta = torch.empty(b, d, L, device=dev, dtype=bf)
    dx_dbl = torch.randn(b, K2, L, device=dev, dtype=bf)
    MB = 1e6
    act = b * d * L * 2 / MB
    small = lambda k: b * k * L * 2 / MB
    rows = []

    def line(name, t_lib, t_new, mbytes):
        rows.append((name, t_lib, t_new, mbytes))
        print(f"{name:42s} library {t_lib:7.1f} us   MFMA kernel {t_new:7.1f} us   {mbytes:6.0f} MB -> {mbytes / t_new / 1e3 * 1e3:5.2f} TB/s"
              f" (floor at 8 TB/s {mbytes / 8:5.1f} us)", flush=True)

    dt_in = x_dbl[:, :R, :]
    line("delta = W_dt @ x_dbl[:R]", timeit(lambda: torch.matmul(w_dt, dt_in, out=delta)),
         timeit(lambda: vms_hip.proj_apply(w_dt, dt_in, delta, False)), act + small(R))
    wxt = w_x.t()
    line("dconv_out += W_x^T @ dx_dbl", timeit(lambda: dconv.baddbmm_(wxt.expand(b, -1, -1), dx_dbl)),
         timeit(lambda: vms_hip.proj_apply(wxt, dx_dbl, dconv, True)), 2 * act + small(K2))
    dw1 = torch.zeros(R, d, device=dev)
    line("dW_dt = ddelta @ x_dbl[:R]^T (+ sum over b)", timeit(lambda: torch.sum(torch.matmul(ddelta, dt_in.transpose(1, 2)), 0)),
         timeit(lambda: vms_hip.proj_wgrad(dt_in, ddelta, dw1)), act + small(R))
    dw2 = torch.zeros(K2, d, device=dev)
    line("dW_x = dx_dbl @ conv_out^T (+ sum over b)", timeit(lambda: torch.sum(torch.matmul(dx_dbl, conv_out.transpose(1, 2)), 0)),
         timeit(lambda: vms_hip.proj_wgrad(dx_dbl, conv_out, dw2)), act + small(K2))
    d_dt = torch.empty(b, R, L, device=dev, dtype=bf)
    wdtt = w_dt.t()
    t = timeit(lambda: torch.matmul(wdtt, ddelta, out=d_dt))
    print(f"{'d_dt = W_dt^T @ ddelta (library only)':42s} library {t:7.1f} us   {act + small(R):6.0f} MB (floor {(act + small(R)) / 8:5.1f} us)")
    xp = torch.empty(b, K2, L, device=dev, dtype=bf)
    t = timeit(lambda: torch.matmul(w_x, conv_out, out=xp))
    print(f"{'x_dbl = W_x @ conv_out (library only)':42s} library {t:7.1f} us   {act + small(K2):6.0f} MB (floor {(act + small(K2)) / 8:5.1f} us)")
    # the backward's tail: library GEMMs + conv backward vs the one-pass kernel
    import causal_conv1d_cuda
    xz = torch.randn(b, 2 * d, L, device=dev, dtype=bf)
    x = xz[:, :d, :]
    conv_w, conv_b = torch.randn(d, 4, device=dev) * 0.3, torch.randn(d, device=dev) * 0.1
    dxz = torch.empty_like(xz)
    dxh = dxz[:, :d, :]

    def tail_lib():
        torch.sum(torch.matmul(dx_dbl, conv_out.transpose(1, 2)), 0)
        dconv.baddbmm_(wxt.expand(b, -1, -1), dx_dbl)
        causal_conv1d_cuda.causal_conv1d_bwd(x, conv_w, conv_b, dconv, dxh, True)

    zw, zb, zx = torch.zeros(d, 4, device=dev), torch.zeros(d, device=dev), torch.zeros(K2, d, device=dev)
    t_lib = timeit(tail_lib)
    print(f"tail: dW_x + dconv_out += + conv1d backward, library + HIP conv: {t_lib:7.1f} us", flush=True)
    for tp in (0, 4, 8, 16, 32, 64):
        t_new = timeit(lambda: vms_hip.proj_conv_bwd(x, dconv, dx_dbl, w_x, conv_w, conv_b, dxh, zw, zb, zx, tiles_per_wg=tp))
        print(f"tail: vms_proj_conv_bwd tiles_per_wg {tp:3d}: {t_new:7.1f} us  ({3 * act / t_new / 1e3 * 1e3:5.2f} TB/s of its 3 passes)", flush=True)
    if "sweep" in sys.argv:
        for tp in (2, 4, 8, 16, 32, 64, 128):
            print(f"tiles_per_wg {tp:3d}:  apply(dt_proj) {timeit(lambda: vms_hip.proj_apply(w_dt, dt_in, delta, False, tp)):6.1f}"
                  f"  apply+acc(x_proj^T) {timeit(lambda: vms_hip.proj_apply(wxt, dx_dbl, dconv, True, tp)):6.1f}"
                  f"  wgrad(m={R}) {timeit(lambda: vms_hip.proj_wgrad(dt_in, ddelta, dw1, tp)):6.1f}"
                  f"  wgrad(m={K2}) {timeit(lambda: vms_hip.proj_wgrad(dx_dbl, conv_out, dw2, tp)):6.1f} us", flush=True)
    print(f"sum of the four replaced GEMMs: library {sum(r[1] for r in rows):.1f} us, MFMA kernels {sum(r[2] for r in rows):.1f} us")


if __name__ == "__main__":
    main()
