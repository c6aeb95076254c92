"""Random-configuration parity sweep of the public ops against the f64 oracle (a measurement helper: the committed tests hold the
fixed grids; this looks for shapes / flag combinations nobody wrote down).
    python tools/fuzz_parity.py [n_scan] [n_conv] [seed] [n_ext] [n_dual]
Every case: selective_scan_fn forward + backward (random batch / dim / dstate / seqlen incl. chunk and vector boundaries, groups,
constant or variable B / C, D, z, delta_bias, softplus, dtype, strided operand layouts) and causal_conv1d_fn (width, bias, SiLU,
channel-last, dtype), compared with oracle/ on the values the kernels saw under tests/test_hip_parity.py's tolerance table.
Prints one line per failure and a summary; exit status 1 if anything failed."""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "video-mamba-suite_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from oracle import oracle as orc
from mamba_ssm.ops.selective_scan_interface import selective_scan_fn
from causal_conv1d import causal_conv1d_fn
import vms_hip

DEV = "cuda"
TOL = {torch.float32: 1e-3, torch.bfloat16: 1e-2, torch.float16: 3e-3}
FACT = dict(out=1, last_state=1, du=2, ddelta=2, dz=2, dB=2, dC=2, dA=5, dD=5, ddelta_bias=5)
f = lambda t: None if t is None else t.detach().float().cpu().numpy()


def rel(a, ref):
    a = f(a).astype(np.float64); ref = np.asarray(ref, np.float64)
    if a.shape != ref.shape:
        return float("inf")
    if not np.isfinite(a).all():
        return float("inf")
    return np.abs(a - ref).max() / max(np.abs(ref).max(), 1e-6)


def strided(t, rng, dims):
    """the same values in another memory layout: a slice of a wider buffer along `dims[0]`, or channel-slowest"""
    mode = rng.choice(["plain", "half", "cslow"])
    if mode == "plain" or t.dim() != 3:
        return t
    if mode == "half":     # the second half of a (B, 2D, L) buffer, like z inside xz
        wide = torch.zeros(t.shape[0], 2 * t.shape[1], t.shape[2], dtype=t.dtype, device=t.device)
        wide[:, t.shape[1]:] = t
        return wide[:, t.shape[1]:]
    return t.permute(1, 0, 2).contiguous().permute(1, 0, 2)   # (L, B L, 1)-strided: what the blocks produce


def scan_case(rng):
    itype = rng.choice([torch.float32, torch.bfloat16, torch.float16])
    N = rng.choice([1, 2, 4, 8, 16, 16, 16, 16, 32, 64])
    L = rng.choice([rng.randint(1, 40), rng.randint(1, 300), rng.choice([8, 16, 64, 128, 256, 1024, 2048, 2056, 4096]),
                    rng.choice([8, 16, 128, 1024, 2048]) + rng.choice([-1, 1, 8, 16, -8]), rng.randint(300, 4500)])
    L = max(1, L)
    dim = rng.choice([1, 2, 3, 4, 5, 8, 12, 16, 24, 32, 48, 64, 96, 100, 128, 192])
    batch = rng.choice([1, 1, 2, 3, 4, 8])
    if rng.random() < 0.5:                 # half of the cases where the fast kernels take the problem: 16 states, whole vectors, >= 32 rows
        N = 16
        dim = rng.choice([32, 64, 96, 128, 160])
        batch = rng.choice([1, 2, 3, 4])
        L = rng.choice([8, 16]) * rng.randint(1, 260)
    if batch * dim * L * N > 6e6:          # the oracle finishes in seconds
        L = max(1, int(6e6 / (batch * dim * N)))
    divs = [g for g in (1, 2, 3, 4) if dim % g == 0]
    groups = rng.choice(divs)
    var_B, var_C = rng.random() < 0.85, rng.random() < 0.85
    has_z, has_D, has_bias, sp = rng.random() < 0.6, rng.random() < 0.8, rng.random() < 0.7, rng.random() < 0.7
    seed = rng.randint(0, 1 << 30)
    g = torch.Generator().manual_seed(seed)
    R = lambda *s: torch.randn(*s, generator=g)
    u = strided(R(batch, dim, L).to(itype).to(DEV), rng, None).requires_grad_()
    delta = strided((0.5 * torch.rand(batch, dim, L, generator=g)).to(itype).to(DEV), rng, None).requires_grad_()
    A = (-0.5 * torch.rand(dim, N, generator=g) - (0.0 if rng.random() < 0.5 else 1e-3)).to(DEV).requires_grad_()
    B = (R(batch, groups, N, L).to(itype) if var_B else R(dim, N)).to(DEV).requires_grad_()
    C = (R(batch, groups, N, L).to(itype) if var_C else R(dim, N)).to(DEV).requires_grad_()
    D = R(dim).to(DEV).requires_grad_() if has_D else None
    z = strided(R(batch, dim, L).to(itype).to(DEV), rng, None).requires_grad_() if has_z else None
    bias = (0.5 * torch.rand(dim, generator=g)).to(DEV).requires_grad_() if has_bias else None
    gout = R(batch, dim, L).to(itype).to(DEV)
    desc = f"scan b{batch} d{dim} N{N} L{L} g{groups} {str(itype)[6:]} varB{int(var_B)} varC{int(var_C)} z{int(has_z)} D{int(has_D)} bias{int(has_bias)} sp{int(sp)} seed{seed}"
    out, last = selective_scan_fn(u, delta, A, B, C, D, z=z, delta_bias=bias, delta_softplus=sp, return_last_state=True)
    kf = vms_hip.last_kernel()
    out.backward(gout)
    kb = vms_hip.last_kernel()
    got = dict(out=out, last_state=last, du=u.grad, ddelta=delta.grad, dA=A.grad, dB=B.grad, dC=C.grad, dD=None if D is None else D.grad,
               dz=None if z is None else z.grad, ddelta_bias=None if bias is None else bias.grad)
    o = orc.scan_fwd(f(u), f(delta), f(A), f(B), f(C), f(D), f(z), f(bias), sp, prec="f64")
    ob = orc.scan_bwd(f(u), f(delta), f(A), f(B), f(C), f(D), f(z), f(bias), f(gout), sp, prec="f64")
    want = dict(out=o["out_z"] if z is not None else o["out"], last_state=o["last_state"], **ob)
    bad = []
    for k, fac in FACT.items():
        if want.get(k) is None or got.get(k) is None:
            continue
        e = rel(got[k], want[k])
        if not e <= TOL[itype] * fac:
            bad.append(f"{k} {e:.2e}")
    return desc + f" [{kf} | {kb}]", bad


def ext_case(rng):
    """the extension mirror (selective_scan_cuda.fwd / .bwd) with its direction extensions: reverse, reverse_from (per batch entry), the
    gated output added into another tensor, dz accumulated -- against the oracle run on per-entry flipped copies"""
    import selective_scan_cuda as ssc
    itype = rng.choice([torch.float32, torch.bfloat16, torch.float16])
    N = rng.choice([4, 8, 16, 16, 16, 32])
    batch = rng.choice([1, 2, 3, 4, 6])
    dim = rng.choice([4, 8, 16, 32, 64, 96, 128])
    L = max(1, rng.choice([rng.randint(1, 64), 8 * rng.randint(1, 200), 16 * rng.randint(1, 150), rng.randint(64, 2500), 2048 + 8 * rng.randint(0, 40)]))
    if rng.random() < 0.45:      # many rows, whole vectors, 16 states: the 4-rows-per-wave backward kernels (and their split / mixed forms)
        N, dim, batch = 16, rng.choice([256, 384, 512, 768]), rng.choice([2, 3, 4])
        L = 8 * rng.randint(1, 60) if rng.random() < 0.7 else 2048 + 8 * rng.randint(1, 30)
    if batch * dim * L * N > 1.2e7:
        L = max(8, int(1.2e7 / (batch * dim * N)) // 8 * 8)
    groups = rng.choice([g for g in (1, 2, 4) if dim % g == 0])
    mode = rng.choice(["fwd", "rev", "from"])
    rf = rng.randint(1, batch - 1) if (mode == "from" and batch > 1) else 0
    reverse = mode == "rev"
    if mode == "from" and batch == 1:
        reverse = True
    has_z, has_D, has_bias = rng.random() < 0.7, rng.random() < 0.8, rng.random() < 0.7
    acc_out, acc_dz = has_z and rng.random() < 0.4, has_z and rng.random() < 0.4
    seed = rng.randint(0, 1 << 30)
    g = torch.Generator().manual_seed(seed)
    R = lambda *s: torch.randn(*s, generator=g)
    u = R(batch, dim, L).to(itype).to(DEV)
    delta = (0.5 * torch.rand(batch, dim, L, generator=g)).to(itype).to(DEV)
    A = (-0.5 * torch.rand(dim, N, generator=g) - 0.02).to(DEV)
    B = R(batch, groups, N, L).to(itype).to(DEV)
    C = R(batch, groups, N, L).to(itype).to(DEV)
    D = R(dim).to(DEV) if has_D else None
    z = R(batch, dim, L).to(itype).to(DEV) if has_z else None
    bias = (0.5 * torch.rand(dim, generator=g)).to(DEV) if has_bias else None
    dout = R(batch, dim, L).to(itype).to(DEV)
    desc = f"ext b{batch} d{dim} N{N} L{L} g{groups} {str(itype)[6:]} {mode} rf{rf} z{int(has_z)} D{int(has_D)} bias{int(has_bias)} accout{int(acc_out)} accdz{int(acc_dz)} seed{seed}"
    base_out = R(batch, dim, L).to(itype).to(DEV) if acc_out else None
    into = base_out.clone() if acc_out else None
    res = ssc.fwd(u, delta, A, B, C, D, z, bias, True, reverse=reverse, out_z_into=into, reverse_from=rf)
    out, x = res[0], res[1]
    out_z = res[2] if has_z else None
    kf = vms_hip.last_kernel()
    base_dz = R(batch, dim, L).to(itype).to(DEV) if acc_dz else None
    dz = base_dz.clone() if acc_dz else (torch.empty_like(z) if has_z else None)
    rb = ssc.bwd(u, delta, A, B, C, D, z, bias, dout, x, out, dz, True, False, reverse=reverse, accumulate_dz=acc_dz, reverse_from=rf)
    kb = vms_hip.last_kernel()
    # oracle: flip the entries that run right-to-left, run left-to-right, flip the per-position results back
    revmask = np.array([reverse or (rf > 0 and bi >= rf) for bi in range(batch)])

    def fl(a):
        if a is None:
            return None
        a = a.copy()
        a[revmask] = a[revmask][..., ::-1]
        return a
    o = orc.scan_fwd(fl(f(u)), fl(f(delta)), f(A), fl(f(B)), fl(f(C)), f(D), fl(f(z)), f(bias), True, prec="f64")
    ob = orc.scan_bwd(fl(f(u)), fl(f(delta)), f(A), fl(f(B)), fl(f(C)), f(D), fl(f(z)), f(bias), fl(f(dout)), True, prec="f64")
    bad = []

    def cmp(name, a, ref, fac):
        e = rel(a, ref)
        if not e <= TOL[itype] * fac:
            bad.append(f"{name} {e:.2e}")
    cmp("out", out, fl(o["out"]), 1)
    if has_z:
        want = fl(o["out_z"]) + (f(base_out).astype(np.float64) if acc_out else 0)
        cmp("out_z", out_z, want, 1.5 if acc_out else 1)
        want = fl(ob["dz"]) + (f(base_dz).astype(np.float64) if acc_dz else 0)
        cmp("dz", rb[7], want, 2.5 if acc_dz else 2)
    cmp("du", rb[0], fl(ob["du"]), 2); cmp("ddelta", rb[1], fl(ob["ddelta"]), 2)
    cmp("dA", rb[2], ob["dA"], 5); cmp("dB", rb[3], fl(ob["dB"]), 2); cmp("dC", rb[4], fl(ob["dC"]), 2)
    if has_D:
        cmp("dD", rb[5], ob["dD"], 5)
    if has_bias:
        cmp("ddelta_bias", rb[6], ob["ddelta_bias"], 5)
    return desc + f" [{kf} | {kb}]", bad


def dual_case(rng):
    """selective_scan_cuda.bwd_dual: both directions of a bidirectional block (own parameters each, shared z and dout) from one call,
    against two oracle runs (the second on flipped inputs); dz = the sum of both directions' dz"""
    import selective_scan_cuda as ssc
    itype = rng.choice([torch.bfloat16, torch.float16, torch.float32])
    N = 16
    batch, dim = rng.choice([(1, 64), (2, 128), (4, 256), (8, 256), (4, 512), (8, 512), (8, 768), (4, 1024), (5, 1024)])   # (8, 512), (4, 1024): whole rounds of 8-wave workgroups
    L = 8 * rng.randint(1, 40) if rng.random() < 0.8 else rng.randint(1, 200)
    if batch * dim * L * N > 1.2e7:
        L = max(8, int(1.2e7 / (batch * dim * N)) // 8 * 8)
    seed = rng.randint(0, 1 << 30)
    g = torch.Generator().manual_seed(seed)
    R = lambda *s: torch.randn(*s, generator=g)
    z = R(batch, dim, L).to(itype).to(DEV)
    dout = R(batch, dim, L).to(itype).to(DEV)
    dirs = []
    for rev in (False, True):
        u = R(batch, dim, L).to(itype).to(DEV)
        delta = (0.5 * torch.rand(batch, dim, L, generator=g)).to(itype).to(DEV)
        A = (-0.5 * torch.rand(dim, N, generator=g) - 0.02).to(DEV)
        B = R(batch, 1, N, L).to(itype).to(DEV)
        C = R(batch, 1, N, L).to(itype).to(DEV)
        D = R(dim).to(DEV)
        bias = (0.5 * torch.rand(dim, generator=g)).to(DEV)
        out, x, _ = ssc.fwd(u, delta, A, B, C, D, z, bias, True, reverse=rev)
        dirs.append((u, delta, A, B, C, D, bias, x, out))
    dz = torch.full_like(z, float("nan"))
    ra, rb = ssc.bwd_dual(dirs[0], dirs[1], z, dout, dz, True, keep_fp32=True)
    k = vms_hip.last_kernel()
    desc = f"dual b{batch} d{dim} L{L} {str(itype)[6:]} seed{seed} [{k}]"
    bad = []
    dz_want = 0
    for got, (u, delta, A, B, C, D, bias, x, out), rev in ((ra, dirs[0], False), (rb, dirs[1], True)):
        fl = (lambda a: np.ascontiguousarray(a[..., ::-1])) if rev else (lambda a: a)
        ob = orc.scan_bwd(fl(f(u)), fl(f(delta)), f(A), fl(f(B)), fl(f(C)), f(D), fl(f(z)), f(bias), fl(f(dout)), True, prec="f64")
        for i, (name, fac, flip) in enumerate((("du", 2, True), ("ddelta", 2, True), ("dA", 5, False), ("dB", 2, True), ("dC", 2, True),
                                               ("dD", 5, False), ("ddelta_bias", 5, False))):
            ref = fl(ob[name]) if flip else ob[name]
            e = rel(got[i], ref)
            if not e <= TOL[itype] * fac:
                bad.append(f"{'b' if rev else 'a'}.{name} {e:.2e}")
        dz_want = dz_want + fl(ob["dz"])
    e = rel(ra[7], dz_want)
    if not e <= TOL[itype] * 2:
        bad.append(f"dz {e:.2e}")
    return desc, bad


def conv_case(rng):
    itype = rng.choice([torch.float32, torch.bfloat16, torch.float16])
    W = rng.choice([2, 3, 4, 4])
    L = max(1, rng.choice([rng.randint(1, 40), rng.randint(1, 600), rng.choice([8, 64, 512, 2048, 4096]) + rng.choice([-1, 0, 1, 8]), rng.randint(600, 5000)]))
    dim = rng.choice([1, 2, 3, 8, 16, 31, 64, 96, 128, 257, 768])
    batch = rng.choice([1, 2, 3, 8])
    if batch * dim * L > 4e6:
        L = max(1, int(4e6 / (batch * dim)))
    silu, has_bias, cl = rng.random() < 0.6, rng.random() < 0.7, rng.random() < 0.3 and dim % 8 == 0   # (channel-last: dim % 8, as the reference)
    wtype = torch.float32 if rng.random() < 0.6 else itype
    seed = rng.randint(0, 1 << 30)
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(batch, dim, L, generator=g).to(itype).to(DEV)
    if cl:
        x = x.transpose(1, 2).contiguous().transpose(1, 2)     # channel-last memory, (B, D, L) shape
    x.requires_grad_()
    w = torch.randn(dim, W, generator=g).to(wtype).to(DEV).requires_grad_()
    b = torch.randn(dim, generator=g).to(wtype).to(DEV).requires_grad_() if has_bias else None
    gout = torch.randn(batch, dim, L, generator=g).to(itype).to(DEV)
    if cl:
        gout = gout.transpose(1, 2).contiguous().transpose(1, 2)
    desc = f"conv b{batch} d{dim} L{L} W{W} {str(itype)[6:]} w{str(wtype)[6:]} silu{int(silu)} bias{int(has_bias)} cl{int(cl)} seed{seed}"
    out = causal_conv1d_fn(x, w, b, "silu" if silu else None)
    kf = vms_hip.last_kernel()
    out.backward(gout)
    o = orc.conv_fwd(f(x), f(w), f(b), silu, prec="f64")
    ob = orc.conv_bwd(f(x), f(w), f(b), f(gout), silu, prec="f64")
    bad = []
    checks = [("out", out, o, 1), ("dx", x.grad, ob["dx"], 2), ("dweight", w.grad, ob["dweight"], 5)]
    if has_bias:
        checks.append(("dbias", b.grad, ob["dbias"], 5))
    for k, a, ref, fac in checks:
        e = rel(a, ref)
        if not e <= TOL[itype] * fac:
            bad.append(f"{k} {e:.2e}")
    return desc + f" [{kf}]", bad


def main():
    n_scan = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    n_conv = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    rng = random.Random(int(sys.argv[3]) if len(sys.argv) > 3 else 0)
    n_ext = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    n_dual = int(sys.argv[5]) if len(sys.argv) > 5 else 0
    fails, kernels = 0, {}
    for kind, n, fn in (("scan", n_scan, scan_case), ("conv", n_conv, conv_case), ("ext", n_ext, ext_case), ("dual", n_dual, dual_case)):
        for i in range(n):
            try:
                desc, bad = fn(rng)
            except Exception as e:   # a declined problem must raise a clear error, not crash: report what it said
                desc, bad = f"{kind} case {i}", [f"EXCEPTION {type(e).__name__}: {str(e)[:200]}"]
            k = desc[desc.rfind("["):]
            kernels[k] = kernels.get(k, 0) + 1
            if bad:
                fails += 1
                print("FAIL", desc, "::", "; ".join(bad), flush=True)
    print(f"{n_scan} scan + {n_conv} conv + {n_ext} extension-level + {n_dual} dual-backward cases, {fails} failed")
    for k, v in sorted(kernels.items(), key=lambda kv: -kv[1]):
        print(f"  {v:4d}  {k}")
    sys.exit(1 if fails else 0)


if __name__ == "__main__":
    main()
