"""GPU parity tests (-m gpu): the HIP path, called through the C ABI (ctypes -> libvms_hip.so), against
 (a) the committed golden vectors produced by the reference's PyTorch path, and
 (b) the CPU oracle (oracle/vms_oracle.c, f64 arithmetic) on the same seeded inputs,
plus size-independent checks at BASELINE.json's full size.

Tolerances (BASELINE.json north_star): 1e-3 for fp32 I/O, 1e-2 for bf16 I/O, relative to the
tensor's scale (max |ref|); reductions over L*batch (dA, dD, dbias, dweight) get the factor the
reference's own tests give them (test_selective_scan.py:137-149)."""
import os
import sys

import numpy as np
import pytest
import torch

from conftest import golden_names, load_golden


def _dbg():
    """vms_hip.debug: the one object that holds the test / profiling switches of the Python layers (set with monkeypatch.setattr)"""
    import vms_hip
    return vms_hip.debug

pytestmark = pytest.mark.gpu

DEV = "cuda"
# ---- the one tolerance table of this file -------------------------------------------------------------------------
# TOL = north_star's bar per I/O dtype (1e-3 fp32, 1e-2 bf16; fp16 in between), relative to the tensor's scale
# (max |reference|).  Every comparison is TOL[dtype] x a factor from FACTOR, by what the tensor is and what it is
# compared with:
#   "oracle"  the f64 C oracle on the very (rounded) inputs the kernel saw        -- the tight one
#   "golden"  the fixture the reference's own fp32 PyTorch path produced          -- carries that path's own rounding
# Non-reduced tensors (one output element per input element) stay at 1x against the oracle for forward results, 2x for
# gradients (two chained recurrences) and 2x against the fixtures; sums over rows (dB, dC) 2x; sums over batch x L
# (dA, dD, ddelta_bias, dweight, dbias, norm dweight / dbias) 5x -- the reference's own tests allow those 10x
# (test_selective_scan.py:137-149).
TOL = {torch.float32: 1e-3, torch.bfloat16: 1e-2, torch.float16: 3e-3}
FACTOR = {
    "fwd": {"oracle": 1, "golden": 2},            # out, out_z, last_state, conv out, norm y
    "grad": {"oracle": 2, "golden": 2},           # du, ddelta, dz, conv dx, norm dx
    "row_sum": {"oracle": 2, "golden": 2},        # dB, dC
    "batch_sum": {"oracle": 5, "golden": 5},      # dA, dD, ddelta_bias, dweight, dbias
}
KIND = {"out": "fwd", "last_state": "fwd", "du": "grad", "ddelta": "grad", "dz": "grad", "dB": "row_sum", "dC": "row_sum",
        "dA": "batch_sum", "dD": "batch_sum", "ddelta_bias": "batch_sum"}


def tol_for(name, itype, vs):
    return TOL[itype] * FACTOR[KIND[name]][vs]


def G(a, dtype=torch.float32, grad=False):
    t = torch.tensor(np.asarray(a), dtype=dtype, device=DEV)
    return t.requires_grad_() if grad else t


def rel_err(a, ref):
    a = a.detach().float().cpu().numpy().astype(np.float64)
    if torch.is_tensor(ref):
        ref = ref.detach().float().cpu().numpy()
    ref = np.asarray(ref, np.float64)
    assert a.shape == ref.shape, (a.shape, ref.shape)
    return np.abs(a - ref).max() / max(np.abs(ref).max(), 1e-6)


def check(a, ref, tol, what):
    e = rel_err(a, ref)
    assert e <= tol, f"{what}: rel err {e:.3e} > {tol:.1e}"


def itype_of(g):
    s = str(g.get("itype", "torch.float32"))
    return torch.bfloat16 if "bfloat16" in s else (torch.float16 if "float16" in s else torch.float32)


# =================================================================================================
# selective scan
# =================================================================================================
def run_scan(g, itype, oracle):
    from mamba_ssm.ops.selective_scan_interface import selective_scan_fn
    var_B, var_C = g["B"].ndim >= 3, g["C"].ndim >= 3
    u = G(g["u"], itype, True)
    delta = G(g["delta"], itype, True)
    A = G(g["A"], grad=True)
    B = G(g["B"], itype if var_B else torch.float32, True)
    C = G(g["C"], itype if var_C else torch.float32, True)
    D = G(g["D"], grad=True) if "D" in g else None
    z = G(g["z"], itype, True) if "z" in g else None
    bias = G(g["delta_bias"], grad=True) if "delta_bias" in g else None
    sp = bool(g["softplus"])
    out, last = selective_scan_fn(u, delta, A, B, C, D, z=z, delta_bias=bias, delta_softplus=sp,
                                  return_last_state=True)
    out.backward(G(g["g"], itype))
    got = dict(out=out, last_state=last, du=u.grad, ddelta=delta.grad, dA=A.grad, dB=B.grad, dC=C.grad,
               dD=D.grad if D is not None else None, dz=z.grad if z is not None else None,
               ddelta_bias=bias.grad if bias is not None else None)
    # oracle on exactly the (rounded) values the kernel saw
    f = lambda t: None if t is None else t.detach().float().cpu().numpy()
    o = oracle.scan_fwd(f(u), f(delta), f(A), f(B), f(C), f(D), f(z), f(bias), sp, prec="f64")
    ob = oracle.scan_bwd(f(u), f(delta), f(A), f(B), f(C), f(D), f(z), f(bias), g["g"], sp, prec="f64")
    want = dict(out=o["out_z"] if z is not None else o["out"], last_state=o["last_state"], **ob)
    return got, want


@pytest.mark.parametrize("name", golden_names("scan_"))
def test_scan_vs_oracle_and_golden(oracle, name):
    g = load_golden(name)
    itype = itype_of(g)
    tol = TOL[itype]
    got, want = run_scan(g, itype, oracle)
    for k in KIND:
        if want.get(k) is not None:
            check(got[k], want[k], tol_for(k, itype, "oracle"), f"{name}:{k} vs oracle")
            check(got[k], g[k], tol_for(k, itype, "golden"), f"{name}:{k} vs golden")


@pytest.mark.parametrize("itype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("shape", [(2, 8, 16, 64, 1), (1, 6, 16, 1024, 1), (2, 12, 16, 1025, 2), (1, 4, 8, 2048, 1),
                                   (2, 4, 16, 3000, 1), (1, 5, 4, 5000, 1), (1, 3, 32, 777, 3), (1, 2, 256, 96, 1)])
@pytest.mark.parametrize("has_z", [True, False])
def test_scan_random_vs_oracle(oracle, shape, itype, has_z):
    """Reference test recipe (test_selective_scan.py:53-88) at shapes covering every chunk /
    checkpoint boundary case, odd lengths (scalar I/O path), groups and large dstate."""
    batch, dim, N, L, groups = shape
    torch.manual_seed(0)
    g = dict(u=torch.randn(batch, dim, L), delta=0.5 * torch.rand(batch, dim, L), A=-0.5 * torch.rand(dim, N),
             B=torch.randn(batch, groups, N, L), C=torch.randn(batch, groups, N, L), D=torch.randn(dim),
             delta_bias=0.5 * torch.rand(dim), g=torch.randn(batch, dim, L), softplus=1)
    if has_z:
        g["z"] = torch.randn(batch, dim, L)
    g = {k: (v.numpy() if torch.is_tensor(v) else v) for k, v in g.items()}
    tol = TOL[itype]
    got, want = run_scan(g, itype, oracle)
    for k in ("out", "last_state", "du", "ddelta", "dB", "dC", "dz"):
        if want.get(k) is not None:
            check(got[k], want[k], tol * (2 if k != "out" else 1), f"{k}")
    for k in ("dA", "dD", "ddelta_bias"):
        check(got[k], want[k], tol * 5, k)


# ---- complex A (csrc/selective_scan_complex.hip; selective_scan.cpp:282-287, SSI:111-116, 144-145) ----------------------
def _cnp(t):
    """tensor -> numpy; complex tensors keep their dtype (the oracle tells constant from variable B / C by it)"""
    if t is None:
        return None
    t = t.detach().cpu()
    return t.numpy() if t.is_complex() else t.float().numpy()


def crel_err(a, ref):
    a = _cnp(a) if torch.is_tensor(a) else np.asarray(a)
    ref = _cnp(ref) if torch.is_tensor(ref) else np.asarray(ref)
    assert a.shape == ref.shape, (a.shape, ref.shape)
    return np.abs(a.astype(np.complex128) - ref.astype(np.complex128)).max() / max(np.abs(ref).max(), 1e-6)


def run_cscan(g, itype, oracle, reverse=False):
    """selective_scan_fn with a complex A on the GPU + the f64 oracle on the (rounded) values the kernel saw"""
    import vms_hip
    from mamba_ssm.ops.selective_scan_interface import selective_scan_fn
    var_B, var_C = not np.iscomplexobj(g["B"]), not np.iscomplexobj(g["C"])
    u, delta = G(g["u"], itype, True), G(g["delta"], itype, True)
    A = G(g["A"], torch.complex64, True)
    B = G(g["B"], itype if var_B else torch.complex64, True)
    C = G(g["C"], itype if var_C else torch.complex64, True)
    D = G(g["D"], grad=True) if "D" in g else None
    z = G(g["z"], itype, True) if "z" in g else None
    bias = G(g["delta_bias"], grad=True) if "delta_bias" in g else None
    sp = bool(g["softplus"])
    out, last = selective_scan_fn(u, delta, A, B, C, D, z=z, delta_bias=bias, delta_softplus=sp, return_last_state=True)
    assert vms_hip.last_kernel() == "scan_fwd_complex"
    out.backward(G(g["g"], itype))   # on an autograd thread: its kernel name is asserted where bwd is called directly
    got = dict(out=out, last_state=last, du=u.grad, ddelta=delta.grad, dA=A.grad, dB=B.grad, dC=C.grad,
               dD=D.grad if D is not None else None, dz=z.grad if z is not None else None,
               ddelta_bias=bias.grad if bias is not None else None)
    o = oracle.cscan_fwd(_cnp(u), _cnp(delta), _cnp(A), _cnp(B), _cnp(C), _cnp(D), _cnp(z), _cnp(bias), sp, prec="f64")
    ob = oracle.cscan_bwd(_cnp(u), _cnp(delta), _cnp(A), _cnp(B), _cnp(C), _cnp(D), _cnp(z), _cnp(bias), g["g"], sp, prec="f64")
    want = dict(out=o["out_z"] if z is not None else o["out"], last_state=o["last_state"], **ob)
    return got, want


def ccheck(got, want, itype, vs, name, var_B=True, var_C=True, scale=1.0):
    for k in KIND:
        if want.get(k) is None:
            continue
        kind = KIND[k]
        if (k == "dB" and not var_B) or (k == "dC" and not var_C):
            kind = "batch_sum"   # constant B / C: sums over batch x L like dA
        tol = TOL[itype] * FACTOR[kind][vs] * scale
        e = crel_err(got[k], want[k])
        assert e <= tol, f"{name}:{k} vs {vs}: rel err {e:.3e} > {tol:.1e}"


@pytest.mark.parametrize("name", golden_names("cscan_"))
def test_complex_scan_vs_oracle_and_golden(oracle, name):
    """The reference's complex-A recipe (test_selective_scan.py:53-88 with wtype = complex64): the HIP kernels against the
    f64 oracle and against the fixtures the reference's selective_scan_ref + autograd produced."""
    g = load_golden(name)
    itype = itype_of(g)
    var_B, var_C = not np.iscomplexobj(g["B"]), not np.iscomplexobj(g["C"])
    got, want = run_cscan(g, itype, oracle)
    ccheck(got, want, itype, "oracle", name, var_B, var_C)
    # 16-bit fixtures: the reference evaluated silu(z) in the input dtype (tests/test_oracle_golden.py): 2x
    ccheck(got, {k: g.get(k) for k in KIND}, itype, "golden", name, var_B, var_C, scale=2.0 if itype != torch.float32 else 1.0)


@pytest.mark.parametrize("itype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("shape", [(2, 8, 16, 64, 1), (1, 6, 16, 512, 1), (2, 12, 4, 513, 2), (1, 4, 8, 2048, 1),
                                   (2, 4, 16, 3001, 1), (1, 5, 3, 5000, 1), (1, 3, 33, 777, 3), (1, 2, 256, 96, 1), (1, 1, 1, 1, 1)])
@pytest.mark.parametrize("has_z", [True, False])
def test_complex_scan_random_vs_oracle(oracle, shape, itype, has_z):
    """every chunk (512) / checkpoint (1024, 2048) boundary case, odd lengths (element-wise I/O), groups, large dstate"""
    batch, dim, N, L, groups = shape
    torch.manual_seed(0)
    g = dict(u=torch.randn(batch, dim, L), delta=0.5 * torch.rand(batch, dim, L),
             A=torch.complex(-0.5 * torch.rand(dim, N), 3.0 * torch.randn(dim, N)),
             B=torch.randn(batch, groups, N, 2 * L), C=torch.randn(batch, groups, N, 2 * L), D=torch.randn(dim),
             delta_bias=0.5 * torch.rand(dim), g=torch.randn(batch, dim, L), softplus=1)
    if has_z:
        g["z"] = torch.randn(batch, dim, L)
    g = {k: (v.numpy() if torch.is_tensor(v) else v) for k, v in g.items()}
    got, want = run_cscan(g, itype, oracle)
    ccheck(got, want, itype, "oracle", str(shape))


@pytest.mark.parametrize("itype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("L", [300, 1536, 2055])
def test_complex_scan_reverse_and_accumulate(oracle, L, itype):
    """the extension flags the complex kernels keep: reverse == the op on flipped copies; out_z += / dz +="""
    import selective_scan_cuda
    torch.manual_seed(3)
    b, d, n = 2, 6, 8
    u, delta = torch.randn(b, d, L, device=DEV, dtype=itype), (0.5 * torch.rand(b, d, L, device=DEV)).to(itype)
    z, dout = torch.randn(b, d, L, device=DEV, dtype=itype), torch.randn(b, d, L, device=DEV, dtype=itype)
    A = torch.complex(-0.5 * torch.rand(d, n, device=DEV), torch.randn(d, n, device=DEV))
    B, C = torch.randn(b, 1, n, 2 * L, device=DEV, dtype=itype), torch.randn(b, 1, n, 2 * L, device=DEV, dtype=itype)
    D, bias = torch.randn(d, device=DEV), torch.rand(d, device=DEV)

    def flipc(t):   # flip the positions of interleaved (re, im) pairs
        return t.view(*t.shape[:-1], L, 2).flip(-2).reshape(t.shape).contiguous()

    fl = lambda t: t.flip(-1).contiguous()
    out, x, out_z = selective_scan_cuda.fwd(u, delta, A, B, C, D, z, bias, True, reverse=True)
    out_f, x_f, out_z_f = selective_scan_cuda.fwd(fl(u), fl(delta), A, flipc(B), flipc(C), D, fl(z), bias, True)
    assert torch.equal(out, fl(out_f)) and torch.equal(out_z, fl(out_z_f)) and torch.equal(x, x_f)
    r = selective_scan_cuda.bwd(u, delta, A, B, C, D, z, bias, dout, x, out, None, True, False, reverse=True)
    import vms_hip
    assert vms_hip.last_kernel() == "scan_bwd_complex"
    rf = selective_scan_cuda.bwd(fl(u), fl(delta), A, flipc(B), flipc(C), D, fl(z), bias, fl(dout), x_f, out_f, None, True, False)
    tol = TOL[itype]
    for i, name in enumerate(("du", "ddelta", "dA", "dB", "dC", "dD", "ddelta_bias", "dz")):
        want = rf[i] if name in ("dA", "dD", "ddelta_bias") else (flipc(rf[i]) if name in ("dB", "dC") else fl(rf[i]))
        # identical arithmetic per row; the atomics' order differs for the sums
        assert crel_err(r[i], want) <= (0 if name in ("du", "ddelta", "dz") else tol), name
    # accumulate flags
    base = torch.randn_like(out_z)
    acc = base.clone()
    selective_scan_cuda.fwd(u, delta, A, B, C, D, z, bias, True, out_z_into=acc)
    out2, _, oz2 = selective_scan_cuda.fwd(u, delta, A, B, C, D, z, bias, True)
    assert crel_err(acc, base.float() + oz2.float()) <= tol
    dzb = torch.randn_like(z)
    dz_acc = dzb.clone()
    x2 = selective_scan_cuda.fwd(u, delta, A, B, C, D, z, bias, True)[1]
    selective_scan_cuda.bwd(u, delta, A, B, C, D, z, bias, dout, x2, out2, dz_acc, True, False, accumulate_dz=True)
    r2 = selective_scan_cuda.bwd(u, delta, A, B, C, D, z, bias, dout, x2, out2, None, True, False)
    assert crel_err(dz_acc, dzb.float() + r2[7].float()) <= tol


@pytest.mark.parametrize("itype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("L", [300, 2055])
def test_complex_scan_reverse_from(oracle, L, itype):
    """reverse_from with a complex A (round 4): batch entries >= reverse_from right-to-left, the others left-to-right, in one launch
    == the two sub-batches as separate calls (identical arithmetic per row; the batch sums within the bar)."""
    import selective_scan_cuda
    torch.manual_seed(5)
    b, d, n, rf = 3, 6, 8, 1
    u, delta = torch.randn(b, d, L, device=DEV, dtype=itype), (0.5 * torch.rand(b, d, L, device=DEV)).to(itype)
    z, dout = torch.randn(b, d, L, device=DEV, dtype=itype), torch.randn(b, d, L, device=DEV, dtype=itype)
    A = torch.complex(-0.5 * torch.rand(d, n, device=DEV), torch.randn(d, n, device=DEV))
    B, C = torch.randn(b, 1, n, 2 * L, device=DEV, dtype=itype), torch.randn(b, 1, n, 2 * L, device=DEV, dtype=itype)
    D, bias = torch.randn(d, device=DEV), torch.rand(d, device=DEV)
    out, x, out_z = selective_scan_cuda.fwd(u, delta, A, B, C, D, z, bias, True, reverse_from=rf)
    lo, hi = slice(0, rf), slice(rf, b)
    parts = [selective_scan_cuda.fwd(u[s], delta[s], A, B[s], C[s], D, z[s], bias, True, reverse=rv) for s, rv in ((lo, False), (hi, True))]
    assert torch.equal(out, torch.cat([parts[0][0], parts[1][0]])) and torch.equal(out_z, torch.cat([parts[0][2], parts[1][2]]))
    assert torch.equal(x, torch.cat([parts[0][1], parts[1][1]]))
    r = selective_scan_cuda.bwd(u, delta, A, B, C, D, z, bias, dout, x, out, None, True, False, reverse_from=rf)
    rs = [selective_scan_cuda.bwd(u[s], delta[s], A, B[s], C[s], D, z[s], bias, dout[s], p[1], p[0], None, True, False, reverse=rv)
          for s, rv, p in ((lo, False, parts[0]), (hi, True, parts[1]))]
    tol = TOL[itype]
    for i, name in enumerate(("du", "ddelta", "dA", "dB", "dC", "dD", "ddelta_bias", "dz")):
        if name in ("dA", "dD", "ddelta_bias"):
            assert crel_err(r[i], rs[0][i] + rs[1][i]) <= tol, name
        else:
            want = torch.cat([rs[0][i], rs[1][i]])
            assert crel_err(r[i], want) <= (0 if name in ("du", "ddelta", "dz") else tol), name


@pytest.mark.parametrize("L", [1024, 2600, 4096 + 512])
def test_complex_scan_checkpoint_slots(oracle, L):
    """x of a complex scan: slot [c][2n + 1] = the state after 2048-chunk c (the reference's slot, last_state = x[:, :, -1, 1::2]),
    slot [c][2n] = the state after the chunk's first 1024 elements, and behind the reference-shaped view the state after every
    512 elements (what bwd starts from) -- all against the oracle run on truncated sequences."""
    import selective_scan_cuda
    torch.manual_seed(5)
    b, d, n = 1, 3, 4
    u, delta = torch.randn(b, d, L, device=DEV), 0.5 * torch.rand(b, d, L, device=DEV)
    A = torch.complex(-0.5 * torch.rand(d, n, device=DEV), torch.randn(d, n, device=DEV))
    B, C = torch.randn(b, 1, n, 2 * L, device=DEV), torch.randn(b, 1, n, 2 * L, device=DEV)
    out, x = selective_scan_cuda.fwd(u, delta, A, B, C, None, None, None, True)
    o = oracle.cscan_fwd(_cnp(u), _cnp(delta), _cnp(A), _cnp(B), _cnp(C), None, None, None, True, prec="f64")
    assert crel_err(out, o["out"]) <= 1e-3
    assert crel_err(x, o["x"]) <= 1e-3          # both reference-shaped slots of every chunk
    # the 512-element checkpoints: x is the (.., 2n) view of a (.., 6n) buffer
    full = torch.as_strided(x, (b, d, x.shape[2], 6 * n), x.stride())
    for k in range((L + 511) // 512):
        cut = min(L, 512 * (k + 1))
        t = oracle.cscan_fwd(_cnp(u[..., :cut]), _cnp(delta[..., :cut]), _cnp(A), _cnp(B[..., :2 * cut]), _cnp(C[..., :2 * cut]),
                             None, None, None, True, prec="f64")
        got = full[:, :, k // 4, 2 * n + (k % 4) * n: 2 * n + (k % 4 + 1) * n]
        assert crel_err(got, t["last_state"]) <= 1e-3, k


def test_complex_scan_extension_checks():
    """shape / dtype checks of the complex instantiations (selective_scan.cpp:240, 270, 276) and what this build declines"""
    import selective_scan_cuda
    u = torch.randn(1, 4, 16, device=DEV)
    A = torch.randn(4, 8, device=DEV, dtype=torch.complex64)
    B = torch.randn(1, 1, 8, 32, device=DEV)
    out, x = selective_scan_cuda.fwd(u, u, A, B, B, None, None, None, False)
    assert x.dtype == torch.complex64 and tuple(x.shape) == (1, 4, 1, 16) and out.shape == u.shape
    with pytest.raises(RuntimeError):   # variable B of a complex A must be (.., 2 * seqlen)
        selective_scan_cuda.fwd(u, u, A, B[..., :16].contiguous(), B, None, None, None, False)
    with pytest.raises(RuntimeError):   # constant B in the weight type
        selective_scan_cuda.fwd(u, u, A, torch.randn(4, 8, device=DEV), B, None, None, None, False)
    with pytest.raises(RuntimeError):   # reverse_from beyond the batch
        selective_scan_cuda.fwd(u, u, A, B, B, None, None, None, False, reverse_from=2)
    L = 1024
    u2, B2 = torch.randn(1, 4, L, device=DEV), torch.randn(1, 1, 8, 2 * L, device=DEV)
    with pytest.raises(RuntimeError):   # the backward needs the forward's checkpoints beyond one 512-element chunk
        selective_scan_cuda.bwd(u2, u2, A, B2, B2, None, None, None, u2, None, None, None, False, False)



@pytest.mark.parametrize("itype", [torch.bfloat16, torch.float32, torch.float16])
@pytest.mark.parametrize("shape", [(2, 64, 128, 1), (1, 128, 136, 2), (2, 64, 1024, 1), (1, 64, 2056, 1),
                                   (1, 192, 4096, 1), (1, 70, 1160, 1)])
@pytest.mark.parametrize("has_z", [True, False])
def test_scan_bwd_fast_path_vs_oracle(oracle, shape, itype, has_z, monkeypatch):
    """Shapes that qualify for the MFMA backward kernel (dstate 16, seqlen % 8 == 0, >= 64 dims per
    group; (1, 70, ..) has a partially filled last workgroup): parity with the oracle, and agreement
    with the generic kernel (VMS_FORCE_GENERIC=1) on the same inputs."""
    batch, dim, L, groups = shape
    N = 16
    torch.manual_seed(0)
    g = dict(u=torch.randn(batch, dim, L), delta=0.5 * torch.rand(batch, dim, L), A=-0.5 * torch.rand(dim, N),
             B=torch.randn(batch, groups, N, L), C=torch.randn(batch, groups, N, L), D=torch.randn(dim),
             delta_bias=0.5 * torch.rand(dim), g=torch.randn(batch, dim, L), softplus=1)
    if has_z:
        g["z"] = torch.randn(batch, dim, L)
    g = {k: (v.numpy() if torch.is_tensor(v) else v) for k, v in g.items()}
    tol = TOL[itype]
    got, want = run_scan(g, itype, oracle)
    monkeypatch.setattr(_dbg(), "force_generic", True)
    got_gen, _ = run_scan(g, itype, oracle)
    monkeypatch.setattr(_dbg(), "force_generic", False)
    for k in ("out", "last_state", "du", "ddelta", "dB", "dC", "dz"):
        if want.get(k) is not None:
            check(got[k], want[k], tol * (2 if k != "out" else 1), f"{k} vs oracle")
            check(got[k], got_gen[k].detach().float().cpu().numpy(), tol * 2, f"{k} fast vs generic kernel")
    for k in ("dA", "dD", "ddelta_bias"):
        check(got[k], want[k], tol * 5, k)


@pytest.mark.parametrize("itype", [torch.bfloat16, torch.float32, torch.float16])
@pytest.mark.parametrize("shape", [(2, 64, 1569, 1), (1, 128, 300, 2), (1, 64, 2049, 1), (2, 32, 17, 1), (1, 64, 1575, 1),
                                   (1, 96, 1031, 1)])
@pytest.mark.parametrize("has_z", [True, False])
def test_scan_ragged_lengths_vs_oracle(oracle, shape, itype, has_z, monkeypatch):
    """seqlen % 16 != 0 (e.g. 8 x 196 patches + 1 class token = 1569): the extension mirror zero-pads B / C
    (vms_hip.h bc_pad) and the paired kernels run with per-element masks and a scalar last vector; rows are not
    16-byte aligned either.  Parity with the oracle and agreement with the generic kernels."""
    batch, dim, L, groups = shape
    g = _rows_problem(shape, itype, has_z, seed=L)
    tol = TOL[itype]
    got, want = run_scan(g, itype, oracle)
    monkeypatch.setattr(_dbg(), "force_generic", True)
    got_gen, _ = run_scan(g, itype, oracle)
    monkeypatch.setattr(_dbg(), "force_generic", False)
    for k in ("out", "last_state", "du", "ddelta", "dB", "dC", "dz"):
        if want.get(k) is not None:
            check(got[k], want[k], tol * (2 if k != "out" else 1), f"{k} vs oracle")
            check(got[k], got_gen[k].detach().float().cpu().numpy(), tol * 2, f"{k} paired vs generic kernel")
    for k in ("dA", "dD", "ddelta_bias"):
        check(got[k], want[k], tol * 5, k)


ROWS_SHAPES = [(2, 64, 128, 1), (1, 128, 144, 2), (2, 64, 1024, 1), (1, 64, 2064, 1), (1, 192, 4096, 1),
               (3, 64, 1168, 1), (1, 256, 16, 1)]


def _rows_problem(shape, itype, has_z, seed=0):
    batch, dim, L, groups = shape
    N = 16
    torch.manual_seed(seed)
    g = dict(u=torch.randn(batch, dim, L), delta=0.5 * torch.rand(batch, dim, L), A=-0.5 * torch.rand(dim, N),
             B=torch.randn(batch, groups, N, L), C=torch.randn(batch, groups, N, L), D=torch.randn(dim),
             delta_bias=0.5 * torch.rand(dim), g=torch.randn(batch, dim, L), softplus=1)
    if has_z:
        g["z"] = torch.randn(batch, dim, L)
    return {k: (v.numpy() if torch.is_tensor(v) else v) for k, v in g.items()}


@pytest.mark.parametrize("itype", [torch.bfloat16, torch.float32, torch.float16])
@pytest.mark.parametrize("shape", ROWS_SHAPES)
@pytest.mark.parametrize("has_z", [True, False])
def test_scan_pair_path_vs_oracle_and_generic(oracle, shape, itype, has_z, monkeypatch):
    """dstate 16, variable B / C, (dim / groups) % 64 == 0, seqlen % 8 == 0 (the shape family of the row-major kernels of rounds
    1-3, removed in round 5): the paired kernels' parity with the oracle for outputs and all gradients, and agreement with
    the generic kernels on the same inputs."""
    g = _rows_problem(shape, itype, has_z)
    tol = TOL[itype]
    got, want = run_scan(g, itype, oracle)
    monkeypatch.setattr(_dbg(), "scan_impl", "generic")
    got_gen, _ = run_scan(g, itype, oracle)
    for k in ("out", "last_state", "du", "ddelta", "dB", "dC", "dz"):
        if want.get(k) is not None:
            check(got[k], want[k], tol * (2 if k != "out" else 1), f"{k} vs oracle")
            check(got[k], got_gen[k].detach().float().cpu().numpy(), tol * 2, f"{k} paired vs generic kernel")
    for k in ("dA", "dD", "ddelta_bias"):
        check(got[k], want[k], tol * 5, k)


@pytest.mark.parametrize("impl", ["pair", "generic"])
@pytest.mark.parametrize("itype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("shape", [(2, 64, 1024, 1), (1, 128, 2192, 2), (1, 64, 144, 1), (2, 64, 1569, 1)])
def test_scan_fwd_reverse_equals_flipped(shape, itype, impl, monkeypatch):
    """reverse=True == flip(fwd(flip(every seqlen-indexed tensor))) for every forward implementation."""
    import selective_scan_cuda
    monkeypatch.setattr(_dbg(), "scan_impl", impl)
    g = _rows_problem(shape, itype, True, seed=3)
    f = lambda k, dt=itype: G(g[k], dt)
    u, dl, A, B, C, D, z, bias = (f("u"), f("delta"), f("A", torch.float32), f("B"), f("C"), f("D", torch.float32),
                                  f("z"), f("delta_bias", torch.float32))
    out_r, x_r, oz_r = selective_scan_cuda.fwd(u, dl, A, B, C, D, z, bias, True, True)
    fl = lambda t: t.flip(-1).contiguous()
    out_f, x_f, oz_f = selective_scan_cuda.fwd(fl(u), fl(dl), A, fl(B), fl(C), D, fl(z), bias, True)
    tol = 1e-6 if itype == torch.float32 else 1e-2
    check(out_r, out_f.flip(-1).float().cpu().numpy(), tol, "out")
    check(oz_r, oz_f.flip(-1).float().cpu().numpy(), tol, "out_z")
    check(x_r[:, :, -1, 1::2], x_f[:, :, -1, 1::2].float().cpu().numpy(), 1e-5, "last_state")


@pytest.mark.parametrize("impl", ["pair", "generic"])
@pytest.mark.parametrize("itype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("shape", [(2, 64, 1024, 1), (1, 128, 2192, 2), (1, 6, 300, 1), (2, 64, 1569, 1), (1, 32, 303, 1)])
def test_scan_bwd_reverse_equals_flipped(shape, itype, impl, monkeypatch):
    """bwd(reverse=True) == the causal backward on flipped copies, gradients flipped back."""
    import selective_scan_cuda
    monkeypatch.setattr(_dbg(), "scan_impl", impl)
    g = _rows_problem(shape, itype, True, seed=7)
    f = lambda k, dt=itype: G(g[k], dt)
    u, dl, A, B, C, D, z, bias, dout = (f("u"), f("delta"), f("A", torch.float32), f("B"), f("C"),
                                        f("D", torch.float32), f("z"), f("delta_bias", torch.float32), f("g"))
    fl = lambda t: t.flip(-1).contiguous()
    out_r, x_r, _ = selective_scan_cuda.fwd(u, dl, A, B, C, D, z, bias, True, True)
    res_r = selective_scan_cuda.bwd(u, dl, A, B, C, D, z, bias, dout, x_r, out_r, None, True, True, True)
    out_f, x_f, _ = selective_scan_cuda.fwd(fl(u), fl(dl), A, fl(B), fl(C), D, fl(z), bias, True)
    res_f = selective_scan_cuda.bwd(fl(u), fl(dl), A, fl(B), fl(C), D, fl(z), bias, fl(dout), x_f, out_f, None, True, True)
    names = ("du", "ddelta", "dA", "dB", "dC", "dD", "ddelta_bias", "dz", "out_z")
    tol = 2e-5 if itype == torch.float32 else 2e-2
    for name, a, b_ in zip(names, res_r, res_f):
        ref = b_.flip(-1) if b_.dim() >= 3 else b_
        wide = 20 if name in ("dA", "dD", "ddelta_bias") else 1  # atomics: summation order differs
        check(a, ref.float().cpu().numpy(), tol * wide, name)


@pytest.mark.parametrize("itype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("width", [2, 3, 4])
@pytest.mark.parametrize("seqlen", [8, 151, 512, 1134])
def test_conv_reverse_equals_flipped(seqlen, width, itype):
    """reverse=True (anti-causal) == flip(conv(flip(x))), forward and backward, vector and scalar paths."""
    import causal_conv1d_cuda
    torch.manual_seed(seqlen + width)
    b, d = 2, 40
    x = torch.randn(b, d, seqlen, device=DEV).to(itype)
    w = torch.randn(d, width, device=DEV)
    bias = torch.randn(d, device=DEV)
    dout = torch.randn(b, d, seqlen, device=DEV).to(itype)
    fl = lambda t: t.flip(-1).contiguous()
    y_r = causal_conv1d_cuda.causal_conv1d_fwd(x, w, bias, True, True)
    y_f = causal_conv1d_cuda.causal_conv1d_fwd(fl(x), w, bias, True).flip(-1)
    assert torch.equal(y_r, y_f)
    dx_r, dw_r, db_r = causal_conv1d_cuda.causal_conv1d_bwd(x, w, bias, dout, None, True, True)
    dx_f, dw_f, db_f = causal_conv1d_cuda.causal_conv1d_bwd(fl(x), w, bias, fl(dout), None, True)
    assert torch.equal(dx_r, dx_f.flip(-1))
    check(dw_r, dw_f.cpu().numpy(), 1e-4, "dweight")
    check(db_r, db_f.cpu().numpy(), 1e-4, "dbias")


@pytest.mark.parametrize("itype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("has_z", [True, False])
@pytest.mark.parametrize("reverse", [False, True])
@pytest.mark.parametrize("shape,segments", [((2, 8, 4096, 1), "2"), ((1, 4, 5136, 1), "6"), ((1, 16, 65536, 1), None),
                                            ((2, 4, 3072, 1), "16"),
                                            # the LDS kernel's four-state tail form in the last range (5184 = 5 x 1024 + 64, 2304 = 2 x 1024 + 256)
                                            ((1, 8, 5184, 1), "3"), ((2, 16, 2304, 1), "3")])
def test_scan_fwd_sequence_split_equals_unsplit(shape, segments, itype, has_z, reverse, monkeypatch):
    """Few rows, long sequences: the forward runs as n ranges of 1024-element chunks per row, each started from the
    state the (P, q) pairs of scan_fwd_carry_kernel give (VMS_FWD_SEGMENTS forces a count; None = the kernel's own
    choice, 16 here).  out, out_z and every checkpoint equal the unsplit kernel's, and the backward runs from them."""
    import selective_scan_cuda
    g = _rows_problem(shape, itype, has_z, seed=shape[2])
    f = lambda k, dt=itype: G(g[k], dt)
    u, dl, A, B, C, D, bias, dout = (f("u"), f("delta"), f("A", torch.float32), f("B"), f("C"), f("D", torch.float32),
                                     f("delta_bias", torch.float32), f("g"))
    z = f("z") if has_z else None
    monkeypatch.setattr(_dbg(), "fwd_segments", int("1"))
    plain = selective_scan_cuda.fwd(u, dl, A, B, C, D, z, bias, True, reverse)
    if segments is None:
        monkeypatch.setattr(_dbg(), "fwd_segments", 0)
    else:
        monkeypatch.setattr(_dbg(), "fwd_segments", int(segments))
    split = selective_scan_cuda.fwd(u, dl, A, B, C, D, z, bias, True, reverse)
    tol = 2e-5 if itype == torch.float32 else 1e-2
    for name, a, b_ in zip(("out", "x", "out_z"), split, plain):
        check(a, b_.float().cpu().numpy(), tol, name)
    # the finer checkpoints behind the reference-shaped view feed the backward: same gradients from either forward
    ga = selective_scan_cuda.bwd(u, dl, A, B, C, D, z, bias, dout, split[1], split[0] if has_z else None, None, True, False, reverse)
    gb = selective_scan_cuda.bwd(u, dl, A, B, C, D, z, bias, dout, plain[1], plain[0] if has_z else None, None, True, False, reverse)
    for name, a, b_ in zip(("du", "ddelta", "dA", "dB", "dC"), ga, gb):
        check(a, b_.float().cpu().numpy(), tol * (20 if name == "dA" else 2), name)


@pytest.mark.parametrize("itype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("has_z", [True, False])
@pytest.mark.parametrize("reverse", [False, True])
@pytest.mark.parametrize("shape,segments", [((2, 64, 2048, 1), "3"), ((1, 32, 1160, 1), "5"), ((1, 64, 16384, 1), None),
                                            ((2, 32, 384, 1), "16")])
def test_scan_bwd_sequence_split_equals_unsplit(shape, segments, itype, has_z, reverse, monkeypatch):
    """Few rows, long sequences: the backward runs as n ranges of chunks per row, chained by the (P, q) adjoint carries
    of scan_bwd_carry_kernel (VMS_BWD_SEGMENTS forces a count; None = the kernel's own choice, 16 here).  Every result
    equals the unsplit kernel's."""
    import selective_scan_cuda
    g = _rows_problem(shape, itype, has_z, seed=shape[2])
    f = lambda k, dt=itype: G(g[k], dt)
    u, dl, A, B, C, D, bias, dout = (f("u"), f("delta"), f("A", torch.float32), f("B"), f("C"), f("D", torch.float32),
                                     f("delta_bias", torch.float32), f("g"))
    z = f("z") if has_z else None
    res = selective_scan_cuda.fwd(u, dl, A, B, C, D, z, bias, True, reverse)
    out, x = res[0], res[1]
    monkeypatch.setattr(_dbg(), "bwd_segments", int("1"))
    plain = selective_scan_cuda.bwd(u, dl, A, B, C, D, z, bias, dout, x, out if has_z else None, None, True, has_z, reverse)
    if segments is None:
        monkeypatch.setattr(_dbg(), "bwd_segments", 0)
    else:
        monkeypatch.setattr(_dbg(), "bwd_segments", int(segments))
    split = selective_scan_cuda.bwd(u, dl, A, B, C, D, z, bias, dout, x, out if has_z else None, None, True, has_z, reverse)
    names = ("du", "ddelta", "dA", "dB", "dC", "dD", "ddelta_bias", "dz", "out_z")
    tol = 2e-5 if itype == torch.float32 else 1e-2
    for name, a, b_ in zip(names, split, plain):
        wide = 20 if name in ("dA", "dD", "ddelta_bias") else 1   # atomics: summation order differs
        check(a, b_.float().cpu().numpy(), tol * wide, name)


@pytest.mark.parametrize("impl", ["pair", "generic"])
@pytest.mark.parametrize("itype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("reverse", [False, True])
def test_scan_bwd_accumulates_dz(itype, impl, reverse, monkeypatch):
    """accumulate_dz (vms_hip.h dz_accumulate): dz += this call's gradient, in every backward kernel."""
    import selective_scan_cuda
    monkeypatch.setattr(_dbg(), "scan_impl", impl)
    g = _rows_problem((2, 64, 1160, 1), itype, True, seed=11)
    f = lambda k, dt=itype: G(g[k], dt)
    u, dl, A, B, C, D, z, bias, dout = (f("u"), f("delta"), f("A", torch.float32), f("B"), f("C"),
                                        f("D", torch.float32), f("z"), f("delta_bias", torch.float32), f("g"))
    out, x, _ = selective_scan_cuda.fwd(u, dl, A, B, C, D, z, bias, True, reverse)
    plain = selective_scan_cuda.bwd(u, dl, A, B, C, D, z, bias, dout, x, out, None, True, False, reverse)
    start = torch.randn_like(z)
    dz = start.clone()
    acc = selective_scan_cuda.bwd(u, dl, A, B, C, D, z, bias, dout, x, out, dz, True, False, reverse,
                                  accumulate_dz=True)
    assert acc[7].data_ptr() == dz.data_ptr()
    want = (start.float() + plain[7].float()).cpu().numpy()
    check(dz, want, 1e-6 if itype == torch.float32 else 1e-2, "dz accumulated")
    for name, a, b_ in zip(("du", "ddelta"), acc, plain):
        assert torch.equal(a, b_), name
    with pytest.raises(RuntimeError):
        selective_scan_cuda.bwd(u, dl, A, B, C, D, z, bias, dout, x, out, None, True, False, reverse, accumulate_dz=True)


@pytest.mark.parametrize("impl", ["pair", "generic"])
@pytest.mark.parametrize("itype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("reverse", [False, True])
def test_scan_fwd_accumulates_out_z(itype, impl, reverse, monkeypatch):
    """out_z_into (vms_hip.h out_z_accumulate): out_z += this call's gated output; out and x are unchanged.  With
    the rows kernels selected the call is served by the next eligible kernel."""
    import selective_scan_cuda
    monkeypatch.setattr(_dbg(), "scan_impl", impl)
    g = _rows_problem((2, 64, 1168, 1), itype, True, seed=13)
    f = lambda k, dt=itype: G(g[k], dt)
    u, dl, A, B, C, D, z, bias = (f("u"), f("delta"), f("A", torch.float32), f("B"), f("C"),
                                  f("D", torch.float32), f("z"), f("delta_bias", torch.float32))
    out, x, oz = selective_scan_cuda.fwd(u, dl, A, B, C, D, z, bias, True, reverse)
    start = torch.randn_like(z)
    acc = start.clone()
    out2, x2, oz2 = selective_scan_cuda.fwd(u, dl, A, B, C, D, z, bias, True, reverse, out_z_into=acc)
    assert oz2.data_ptr() == acc.data_ptr()
    check(acc, (start.float() + oz.float()).cpu().numpy(), 1e-6 if itype == torch.float32 else 1e-2, "out_z accumulated")
    check(out2, out.float().cpu().numpy(), 1e-6 if itype == torch.float32 else 1e-2, "out")
    check(x2[..., 1::2], x[..., 1::2].cpu().numpy(), 1e-5, "checkpoints")


@pytest.mark.parametrize("itype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("layout", ["seq", "seq_ragged", "channel_last"])
@pytest.mark.parametrize("reverse", [False, True])
def test_conv_bwd_accumulates_dx(layout, itype, reverse):
    """accumulate_dx (vms_hip.h dx_accumulate): dx += this call's gradient (buffer-addressed, generic and
    channel-last kernels)."""
    import causal_conv1d_cuda
    if layout == "channel_last" and reverse:
        pytest.skip("reverse needs the seqlen-contiguous layout")
    torch.manual_seed(3)
    b, d, L = 2, 48, 1134 if layout == "seq_ragged" else 1136
    mk = lambda: torch.randn(b, d, L, device=DEV).to(itype)
    x, dout, start = mk(), mk(), mk()
    if layout == "channel_last":
        cl = lambda t: t.transpose(1, 2).contiguous().transpose(1, 2)
        x, dout, start = cl(x), cl(dout), cl(start)
    w, bias = torch.randn(d, 4, device=DEV), torch.randn(d, device=DEV)
    plain = causal_conv1d_cuda.causal_conv1d_bwd(x, w, bias, dout, None, True, reverse)
    dx = start.clone()  # keeps start's (possibly channel-last) strides
    acc = causal_conv1d_cuda.causal_conv1d_bwd(x, w, bias, dout, dx, True, reverse, accumulate_dx=True)
    assert acc[0].data_ptr() == dx.data_ptr()
    want = (start.float() + plain[0].float()).cpu().numpy()
    check(dx, want, 1e-6 if itype == torch.float32 else 1e-2, "dx accumulated")
    check(acc[1], plain[1].cpu().numpy(), 1e-5, "dweight")
    with pytest.raises(RuntimeError):
        causal_conv1d_cuda.causal_conv1d_bwd(x, w, bias, dout, None, True, reverse, accumulate_dx=True)


@pytest.mark.parametrize("itype", [torch.float32, torch.bfloat16])
def test_bidirectional_node_equals_two_nodes(itype):
    """BiMambaInnerFnNoOutProj (one node, dx / dz accumulated by the kernels) == the sum of a left-to-right and a
    right-to-left MambaInnerFnNoOutProj, values and every gradient."""
    import mamba_ssm.ops.selective_scan_interface as ssi
    torch.manual_seed(5)
    b, d, L, n, R = 2, 64, 1100 if itype == torch.float32 else 1104, 16, 4

    def params():
        return [torch.randn(d, 1, 4, device=DEV) * 0.3, torch.randn(d, device=DEV) * 0.1,
                torch.randn(R + 2 * n, d, device=DEV) * 0.1, torch.randn(d, R, device=DEV) * 0.3,
                -torch.rand(d, n, device=DEV) - 0.5, torch.randn(d, device=DEV), torch.randn(d, device=DEV) * 0.1]
    pf, pb = params(), params()
    xz0 = (torch.randn(b, 2 * d, L, device=DEV) * 0.5).to(itype)
    gout = torch.randn(b, d, L, device=DEV).to(itype)

    def run(fused):
        leaves = [t.clone().requires_grad_() for t in [xz0] + pf + pb]
        xz, f, bk = leaves[0], leaves[1:8], leaves[8:]
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=itype == torch.bfloat16):
            if fused:
                out = ssi.bimamba_inner_fn_no_out_proj(xz, tuple(f), tuple(bk))
            else:
                one = lambda q, rev: ssi.mamba_inner_fn_no_out_proj(xz, q[0], q[1], q[2], q[3], q[4], None, None, q[5],
                                                                    delta_bias=q[6], delta_softplus=True, reverse=rev)
                out = one(f, False) + one(bk, True)
        out.backward(gout)
        return out, [t.grad for t in leaves]
    out_a, g_a = run(True)
    out_b, g_b = run(False)
    tol = 1e-5 if itype == torch.float32 else 2e-2
    check(out_a, out_b.float().detach().cpu().numpy(), tol, "out")
    for i, (a, b_) in enumerate(zip(g_a, g_b)):
        check(a, b_.float().cpu().numpy(), tol * (1 if i == 0 else 5), f"grad {i}")


# 2064 / 2304 / 1280 / 256 / 64: a last chunk of <= 256 elements -- the LDS forward's four-states-at-a-time tail form; 2512 / 1296: a
# longer one (whole-wave pass)
@pytest.mark.parametrize("L", [2512, 2504, 8, 2064, 2304, 1280, 1296, 256, 64])
@pytest.mark.parametrize("reverse", [False, True])
def test_scan_lane_checkpoints(oracle, L, reverse):
    """ABI v7, x_has_sub == 3: where the whole-vector backward kernel takes the problem, the forward (the LDS kernel at
    L % 16 == 0, the per-wave kernel otherwise) leaves the state after every 8 elements behind the reference-shaped x, and the
    backward starts every lane from them.  Checkpoints vs the oracle's last_state of the truncated problem; gradients vs the oracle."""
    import selective_scan_cuda
    import vms_hip
    torch.manual_seed(3)
    b, d, N = 2, 32, 16
    u = torch.randn(b, d, L, device=DEV)
    z = torch.randn(b, d, L, device=DEV)
    delta = 0.5 * torch.rand(b, d, L, device=DEV)
    A = -0.5 * torch.rand(d, N, device=DEV)
    B = torch.randn(b, 1, N, L, device=DEV)
    C = torch.randn(b, 1, N, L, device=DEV)
    D = torch.randn(d, device=DEV)
    bias = 0.5 * torch.rand(d, device=DEV)
    nch = (L + 2047) // 2048
    if L % 16 == 0:
        out, x, out_z = selective_scan_cuda.fwd(u, delta, A, B, C, D, z, bias, True, reverse=reverse)
    else:
        # vms_scan_x_pitch() keeps the 128-element layout at these lengths (the per-wave forward kernel writes the checkpoints 4
        # bytes at a time); a caller of the C ABI may still hand over an x with room for them: every forward kernel fills it
        assert selective_scan_cuda.fwd(u, delta, A, B, C, D, z, bias, True, reverse=reverse)[1].stride(2) == 18 * N
        x = torch.empty(b, d, nch, 258 * N, device=DEV)[..., :2 * N]
        out, out_z = torch.empty_like(delta), torch.empty_like(z)
        Bp, Cp, pad = selective_scan_cuda.pad_bc(B, C, reverse, False)
        vms_hip.scan_fwd(u, delta, A, Bp, Cp, D, z, bias, out, out_z, x, True, reverse, False, pad)
    assert x.shape == (b, d, nch, 2 * N) and x.stride(2) == 258 * N and vms_hip.x_layout_of(x, N) == 3
    assert vms_hip.last_kernel().startswith("scan_fwd_pair_lds" if L % 16 == 0 else "scan_fwd_pair")
    xfull = x.as_strided((b, d, nch, 258 * N), (d * nch * 258 * N, nch * 258 * N, 258 * N, 1))
    f = lambda t: t.detach().float().cpu().numpy()
    lf = (lambda t: t.flip(-1)) if reverse else (lambda t: t)     # x is in scan order
    for cut in sorted({8, 16, 128, 1024, 1032, 2048, 2056, 2496, L} & set(range(8, L + 1, 8))):
        t = oracle.scan_fwd(f(lf(u)[..., :cut]), f(lf(delta)[..., :cut]), f(A), f(lf(B)[..., :cut]), f(lf(C)[..., :cut]), f(D),
                            f(lf(z)[..., :cut]), f(bias), True, prec="f64")
        i8 = cut // 8 - 1
        c, i = i8 // 256, i8 % 256
        got = torch.stack([xfull[:, :, c, 2 * N + ((n // 4) * 256 + i) * 4 + n % 4] for n in range(N)], dim=-1)
        check(got, t["last_state"], 1e-3, f"state after {cut} elements")
    o = oracle.scan_fwd(f(lf(u)), f(lf(delta)), f(A), f(lf(B)), f(lf(C)), f(D), f(lf(z)), f(bias), True, prec="f64")
    check(lf(out_z), o["out_z"], 1e-3, "out_z")
    check(x, o["x"], 1e-3, "x (mid + end slots)")
    dout = torch.randn(b, d, L, device=DEV)
    res = selective_scan_cuda.bwd(u, delta, A, B, C, D, z, bias, dout, x, out, None, True, False, reverse=reverse)
    assert vms_hip.last_kernel().startswith("scan_bwd_pair4")
    ob = oracle.scan_bwd(f(lf(u)), f(lf(delta)), f(A), f(lf(B)), f(lf(C)), f(D), f(lf(z)), f(bias), f(lf(dout)), True, prec="f64")
    names = ("du", "ddelta", "dA", "dB", "dC", "dD", "ddelta_bias", "dz")
    for name, got in zip(names, res):
        want = ob[name]
        g = lf(got) if got.ndim >= 3 and got.shape[-1] == L else got
        check(g, want.reshape(g.shape), 2e-3, f"{name} (lane checkpoints, reverse={reverse})")
    # the same backward from the 128-element layout: x_has_sub == 1 (VMS_X_LAYOUT=1) gives the same gradients
    out1, x1, _ = selective_scan_cuda.fwd(u, delta, A, B, C, D, z, bias, True, reverse=reverse, for_backward=False)
    assert x1.stride(2) == 18 * N
    res1 = selective_scan_cuda.bwd(u, delta, A, B, C, D, z, bias, dout, x1, out1, None, True, False, reverse=reverse)
    for name, a, c_ in zip(names, res, res1):
        check(a, f(c_), 1e-4, f"{name}: lane checkpoints vs 128-element checkpoints")


def test_scan_strided_views_and_checkpoints(oracle):
    """u/z are channel halves of one xz buffer, delta is d-slowest, out inherits delta's layout,
    dz is written into a slice of a pre-allocated dxz (SSI:175, 182, 244-248); raw extension ABI."""
    import selective_scan_cuda
    torch.manual_seed(1)
    b, d, N, L = 2, 16, 16, 2500
    xz = torch.randn(b, 2 * d, L, device=DEV)
    u, z = xz[:, :d], xz[:, d:]
    delta = (0.5 * torch.rand(d, b, L, device=DEV)).permute(1, 0, 2)
    A = -0.5 * torch.rand(d, N, device=DEV)
    B = torch.randn(b, 1, N, L, device=DEV)
    C = torch.randn(b, 1, N, L, device=DEV)
    D = torch.randn(d, device=DEV)
    bias = 0.5 * torch.rand(d, device=DEV)
    out, x, out_z = selective_scan_cuda.fwd(u, delta, A, B, C, D, z, bias, True)
    assert out.stride() == delta.stride() and x.shape == (b, d, 2, 2 * N)
    # the 128-element sub-checkpoints ride behind the reference-shaped view (include/vms_hip.h)
    xfull = x.as_strided((b, d, 2, 18 * N), (d * 2 * 18 * N, 2 * 18 * N, 18 * N, 1))
    for cut in (128, 1024, 2048, 2176):
        t = oracle.scan_fwd(*(a[..., :cut] if a.ndim >= 3 and a.shape[-1] == L else a
                              for a in map(lambda t_: t_.detach().float().cpu().numpy(), (u, delta, A, B, C))),
                            D.cpu().numpy(), z[..., :cut].cpu().numpy(), bias.cpu().numpy(), True, prec="f64")
        e = cut // 128 - 1
        check(xfull[:, :, e // 16, 2 * N + (e % 16) * N: 2 * N + (e % 16 + 1) * N], t["last_state"], 1e-3,
              f"sub-checkpoint after {cut} elements")
    f = lambda t: t.detach().float().cpu().numpy()
    o = oracle.scan_fwd(f(u), f(delta), f(A), f(B), f(C), f(D), f(z), f(bias), True, prec="f64")
    check(out, o["out"], 1e-3, "out")
    check(out_z, o["out_z"], 1e-3, "out_z")
    check(x, o["x"], 1e-3, "x checkpoints (mid + end slots)")
    dout = torch.randn(b, d, L, device=DEV)
    dxz = torch.full_like(xz, float("nan"))
    dz_view = dxz[:, d:]
    res = selective_scan_cuda.bwd(u, delta, A, B, C, D, z, bias, dout, x, out, dz_view, True, True)
    du, ddelta, dA, dB, dC, dD, dbias, dz, out_z2 = res
    assert dz.data_ptr() == dz_view.data_ptr() and ddelta.stride() == delta.stride()
    assert torch.isnan(dxz[:, :d]).all() and not torch.isnan(dxz[:, d:]).any()
    ob = oracle.scan_bwd(f(u), f(delta), f(A), f(B), f(C), f(D), f(z), f(bias), f(dout), True, prec="f64")
    for k, v in (("du", du), ("ddelta", ddelta), ("dB", dB), ("dC", dC), ("dz", dz)):
        check(v, ob[k], 2e-3, k)
    for k, v in (("dA", dA), ("dD", dD), ("ddelta_bias", dbias)):
        check(v, ob[k], 5e-3, k)
    check(out_z2, o["out_z"], 1e-3, "recomputed out_z")


def test_scan_extension_error_behaviour():
    import selective_scan_cuda
    u = torch.randn(1, 4, 16, device=DEV)
    A = -torch.rand(4, 8, device=DEV)
    B = torch.randn(1, 1, 8, 16, device=DEV)
    with pytest.raises(RuntimeError):  # dtype mismatch (selective_scan.cpp:242)
        selective_scan_cuda.fwd(u, u.half(), A, B, B, None, None, None, False)
    with pytest.raises(RuntimeError):  # dstate > 256 (:265)
        selective_scan_cuda.fwd(u, u, -torch.rand(4, 300, device=DEV), torch.randn(1, 1, 300, 16, device=DEV),
                                torch.randn(1, 1, 300, 16, device=DEV), None, None, None, False)
    with pytest.raises(RuntimeError):  # x required when n_chunks > 1 (:449)
        L = 4096
        u2 = torch.randn(1, 4, L, device=DEV)
        B2 = torch.randn(1, 1, 8, L, device=DEV)
        selective_scan_cuda.bwd(u2, u2, A, B2, B2, None, None, None, u2, None, None, None, False, False)


@pytest.mark.parametrize("L", [2048, 1031])
def test_scan_bwd_softplus_derivative_for_strongly_negative_delta(oracle, L):
    """fp32: delta + bias in [-14, -5], where softplus' = sigmoid is 1e-6 .. 7e-3.  The derivative must come from the raw
    pre-activation (e / (1 + e)) as in the reference (selective_scan_bwd_kernel.cuh:439-452); rebuilding it as
    1 - exp(-softplus) subtracts two numbers next to 1 and loses 2 .. 5 digits there.  ddelta and ddelta_bias
    are compared RELATIVE TO THEIR OWN SCALE, so the small derivatives are what is tested."""
    import selective_scan_cuda
    import vms_hip
    b, d, N = 2, 32, 16
    rng = np.random.default_rng(3)
    f = lambda *s: rng.standard_normal(s).astype(np.float32)
    u, z, g = f(b, d, L), f(b, d, L), f(b, d, L)
    delta = (-9.0 + 4.0 * rng.random((b, d, L))).astype(np.float32)          # raw delta
    bias = (-1.0 + rng.random(d)).astype(np.float32)                          # delta + bias in [-14, -5]
    A = (-0.5 - rng.random((d, N))).astype(np.float32)
    Bm, Cm, Dv = f(b, 1, N, L), f(b, 1, N, L), f(d)
    ref = oracle.scan_bwd(u, delta, A, Bm, Cm, Dv, z, bias, g, True, prec="f64")
    t = lambda a: torch.tensor(a, device=DEV)
    out, x, _ = selective_scan_cuda.fwd(t(u), t(delta), t(A), t(Bm), t(Cm), t(Dv), t(z), t(bias), True)
    got = selective_scan_cuda.bwd(t(u), t(delta), t(A), t(Bm), t(Cm), t(Dv), t(z), t(bias), t(g), x, out, None, True, False)
    assert vms_hip.last_kernel().startswith("scan_bwd_pair")
    check(got[1], ref["ddelta"], 1e-3, "ddelta at strongly negative delta + bias")
    check(got[6], ref["ddelta_bias"], 1e-3, "ddelta_bias at strongly negative delta + bias")
    # element-wise relative error where the reference is not tiny: the cancellation showed up as 1e-2 .. 1e-4 here
    r = ref["ddelta"]
    m = np.abs(r) > 1e-3 * np.abs(r).max()
    rel = np.abs(got[1].cpu().numpy() - r)[m] / np.abs(r)[m]
    assert rel.max() < 2e-3, f"element-wise relative error of ddelta {rel.max():.2e}"


def test_dispatch_is_visible_and_parameter_driven(monkeypatch):
    """ABI v4: the library reads no environment; the binding turns VMS_SCAN_IMPL / VMS_*_SEGMENTS into the impl /
    segments fields, and vms_last_kernel() names what ran -- so a declined fast path is visible to the caller."""
    import selective_scan_cuda
    import vms_hip
    g = _rows_problem((1, 64, 16384, 1), torch.bfloat16, True)
    f = lambda k, dt=torch.bfloat16: G(g[k], dt)
    args = (f("u"), f("delta"), f("A", torch.float32), f("B"), f("C"), f("D", torch.float32), f("z"),
            f("delta_bias", torch.float32), True)
    selective_scan_cuda.fwd(*args)
    assert vms_hip.last_kernel() == "scan_fwd_pair_lds+split"   # 64 rows: fewer waves than SIMDs -> ranges of chunks
    monkeypatch.setattr(_dbg(), "fwd_segments", int("1"))
    out, x, _ = selective_scan_cuda.fwd(*args)
    assert vms_hip.last_kernel() == "scan_fwd_pair_lds"         # the default kernel of whole-vector rows (B / C through LDS)
    monkeypatch.setattr(_dbg(), "bwd_segments", int("1"))
    selective_scan_cuda.bwd(*args[:8], f("g"), x, out, None, True, False)
    assert vms_hip.last_kernel() == "scan_bwd_pair4"            # ... and of the backward (4 rows per wave)
    monkeypatch.setattr(_dbg(), "bwd_segments", 0)
    selective_scan_cuda.bwd(*args[:8], f("g"), x, out, None, True, False)
    assert vms_hip.last_kernel() == "scan_bwd_pair4+split"      # 2 workgroups for 256 CUs: the kernel's own choice is to split
    monkeypatch.setattr(_dbg(), "scan_impl", "generic")
    out, x, _ = selective_scan_cuda.fwd(*args)
    assert vms_hip.last_kernel() == "scan_fwd_generic"
    selective_scan_cuda.bwd(*args[:8], f("g"), x, out, None, True, False)
    assert vms_hip.last_kernel() == "scan_bwd_generic"
    monkeypatch.setattr(_dbg(), "scan_impl", None)
    # dstate 8 (and 4) have their own instantiation of the LDS forward since round 6 (tests/test_small_dstate.py) ...
    A8 = -torch.rand(64, 8, device=DEV)
    B8 = torch.randn(1, 1, 8, 16384, device=DEV, dtype=torch.bfloat16)
    selective_scan_cuda.fwd(args[0], args[1], A8, B8, B8, None, None, None, True)
    assert vms_hip.last_kernel() == "scan_fwd_pair_lds_n"
    # ... dstate 12 is outside the fast paths: the generic kernels take it, and say so
    A12 = -torch.rand(64, 12, device=DEV)
    B12 = torch.randn(1, 1, 12, 16384, device=DEV, dtype=torch.bfloat16)
    selective_scan_cuda.fwd(args[0], args[1], A12, B12, B12, None, None, None, True)
    assert vms_hip.last_kernel() == "scan_fwd_generic"


@pytest.mark.parametrize("shape,fwd_name,bwd_name", [
    ((8, 8192, 1024), "scan_fwd_pair_lds", "scan_bwd_pair4"),          # BASELINE configs[1] (bench.py default)
    ((8, 3136, 768), "scan_fwd_pair_lds", "scan_bwd_pair4"),           # configs[2]
    ((2, 2304, 512), "scan_fwd_pair_lds", "scan_bwd_pair4"),           # configs[3], one direction of the DBM block
    ((8, 1569, 768), "scan_fwd_pair_ragged", "scan_bwd_pair_ragged"),  # 8 x 196 patches + class token
    # 12 channels per group: a forward workgroup's 8 rows would straddle groups -> the per-wave B / C kernel; the
    # backward's 32-row workgroups cannot share a group either -> generic
    ((2, 2048, 24, 2), "scan_fwd_pair", "scan_bwd_generic"),
    # 32 channels per group, not a multiple of 8 x ... : forward LDS kernel, paired backward
    ((2, 2048, 64, 2), "scan_fwd_pair_lds", "scan_bwd_pair4"),
])
def test_kernel_choice_per_shape(shape, fwd_name, bwd_name):
    """Which kernel a shape runs on is part of the contract the parity tests rely on: every BASELINE shape must land on
    the tuned kernels, and a shape that falls off them must say so through vms_last_kernel()."""
    import selective_scan_cuda
    import vms_hip
    b, L, d = shape[:3]
    groups = shape[3] if len(shape) > 3 else 1
    dt = torch.bfloat16
    torch.manual_seed(0)
    u = torch.randn(b, d, L, device=DEV, dtype=dt)
    delta = (0.5 * torch.rand(b, d, L, device=DEV)).to(dt)
    A = -torch.rand(d, 16, device=DEV)
    Bm = torch.randn(b, groups, 16, L, device=DEV, dtype=dt)
    Cm = torch.randn(b, groups, 16, L, device=DEV, dtype=dt)
    D = torch.randn(d, device=DEV)
    z = torch.randn(b, d, L, device=DEV, dtype=dt)
    bias = torch.rand(d, device=DEV)
    out, x, _ = selective_scan_cuda.fwd(u, delta, A, Bm, Cm, D, z, bias, True)
    assert vms_hip.last_kernel().split("+")[0] == fwd_name, vms_hip.last_kernel()
    selective_scan_cuda.bwd(u, delta, A, Bm, Cm, D, z, bias, torch.randn_like(u), x, out, None, True, False)
    assert vms_hip.last_kernel().split("+")[0] == bwd_name, vms_hip.last_kernel()
    torch.cuda.synchronize()


@pytest.mark.parametrize("b,L,d,N,fwd_name,bwd_name", [
    # the (model, shape) pairs of tools/suite_shapes.py (profiles/r06_suite_shapes.md) at the scan level: no suite shape on a generic
    # scan kernel or on the ragged generation; lengths as the mixers pad them (785 -> 800, 3137 -> 3152, 188 -> 192)
    (32, 800, 384, 4, "scan_fwd_pair_lds_n", "scan_bwd_pair4_n"),        # CLIP ViViM, d_state = 4 (model_clip.py:945-947), 4 frames
    (8, 3152, 384, 4, "scan_fwd_pair_lds_n", "scan_bwd_pair4_n"),        # ... 16 frames
    (8, 1584, 384, 16, "scan_fwd_pair_lds", "scan_bwd_pair4"),           # ViViM tiny, 8 frames + cls
    (8, 3152, 768, 16, "scan_fwd_pair_lds", "scan_bwd_pair4"),           # ViViM small, 16 frames
    (1568, 8, 768, 16, "scan_fwd_short", "scan_bwd_short"),              # TimeMamba along time, 8 frames
    (1568, 16, 768, 16, "scan_fwd_short", "scan_bwd_short"),             # ... 16 frames
    (8, 1568, 768, 16, "scan_fwd_pair_lds", "scan_bwd_pair4"),           # TimeMamba joint
    (1, 6000, 128, 16, "scan_fwd_pair_lds", "scan_bwd_pair4"),           # TAS, batch 1
    (1, 192, 1024, 16, "scan_fwd_pair_lds", "scan_bwd_pair4"),           # PDVC, 188 tokens padded
    (32, 112, 512, 16, "scan_fwd_pair_lds", "scan_bwd_pair4"),           # UniVTG, 107 tokens padded
    (16, 32, 2048, 16, "scan_fwd_short+segments", "scan_bwd_short+segments"),   # LSTR work memory: 17 .. 64 elements, many rows
    (16, 512, 2048, 16, "scan_fwd_pair_lds", "scan_bwd_pair4"),          # LSTR long memory
])
def test_kernel_choice_suite_shapes(b, L, d, N, fwd_name, bwd_name):
    import selective_scan_cuda
    import vms_hip
    dt = torch.bfloat16
    torch.manual_seed(0)
    u = torch.randn(b, d, L, device=DEV, dtype=dt)
    delta = (0.5 * torch.rand(b, d, L, device=DEV)).to(dt)
    A = -torch.rand(d, N, device=DEV)
    Bm = torch.randn(b, 1, N, L, device=DEV, dtype=dt)
    Cm = torch.randn(b, 1, N, L, device=DEV, dtype=dt)
    z = torch.randn(b, d, L, device=DEV, dtype=dt)
    out, x, _ = selective_scan_cuda.fwd(u, delta, A, Bm, Cm, None, z, None, True)
    got = vms_hip.last_kernel()
    assert (got if "segments" in fwd_name else got.split("+")[0]) == fwd_name, got
    selective_scan_cuda.bwd(u, delta, A, Bm, Cm, None, z, None, torch.randn_like(u), x, out, None, True, False)
    got = vms_hip.last_kernel()
    assert (got if "segments" in bwd_name else got.split("+")[0]) == bwd_name, got
    torch.cuda.synchronize()


def test_tensors_beyond_31_bit_offsets_take_the_generic_kernels(oracle):
    """The paired kernels address with 32-bit element offsets; a problem whose tensors span >= 2^31 elements must not
    fail or wrap: it runs on the generic kernels (64-bit strides) and vms_last_kernel() says so.  Rows sampled against
    the oracle (rows are independent given B, C).  8.6 GB of bf16 activations: fine on a 288 GB part."""
    import selective_scan_cuda
    import vms_hip
    b, d, L, N = 1, 1024 + 32, 2 * 1024 * 1024, 16     # d * L = 2.2e9 elements per tensor
    dt = torch.bfloat16
    torch.manual_seed(0)
    u = torch.empty(b, d, L, device=DEV, dtype=dt).normal_()
    delta = torch.empty(b, d, L, device=DEV, dtype=dt).uniform_(0.0, 0.3)
    A = -torch.rand(d, N, device=DEV) - 0.5
    Bm = torch.randn(b, 1, N, L, device=DEV, dtype=dt)
    Cm = torch.randn(b, 1, N, L, device=DEV, dtype=dt)
    out, x = selective_scan_cuda.fwd(u, delta, A, Bm, Cm, None, None, None, False)
    assert vms_hip.last_kernel() == "scan_fwd_generic", vms_hip.last_kernel()
    torch.cuda.synchronize()
    # the last rows sit beyond the 2^31-element mark: a wrapped offset would read / write another row
    rows = [0, 1023, d - 1]
    Lc = 4096      # the scan is causal: the first Lc outputs depend on the first Lc inputs only
    f = lambda t: t.float().cpu().numpy()
    ref = oracle.scan_fwd(f(u[:, rows, :Lc]), f(delta[:, rows, :Lc]), f(A[rows]), f(Bm[..., :Lc]), f(Cm[..., :Lc]),
                          None, None, None, False, prec="f64")["out"]
    got = f(out[:, rows, :Lc])
    scale = np.abs(ref).max()
    assert np.abs(got - ref).max() <= 1e-2 * scale
    # and the far end of the last row was written (not left to a wrapped store)
    tail = out[0, d - 1, -4096:].float()
    assert torch.isfinite(tail).all() and tail.abs().max() > 0
    del out, x
    torch.cuda.empty_cache()


def test_c_abi_is_reentrant_across_threads_and_streams(oracle):
    """Two host threads, each with its own stream, call the C ABI concurrently (forward + backward of the paired scan
    kernels -- the backward needs its > 64 KB LDS attribute, set under std::call_once per device -- and the conv): the
    results equal the single-threaded ones bit for bit where the kernels are deterministic (out, du, ddelta, dz, conv
    out / dx).  This is how the reference's nn.DataParallel callers drive the extension (train_eval.py:76: one
    autograd thread per device)."""
    import threading
    import causal_conv1d_cuda
    import selective_scan_cuda
    problems = []
    for seed in (1, 2):
        g = _rows_problem((2, 64, 2048, 1), torch.bfloat16, True, seed=seed)
        f = lambda k, dt=torch.bfloat16: G(g[k], dt)
        problems.append((f("u"), f("delta"), f("A", torch.float32), f("B"), f("C"), f("D", torch.float32), f("z"),
                         f("delta_bias", torch.float32), f("g")))

    def work(pr):
        u, dl, A, B, C, D, z, bias, dout = pr
        out, x, oz = selective_scan_cuda.fwd(u, dl, A, B, C, D, z, bias, True)
        res = selective_scan_cuda.bwd(u, dl, A, B, C, D, z, bias, dout, x, out, None, True, False)
        w = A[:, :4].contiguous()
        y = causal_conv1d_cuda.causal_conv1d_fwd(u, w, bias, True)
        dx = causal_conv1d_cuda.causal_conv1d_bwd(u, w, bias, dout, None, True)[0]
        return [oz, res[0], res[1], res[7], y, dx]
    want = [work(pr) for pr in problems]
    torch.cuda.synchronize()
    got, errs = [None, None], []

    def runner(i):
        try:
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                for _ in range(20):
                    got[i] = work(problems[i])
            s.synchronize()
        except Exception as e:  # noqa: BLE001
            errs.append(e)
    ths = [threading.Thread(target=runner, args=(i,)) for i in range(2)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert not errs, errs
    for i in range(2):
        for a, b_ in zip(got[i], want[i]):
            assert torch.equal(a, b_)


def test_compiled_binding_equals_ctypes_binding(monkeypatch):
    """The two host paths over the C ABI -- the compiled PyTorch extension (csrc/torch_binding, what the reference's
    pybind11 modules are) and the ctypes binding -- fill the same parameter blocks: every deterministic result is
    bit-identical, returned lists have the same structure, and errors surface as RuntimeError from both."""
    import causal_conv1d_cuda
    import selective_scan_cuda
    import vms_hip
    if vms_hip.ext() is None:
        pytest.skip("compiled binding not built (python video-mamba-suite_amd/csrc/torch_binding/build.py)")
    ext = vms_hip.ext()
    g = _rows_problem((2, 64, 1569, 1), torch.bfloat16, True, seed=21)   # ragged length: pad_bc + bc_pad through both
    f = lambda k, dt=torch.bfloat16: G(g[k], dt)
    u, dl, A, B, C, D, z, bias, dout = (f("u"), f("delta"), f("A", torch.float32), f("B"), f("C"), f("D", torch.float32),
                                        f("z"), f("delta_bias", torch.float32), f("g"))
    w = torch.randn(64, 4, device=DEV)

    def run():
        out, x, oz = selective_scan_cuda.fwd(u, dl, A, B, C, D, z, bias, True, True)
        res = selective_scan_cuda.bwd(u, dl, A, B, C, D, z, bias, dout, x, out, None, True, True, True)
        y = causal_conv1d_cuda.causal_conv1d_fwd(u, w, bias, True)
        cb = causal_conv1d_cuda.causal_conv1d_bwd(u, w, bias, dout, None, True)
        return [out, x, oz] + list(res) + [y] + list(cb)
    a = run()
    assert vms_hip.last_kernel().startswith("conv_bwd")
    monkeypatch.setattr(vms_hip, "_ext", None)
    b_ = run()
    monkeypatch.setattr(vms_hip, "_ext", ext)
    assert len(a) == len(b_)
    det = {0, 1, 2, 3, 4, 10, 11, 12, 13}   # out, x, out_z, du, ddelta, dz, recomputed out_z, conv out, conv dx
    for i, (p_, q_) in enumerate(zip(a, b_)):
        assert p_.shape == q_.shape and p_.dtype == q_.dtype and p_.stride() == q_.stride(), i
        if i in det:
            assert torch.equal(p_, q_), i
        else:
            check(p_, q_, 1e-3, f"result {i}")
    for use_ext in (True, False):
        monkeypatch.setattr(vms_hip, "_ext", ext if use_ext else None)
        with pytest.raises(RuntimeError):
            selective_scan_cuda.fwd(u, dl.float(), A, B, C, D, z, bias, True)
        with pytest.raises(RuntimeError):
            causal_conv1d_cuda.causal_conv1d_fwd(u, torch.randn(64, 5, device=DEV), None, True)


# BASELINE.json configs at full size: [1] block shape, [2] TimeMamba-B tokens, [3] DBM feature sequence,
# [4] long-video regime (per-GPU shard)
FULL_SIZES = {"cfg2_8x1024x8192": (8, 1024, 8192), "cfg3_8x768x3136": (8, 768, 3136),
              "cfg4_2x512x2304": (2, 512, 2304), "cfg5_1x768x65536": (1, 768, 65536)}


@pytest.mark.parametrize("seqlen", [2304, 1569])
@pytest.mark.parametrize("itype", [torch.bfloat16, torch.float32])
def test_reverse_from_equals_two_calls(seqlen, itype):
    """vms_hip.h ABI v5 `reverse_from`: batch entries >= reverse_from run right-to-left.  Scan forward / backward and conv
    forward / backward on a batch of 4 with reverse_from = 2 (and 1: uneven halves) equal the two calls on the two
    sub-batches -- bit for bit where the kernels are deterministic (out, out_z, x, du, ddelta, dz, conv out / dx), to
    atomics noise for the reduced gradients."""
    import causal_conv1d_cuda
    import selective_scan_cuda
    b, d, N, L = 4, 64, 16, seqlen
    torch.manual_seed(0)
    xz = torch.randn(b, 2 * d, L, device=DEV, dtype=itype)
    u, z = xz[:, :d], xz[:, d:]
    delta = (0.5 * torch.rand(d, b, L, device=DEV)).to(itype).permute(1, 0, 2)
    A = -torch.rand(d, N, device=DEV) - 0.2
    Bm, Cm = torch.randn(b, 1, N, L, device=DEV, dtype=itype), torch.randn(b, 1, N, L, device=DEV, dtype=itype)
    D, bias = torch.randn(d, device=DEV), torch.rand(d, device=DEV)
    dout = torch.randn(b, d, L, device=DEV, dtype=itype)
    w, cb = torch.randn(d, 4, device=DEV), torch.randn(d, device=DEV)
    for k in (2, 1):
        out, x, out_z = selective_scan_cuda.fwd(u, delta, A, Bm, Cm, D, z, bias, True, reverse_from=k)
        parts = [selective_scan_cuda.fwd(u[a:e], delta[a:e], A, Bm[a:e], Cm[a:e], D, z[a:e], bias, True, rv)
                 for a, e, rv in ((0, k, False), (k, b, True))]
        for i, name in enumerate(("out", "x", "out_z")):
            ref = torch.cat([parts[0][i], parts[1][i]], dim=0)
            got = (out, x, out_z)[i]
            assert torch.equal(got, ref), f"scan fwd {name} reverse_from={k}"
        dz = torch.empty_like(xz)[:, d:]
        g = selective_scan_cuda.bwd(u, delta, A, Bm, Cm, D, z, bias, dout, x, out, dz, True, False, reverse_from=k)
        gp = []
        for a, e, rv in ((0, k, False), (k, b, True)):
            dzp = torch.empty_like(xz[a:e])[:, d:]
            gp.append(selective_scan_cuda.bwd(u[a:e], delta[a:e], A, Bm[a:e], Cm[a:e], D, z[a:e], bias, dout[a:e], parts[0 if a == 0 else 1][1],
                                              parts[0 if a == 0 else 1][0], dzp, True, False, rv))
        for i, name in ((0, "du"), (1, "ddelta"), (7, "dz")):
            assert torch.equal(g[i], torch.cat([gp[0][i], gp[1][i]], dim=0)), f"scan bwd {name} reverse_from={k}"
        for i, name in ((3, "dB"), (4, "dC")):
            check(g[i], torch.cat([gp[0][i], gp[1][i]], dim=0), 1e-2 if itype == torch.bfloat16 else 1e-4, f"scan bwd {name} reverse_from={k}")
        for i, name in ((2, "dA"), (5, "dD"), (6, "ddelta_bias")):
            check(g[i], gp[0][i] + gp[1][i], 1e-4, f"scan bwd {name} reverse_from={k}")
        y = causal_conv1d_cuda.causal_conv1d_fwd(u, w, cb, True, reverse_from=k)
        yp = torch.cat([causal_conv1d_cuda.causal_conv1d_fwd(u[:k], w, cb, True, False), causal_conv1d_cuda.causal_conv1d_fwd(u[k:], w, cb, True, True)])
        assert torch.equal(y, yp), f"conv fwd reverse_from={k}"
        dx, dw, db = causal_conv1d_cuda.causal_conv1d_bwd(u, w, cb, dout, None, True, reverse_from=k)
        lo = causal_conv1d_cuda.causal_conv1d_bwd(u[:k], w, cb, dout[:k], None, True, False)
        hi = causal_conv1d_cuda.causal_conv1d_bwd(u[k:], w, cb, dout[k:], None, True, True)
        assert torch.equal(dx, torch.cat([lo[0], hi[0]])), f"conv bwd dx reverse_from={k}"
        check(dw, lo[1] + hi[1], 1e-4, "conv dweight")
        check(db, lo[2] + hi[2], 1e-4, "conv dbias")


def test_dbm_stacked_node_equals_two_nodes(monkeypatch):
    """The DBM block as one node on a batch of 2 B (reverse_from = B, stacking folded into the projections' weight layouts)
    against one node per direction + torch.cat (VMS_DBM_TWO_NODES): same output, same gradients."""
    import mamba_ssm.modules._core as core
    from mamba_ssm.modules.mamba_new import Mamba as DBM
    torch.manual_seed(0)
    for (b, L, dm) in ((2, 2304, 512), (3, 777, 64)):
        m = DBM(dm, expand=1, bias=(dm == 64)).to(DEV)
        h = torch.randn(b, L, dm, device=DEV, dtype=torch.bfloat16, requires_grad=True)
        gout = torch.randn(b, L, dm, device=DEV, dtype=torch.bfloat16)
        params = list(m.parameters())

        def run():
            with torch.autocast("cuda", dtype=torch.bfloat16):
                y = m(h)
            return y, torch.autograd.grad(y, [h] + params, gout)
        assert core._DBM_STACKED
        y1, g1 = run()
        monkeypatch.setattr(core, "_DBM_STACKED", False)
        y2, g2 = run()
        monkeypatch.undo()
        check(y1, y2, 1e-2, "DBM y: stacked node vs two nodes")
        for (k, _), a, c in zip([("dh", None)] + list(m.named_parameters()), g1, g2):
            check(a, c, 2e-2, f"DBM grad {k}: stacked node vs two nodes")


@pytest.mark.parametrize("variant", ["vim", "dbm"])
@pytest.mark.parametrize("shape", [(2, 1569, 128), (2, 2048, 256)])
def test_compiled_inner_node_equals_python_node(variant, shape, monkeypatch):
    """csrc/torch_binding/vms_torch.cpp inner_fwd / inner_bwd (the fused node as ONE host call) against the Python statement
    of the same node (VMS_NO_INNER_EXT=1): same ATen / C-ABI calls in the same order, so the outputs are identical and the
    gradients agree to the run-to-run noise of the fp32 atomics (dA, dB, dC)."""
    import vms_hip
    if vms_hip.ext() is None or not hasattr(vms_hip.ext(), "inner_fwd"):
        pytest.skip("compiled binding not built")
    from mamba_ssm.modules.mamba_new import Mamba as DBM
    from mamba_ssm.modules.mamba_simple import Mamba
    b, L, dm = shape
    torch.manual_seed(0)
    m = (Mamba(dm, expand=1, bimamba_type="v2") if variant == "vim" else DBM(dm, expand=1)).to(DEV)
    h = torch.randn(b, L, dm, device=DEV, dtype=torch.bfloat16, requires_grad=True)
    gout = torch.randn(b, L, dm, device=DEV, dtype=torch.bfloat16)
    params = list(m.parameters())

    def run():
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = m(h)
        return y, torch.autograd.grad(y, [h] + params, gout)
    y1, g1 = run()
    monkeypatch.setattr(_dbg(), "no_inner_ext", True)
    y2, g2 = run()
    monkeypatch.setattr(_dbg(), "no_inner_ext", False)
    assert torch.equal(y1, y2)
    for (k, _), a, c in zip([("dh", None)] + list(m.named_parameters()), g1, g2):
        check(a, c, 1e-2, f"grad {k}: compiled node vs Python node")   # bf16: one ulp of the fp32-atomics noise


@pytest.mark.parametrize("itype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("cfg", list(FULL_SIZES))
def test_scan_full_size_rows_vs_oracle(oracle, itype, cfg):
    """BASELINE.json sizes ((8, 8192, 1024, 16) etc.): the oracle cannot run all rows in
    seconds, but rows are independent given (B, C) -- check a spread of rows fwd and bwd, plus a
    checksum-style property: dB/dC are sums over rows, so they must equal the sum of the
    per-row-block results computed by separate launches on row subsets."""
    from mamba_ssm.ops.selective_scan_interface import selective_scan_fn
    torch.manual_seed(0)
    (b, d, L), N = FULL_SIZES[cfg], 16
    if cfg != "cfg2_8x1024x8192" and itype == torch.float32:
        pytest.skip("fp32 I/O at full size is covered by cfg2")
    u = torch.randn(b, d, L, device=DEV, dtype=itype)
    delta = (0.5 * torch.rand(b, d, L, device=DEV)).to(itype)
    z = torch.randn(b, d, L, device=DEV, dtype=itype)
    A = -0.5 * torch.rand(d, N, device=DEV)
    B = torch.randn(b, 1, N, L, device=DEV, dtype=itype)
    C = torch.randn(b, 1, N, L, device=DEV, dtype=itype)
    D = torch.randn(d, device=DEV)
    bias = 0.5 * torch.rand(d, device=DEV)
    for t in (u, delta, z, A, B, C, D, bias):
        t.requires_grad_()
    out = selective_scan_fn(u, delta, A, B, C, D, z=z, delta_bias=bias, delta_softplus=True)
    gout = torch.randn_like(out)
    out.backward(gout)
    rows = [0, 1, d // 2 - 1, d - 1]
    batches = sorted({0, b - 1})
    f = lambda t: t.detach().float().cpu().numpy()
    tol = TOL[itype]
    for bi in batches:
        sl = (slice(bi, bi + 1), rows)
        o = oracle.scan_fwd(f(u[sl]), f(delta[sl]), f(A[rows]), f(B[bi:bi + 1]), f(C[bi:bi + 1]), f(D[rows]),
                            f(z[sl]), f(bias[rows]), True, prec="f64")
        check(out[sl], o["out_z"], tol, f"out rows batch {bi}")
        ob = oracle.scan_bwd(f(u[sl]), f(delta[sl]), f(A[rows]), f(B[bi:bi + 1]), f(C[bi:bi + 1]), f(D[rows]),
                             f(z[sl]), f(bias[rows]), f(gout[sl]), True, prec="f64")
        check(u.grad[sl], ob["du"], tol * 2, "du rows")
        check(delta.grad[sl], ob["ddelta"], tol * 2, "ddelta rows")
        check(z.grad[sl], ob["dz"], tol * 2, "dz rows")
        # the same rows under the REFERENCE'S OWN metric: element-wise rtol / atol, mamba/tests/ops/test_selective_scan.py:45-51,
        # 137-149 (the scale-relative metric above lets a wrong small element hide behind the row's largest)
        from test_parity_hardening import allclose_ref, ref_tolerances
        rt = ref_tolerances(itype, True)
        for name, got, want in (("out", out[sl], o["out_z"]), ("du", u.grad[sl], ob["du"]), ("ddelta", delta.grad[sl], ob["ddelta"]),
                                ("dz", z.grad[sl], ob["dz"])):
            allclose_ref(got, want, *rt[name], f"{name} rows of batch {bi}, the reference's rtol / atol")
    # additivity of the group sums: rerun batch 0 in two halves of the rows
    dB_full = B.grad[0:1].float()
    parts = []
    for lo, hi in ((0, d // 2), (d // 2, d)):
        args = [t[0:1, lo:hi].detach().clone().requires_grad_() for t in (u, delta)]
        Bp, Cp = B[0:1].detach().clone().requires_grad_(), C[0:1].detach().clone().requires_grad_()
        o2 = selective_scan_fn(args[0], args[1], A[lo:hi].detach(), Bp, Cp, D[lo:hi].detach(),
                               z=z[0:1, lo:hi].detach(), delta_bias=bias[lo:hi].detach(), delta_softplus=True)
        o2.backward(gout[0:1, lo:hi])
        parts.append(Bp.grad.float())
    check(parts[0] + parts[1], f(dB_full), 2e-2 if itype == torch.bfloat16 else 2e-3, "dB additivity over row blocks")


# =================================================================================================
# causal conv1d
# =================================================================================================
@pytest.mark.parametrize("name", golden_names("conv_"))
def test_conv_vs_oracle_and_golden(oracle, name):
    from causal_conv1d import causal_conv1d_fn
    g = load_golden(name)
    itype = itype_of(g)
    tol = TOL[itype]
    x = G(g["x"], itype, True)
    w = G(g["weight"], grad=True)
    b = G(g["bias"], grad=True) if "bias" in g else None
    act = "silu" if g["silu"] else None
    out = causal_conv1d_fn(x, w, b, act)
    out.backward(G(g["g"], itype))
    f = lambda t: None if t is None else t.detach().float().cpu().numpy()
    o = oracle.conv_fwd(f(x), f(w), f(b), bool(g["silu"]), prec="f64")
    ob = oracle.conv_bwd(f(x), f(w), f(b), g["g"], bool(g["silu"]), prec="f64")
    check(out, o, tol, "out vs oracle")
    check(out, g["out"], tol * 2, "out vs golden")
    check(x.grad, ob["dx"], tol, "dx vs oracle")
    check(x.grad, g["dx"], tol * 2, "dx vs golden")
    check(w.grad, ob["dweight"], tol * 2, "dweight vs oracle")
    check(w.grad, g["dweight"], tol * 3, "dweight vs golden")
    if b is not None:
        check(b.grad, ob["dbias"], tol * 2, "dbias vs oracle")


@pytest.mark.parametrize("channel_last", [False, True])
@pytest.mark.parametrize("itype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("silu", [False, True])
@pytest.mark.parametrize("width", [2, 3, 4])
@pytest.mark.parametrize("seqlen", [8, 151, 512, 1134, 4096])
def test_conv_reference_grid(oracle, seqlen, width, silu, itype, channel_last):
    """The reference's own grid (test_causal_conv1d.py:14-75): x is a channel slice of a wider
    tensor (non-trivial batch stride), both memory layouts, every width; dim reduced 4128 -> 264."""
    from causal_conv1d import causal_conv1d_fn
    torch.manual_seed(0)
    batch, dim = 2, 264
    if not channel_last:
        x = torch.randn(batch, 64 + dim + 24, seqlen, device=DEV, dtype=itype)[:, 64:64 + dim, :].requires_grad_()
    else:
        x = torch.randn(batch, seqlen, 64 + dim + 24, device=DEV, dtype=itype)[:, :, 64:64 + dim].transpose(1, 2).requires_grad_()
    w = torch.randn(dim, width, device=DEV, requires_grad=True)
    b = torch.randn(dim, device=DEV, requires_grad=True)
    out = causal_conv1d_fn(x, w, b, "silu" if silu else None)
    gout = torch.randn_like(out)
    out.backward(gout)
    f = lambda t: t.detach().float().cpu().numpy()
    tol = TOL[itype]
    check(out, oracle.conv_fwd(f(x), f(w), f(b), silu, prec="f64"), tol, "out")
    ob = oracle.conv_bwd(f(x), f(w), f(b), f(gout), silu, prec="f64")
    check(x.grad, ob["dx"], tol, "dx")
    check(w.grad, ob["dweight"], tol * 3, "dweight")
    check(b.grad, ob["dbias"], tol * 3, "dbias")


@pytest.mark.parametrize("name", golden_names("convupd_"))
@pytest.mark.parametrize("itype", [torch.float32, torch.bfloat16])
def test_conv_update(oracle, name, itype):
    from causal_conv1d import causal_conv1d_update
    g = load_golden(name)
    x = G(g["x"], itype)
    cs = G(g["conv_state_in"], itype)
    w = G(g["weight"])
    b = G(g["bias"]) if "bias" in g else None
    cs_in = cs.clone()
    out = causal_conv1d_update(x, cs, w, b, "silu" if g["silu"] else None)
    # state roll must be bit-exact (test_causal_conv1d.py:113)
    want_cs = torch.cat([cs_in[:, :, 1:], x[:, :, None]], dim=-1)
    assert torch.equal(cs, want_cs)
    f = lambda t: None if t is None else t.detach().float().cpu().numpy()
    o, _ = oracle.conv_update(f(x), f(cs_in), f(w), f(b), bool(g["silu"]), prec="f64")
    check(out, o, TOL[itype], "out")
    if itype == torch.float32:
        check(out, g["out"], 1e-3, "out vs golden")


@pytest.mark.parametrize("itype", [torch.float32, torch.float16, torch.bfloat16])
@pytest.mark.parametrize("silu_activation", [False, True])
@pytest.mark.parametrize("has_bias", [False, True])
@pytest.mark.parametrize("width", [2, 3, 4])
@pytest.mark.parametrize("dim", [2048, 2048 + 16, 4096])
def test_conv_update_reference_grid(dim, width, has_bias, silu_activation, itype):
    """The reference's own test of the single-token conv step, as it runs it (causal-conv1d/tests/test_causal_conv1d.py:78-114: batch 2,
    fp32 weight / bias, kernel against causal_conv1d_update_ref under its rtol / atol, the rolled state bit for bit)."""
    from causal_conv1d.causal_conv1d_interface import causal_conv1d_update, causal_conv1d_update_ref
    rtol, atol = (3e-4, 1e-3) if itype == torch.float32 else (3e-3, 5e-3)
    if itype == torch.bfloat16:
        rtol, atol = 1e-2, 5e-2
    torch.manual_seed(0)
    b = 2
    x = torch.randn(b, dim, device=DEV, dtype=itype)
    conv_state = torch.randn(b, dim, width, device=DEV, dtype=itype)
    weight = torch.randn(dim, width, device=DEV, dtype=torch.float32)
    bias = torch.randn(dim, device=DEV, dtype=torch.float32) if has_bias else None
    conv_state_ref = conv_state.detach().clone()
    act = "silu" if silu_activation else None
    out = causal_conv1d_update(x, conv_state, weight, bias, activation=act)
    out_ref = causal_conv1d_update_ref(x, conv_state_ref, weight, bias, activation=act)
    assert out.dtype == out_ref.dtype and out.shape == out_ref.shape
    assert torch.equal(conv_state, conv_state_ref)
    assert torch.allclose(out, out_ref, rtol=rtol, atol=atol), (out.float() - out_ref.float()).abs().max().item()


def test_scan_deterministic_outputs():
    """The scan's non-atomic results -- out, out_z, the checkpoints x, du, ddelta, dz -- are bit-identical over 10,000
    forward + backward repeats (what test_causal_conv1d_race_condition asks of the conv, asked of the scan); dA / dB /
    dC / dD / ddelta_bias (fp32 atomics) agree to 1e-4 of their scale."""
    import selective_scan_cuda
    g = _rows_problem((2, 64, 2048, 1), torch.bfloat16, True, seed=17)
    f = lambda k, dt=torch.bfloat16: G(g[k], dt)
    u, dl, A, B, C, D, z, bias, dout = (f("u"), f("delta"), f("A", torch.float32), f("B"), f("C"), f("D", torch.float32),
                                        f("z"), f("delta_bias", torch.float32), f("g"))

    def run():
        out, x, oz = selective_scan_cuda.fwd(u, dl, A, B, C, D, z, bias, True)
        r = selective_scan_cuda.bwd(u, dl, A, B, C, D, z, bias, dout, x, out, None, True, False, keep_fp32=True)
        return [out, x, oz, r[0], r[1], r[7]], [r[2], r[3], r[4], r[5], r[6]]
    det0, at0 = run()
    for it in range(10000):
        det, at = run()
        for a, b_ in zip(det, det0):
            assert torch.equal(a, b_), it
        if it % 100 == 0:
            for a, b_ in zip(at, at0):
                assert (a - b_).abs().max() <= 1e-4 * b_.abs().max()


@pytest.mark.parametrize("channel_last", [True, False])
def test_conv_deterministic_outputs(channel_last):
    """test_causal_conv1d_race_condition (causal-conv1d/tests/test_causal_conv1d.py:117-173) with its own 10,000 repeats, its
    layouts (channel-last = the case it runs, a dim not divisible by 64 sliced out of a wider buffer; and the L-contiguous slice
    it keeps commented out) and its tolerances: out and dx bit-identical to the first run every time, dweight / dbias (fp32
    atomics over batch x L) within atol 1e-4 ... of a scale-relative 1e-4 (theirs is absolute at |dw| ~ 100).  dim is 520 here
    instead of 4128 so that 10,000 forward + backward pairs stay within seconds."""
    from causal_conv1d import causal_conv1d_fn
    torch.manual_seed(0)
    dim, L = 512 + 8, 2048
    if channel_last:
        x = torch.randn(2, L, 64 + dim + 24, device=DEV, dtype=torch.bfloat16)[:, :, 64:64 + dim].transpose(1, 2).requires_grad_()
    else:
        x = torch.randn(2, 64 + dim + 24, L, device=DEV, dtype=torch.bfloat16)[:, 64:64 + dim, :].requires_grad_()
    w = torch.randn(dim, 4, device=DEV, requires_grad=True)
    b = torch.randn(dim, device=DEV, requires_grad=True)
    out0 = causal_conv1d_fn(x, w, b, "silu")
    g = torch.randn_like(out0)
    dx0, dw0, db0 = torch.autograd.grad(out0, (x, w, b), g)
    for it in range(10000):
        out = causal_conv1d_fn(x, w, b, "silu")
        dx, dw, db = torch.autograd.grad(out, (x, w, b), g)
        assert torch.equal(out, out0) and torch.equal(dx, dx0), it
        assert (dw - dw0).abs().max() <= 1e-4 * dw0.abs().max() and (db - db0).abs().max() <= 1e-4 * db0.abs().max(), it


@pytest.mark.parametrize("itype", [torch.bfloat16, torch.float16, torch.float32])
@pytest.mark.parametrize("seqlen", [6144, 8191, 8192, 65536])
@pytest.mark.parametrize("reverse", [False, True])
@pytest.mark.parametrize("accumulate", [False, True])
def test_conv_long_rows_vs_oracle(oracle, seqlen, itype, reverse, accumulate):
    """Row lengths that take the 4-strips-per-wave kernels (seqlen >= 768 vectors: 6144 for 16-bit, 3072 for fp32;
    conv_fwd/bwd_kernel<T, SILU, VEC, 4, DIR> -- what the benchmark at L = 8192 and the L = 65536 config launch),
    in the layout the block uses: x = the first channel half of an xz buffer (batch stride 2 D L, SSI:175), dx
    written into / accumulated onto the matching half of dxz.  Every element against the oracle; reverse = the
    oracle on flipped copies; 8191 = ragged rows on the same kernels."""
    import causal_conv1d_cuda
    torch.manual_seed(seqlen % 1000 + 7 * reverse)
    b, d = 2, 40
    xz = torch.randn(b, 2 * d, seqlen, device=DEV).to(itype)
    x = xz[:, :d]
    w, bias = torch.randn(d, 4, device=DEV), torch.randn(d, device=DEV)
    dout = torch.randn(b, d, seqlen, device=DEV).to(itype)
    f = lambda t: t.detach().float().cpu().numpy()
    fl = (lambda a: np.ascontiguousarray(a[..., ::-1])) if reverse else (lambda a: a)
    tol = TOL[itype]
    import vms_hip
    out = causal_conv1d_cuda.causal_conv1d_fwd(x, w, bias, True, reverse)
    assert vms_hip.last_kernel() == "conv_fwd_strips4"
    check(out, fl(oracle.conv_fwd(fl(f(x)), f(w), f(bias), True, prec="f64")), tol, "out")
    dxz = torch.randn(b, 2 * d, seqlen, device=DEV).to(itype)
    start = dxz.clone()
    dx, dw, db = causal_conv1d_cuda.causal_conv1d_bwd(x, w, bias, dout, dxz[:, :d], True, reverse, accumulate_dx=accumulate)
    assert dx.data_ptr() == dxz.data_ptr() and vms_hip.last_kernel() == "conv_bwd_strips4"
    ob = oracle.conv_bwd(fl(f(x)), f(w), f(bias), fl(f(dout)), True, prec="f64")
    want = fl(ob["dx"]) + (f(start[:, :d]) if accumulate else 0.0)
    check(dx, want, tol, "dx")
    assert torch.equal(dxz[:, d:], start[:, d:]), "the z half of dxz was touched"
    check(dw, ob["dweight"], tol * 3, "dweight")
    check(db, ob["dbias"], tol * 3, "dbias")


@pytest.mark.parametrize("cfg", ["cfg2_8x1024x8192", "cfg5_1x768x65536"])
def test_conv_full_size_block_layout(oracle, cfg):
    """The conv launches of the benchmark itself ((8, 1024, 8192) bf16 inside a (8, 2048, 8192) xz buffer, both
    directions, dx accumulated by the second direction as BiMambaInnerFnNoOutProj does) and of the long-video config
    ((1, 768, 65536)): a spread of channels (rows are independent) against the oracle, all batches, so dweight / dbias
    of those channels are checked too."""
    import causal_conv1d_cuda
    (b, d, L) = FULL_SIZES[cfg]
    torch.manual_seed(1)
    xz = torch.randn(b, 2 * d, L, device=DEV, dtype=torch.bfloat16)
    x = xz[:, :d]
    w, bias = torch.randn(d, 4, device=DEV), torch.randn(d, device=DEV)
    dout = torch.randn(b, d, L, device=DEV, dtype=torch.bfloat16)
    rows = [0, 1, 63, 64, d // 2 - 1, d // 2, d - 2, d - 1]
    f = lambda t: t.detach().float().cpu().numpy()
    dxz = torch.zeros_like(xz)
    for reverse in (False, True):
        fl = (lambda a: np.ascontiguousarray(a[..., ::-1])) if reverse else (lambda a: a)
        out = causal_conv1d_cuda.causal_conv1d_fwd(x, w, bias, True, reverse)
        check(out[:, rows], fl(oracle.conv_fwd(fl(f(x[:, rows])), f(w[rows]), f(bias[rows]), True, prec="f64")), 1e-2,
              f"out rows reverse={reverse}")
        before = f(dxz[:, rows])
        dx, dw, db = causal_conv1d_cuda.causal_conv1d_bwd(x, w, bias, dout, dxz[:, :d], True, reverse,
                                                          accumulate_dx=reverse)
        ob = oracle.conv_bwd(fl(f(x[:, rows])), f(w[rows]), f(bias[rows]), fl(f(dout[:, rows])), True, prec="f64")
        check(dx[:, rows], fl(ob["dx"]) + (before if reverse else 0.0), 1e-2, f"dx rows reverse={reverse}")
        check(dw[rows], ob["dweight"], 3e-2, "dweight rows")
        check(db[rows], ob["dbias"], 3e-2, "dbias rows")
    assert float(dxz[:, d:].abs().max()) == 0.0


# =================================================================================================
# fused nodes and modules vs golden (reference CPU path)
# =================================================================================================
INNER_KEYS = ("xz", "conv1d_weight", "conv1d_bias", "x_proj_weight", "delta_proj_weight", "out_proj_weight",
              "A", "A_b", "D", "delta_bias")


@pytest.mark.parametrize("kind", ["no_out_proj", "out_proj", "bi"])
def test_fused_inner_vs_golden(kind):
    from mamba_ssm.ops import selective_scan_interface as ssi
    g = load_golden("inner_" + kind)
    t = {k: G(g[k], grad=True) for k in INNER_KEYS}
    if kind == "no_out_proj":
        out = ssi.mamba_inner_fn_no_out_proj(t["xz"], t["conv1d_weight"], t["conv1d_bias"], t["x_proj_weight"],
                                             t["delta_proj_weight"], t["A"], None, None, t["D"],
                                             delta_bias=t["delta_bias"], delta_softplus=True)
    elif kind == "out_proj":
        out = ssi.mamba_inner_fn(t["xz"], t["conv1d_weight"], t["conv1d_bias"], t["x_proj_weight"],
                                 t["delta_proj_weight"], t["out_proj_weight"], None, t["A"], None, None, t["D"],
                                 delta_bias=t["delta_bias"], delta_softplus=True)
    else:
        out = ssi.bimamba_inner_fn(t["xz"], t["conv1d_weight"], t["conv1d_bias"], t["x_proj_weight"],
                                   t["delta_proj_weight"], t["out_proj_weight"], None, t["A"], t["A_b"], None, None,
                                   t["D"], delta_bias=t["delta_bias"], delta_softplus=True)
    check(out, g["out"], 2e-3, "out")
    out.backward(G(g["g"]))
    for k in INNER_KEYS:
        if "d" + k in g:
            check(t[k].grad, g["d" + k], 5e-3, "d" + k)


def _inner768_cases():
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    from recipes import INNER768_CASES
    return sorted(INNER768_CASES)


@pytest.mark.parametrize("name", _inner768_cases())
def test_inner768_vs_reference_fixtures(name):
    """The three inner functions at the reference's OWN test problem (mamba/tests/ops/test_selective_scan.py:152-249: dim 768,
    dstate 8, dt_rank 48, L 128, W 3, batch 2, seed 0) over its grid -- constant / variable B and C, real / complex A -- plus what
    that test leaves out: the projection biases (SSI:164, 324, 358-359: dB_proj_bias = dB.sum over (batch, L)), the
    no-out-proj and the bidirectional function.  Expected values: the reference's mamba_inner_ref / bimamba_inner_ref on CPU
    (tests/golden/make_golden.py gen_inner768; the fixture holds every k-th element + sum and sum of squares of the output and of
    EVERY gradient).  Every gradient slot the reference returns is asserted (the reference's test only prints them)."""
    from recipes import INNER768_CASES, checksum, inner768_inputs, sample
    from mamba_ssm.ops import selective_scan_interface as ssi
    fn, var_B, var_C, is_complex, pbias = INNER768_CASES[name]
    g = load_golden(name)
    t = inner768_inputs(var_B, var_C, is_complex, pbias)
    for k, v in t.items():   # the same problem as the generator's (torch's CPU generator, same seed, same order)
        if v is not None:
            np.testing.assert_allclose(np.array(checksum(v)), g["in_sum." + k], rtol=1e-10, err_msg=f"input {k} differs from the fixture's")
    leaves = {k: v.to(DEV).requires_grad_() for k, v in t.items() if v is not None and not k.startswith("g_")}
    a = leaves.get
    if fn == "out_proj":
        out = ssi.mamba_inner_fn(a("xz"), a("conv1d_weight"), a("conv1d_bias"), a("x_proj_weight"), a("delta_proj_weight"),
                                 a("out_proj_weight"), None, a("A"), a("B"), a("C"), a("D"), delta_bias=a("delta_bias"),
                                 B_proj_bias=a("B_proj_bias"), C_proj_bias=a("C_proj_bias"), delta_softplus=True)
        gout = t["g_out_proj"]
    elif fn == "bi":
        out = ssi.bimamba_inner_fn(a("xz"), a("conv1d_weight"), a("conv1d_bias"), a("x_proj_weight"), a("delta_proj_weight"),
                                   a("out_proj_weight"), None, a("A"), a("A_b"), a("B"), a("C"), a("D"), delta_bias=a("delta_bias"),
                                   B_proj_bias=a("B_proj_bias"), C_proj_bias=a("C_proj_bias"), delta_softplus=True)
        gout = t["g_out_proj"]
    else:
        out = ssi.mamba_inner_fn_no_out_proj(a("xz"), a("conv1d_weight"), a("conv1d_bias"), a("x_proj_weight"), a("delta_proj_weight"),
                                             a("A"), a("B"), a("C"), a("D"), delta_bias=a("delta_bias"),
                                             B_proj_bias=a("B_proj_bias"), C_proj_bias=a("C_proj_bias"), delta_softplus=True)
        gout = t["g_no_out_proj"]

    def cmp(got, key, tol):
        got = got.detach().cpu()
        sm, stride = sample(got, 16384 if key == "out" else 8192)
        assert stride == int(g[key + ".stride"]), key
        check(sm, g[key], tol, key + " (sampled elements)")
        s_got, s_ref = np.array(checksum(got)), g[key + ".sums"]
        # sum of squares: a whole-tensor check of what the samples skip (2 x the element tolerance on the norm)
        assert abs(np.sqrt(s_got[1]) - np.sqrt(s_ref[1])) <= 2 * tol * np.sqrt(s_ref[1]), f"{key}: norm {np.sqrt(s_got[1])} vs {np.sqrt(s_ref[1])}"

    assert out.shape == tuple(gout.shape)
    cmp(out, "out", 2e-3)
    out.backward(gout.to(DEV))
    for k, leaf in leaves.items():
        if "d" + k in g:
            assert leaf.grad is not None, f"no gradient for {k}"
            cmp(leaf.grad, "d" + k, 5e-3 if k in ("xz",) else 1e-2)
        else:   # (A_b outside the bidirectional function, out_proj_weight of the function without one)
            assert leaf.grad is None or float(leaf.grad.abs().max()) == 0.0, k


@pytest.mark.parametrize("name,which,kw", [
    ("block_vim", "mamba_simple", dict(bimamba_type="v2")),
    ("block_vim_div", "mamba_simple", dict(bimamba_type="v2", if_devide_out=True)),
    ("block_vim_norm", "mamba_simple_scan_norm", dict(bimamba_type="v2", if_devide_out=True)),
    ("block_dbm", "mamba_new", dict(expand=1)),
    # round 6: d_state = 4 (the suite's CLIP ViViM, avion/models/model_clip.py:945-947) and expand = 2 at d_state 16
    ("block_vim_n4_div", "mamba_simple", dict(bimamba_type="v2", if_devide_out=True)),
    ("block_vim_n4", "mamba_simple", dict(bimamba_type="v2")),
    ("block_dbm_n4", "mamba_new", dict(expand=1)),
    ("block_vim_e2_n16", "mamba_simple", dict(bimamba_type="v2")),
])
@pytest.mark.parametrize("fast", [True, False])
def test_block_vs_golden(name, which, kw, fast):
    import importlib
    if which == "mamba_new" and not fast:
        pytest.skip("DBM has no slow path (reference mamba_new.py:216)")
    g = load_golden(name)
    Mamba = importlib.import_module("mamba_ssm.modules." + which).Mamba
    sd = {k[3:]: torch.tensor(v) for k, v in g.items() if k.startswith("sd.")}
    m = Mamba(g["x"].shape[-1], d_state=g["sd.A_log"].shape[1], d_conv=4, use_fast_path=fast, **({"expand": 2} | kw))
    m.load_state_dict(sd)
    m = m.to(DEV)
    x = G(g["x"], grad=True)
    y = m(x)
    check(y, g["y"], 2e-3, "y")
    y.backward(G(g["g"]))
    check(x.grad, g["dx"], 5e-3, "dx")
    for k, p in m.named_parameters():
        check(p.grad, g["grad." + k], 1e-2, "grad " + k)


@pytest.mark.parametrize("name", ["stack_ln", "stack_rms_fp32res"])
@pytest.mark.parametrize("fused_add_norm", [False, True])
@pytest.mark.parametrize("fast", [True, False])
def test_block_stack_vs_golden(name, fused_add_norm, fast):
    """Three reference Blocks (Add -> Norm -> ViM mixer, mamba_simple.py:381-437) + the closing add / norm_f, from the
    fixture the reference's own Block class produced on CPU: the fused add+norm HIP kernels (prenorm,
    residual_in_fp32) and the unfused form, through the one-node mixer and the unfused mixer."""
    from conftest import build_stack
    g = load_golden(name)
    layers, norm_f, run = build_stack(g, device=DEV, fused_add_norm=fused_add_norm, use_fast_path=fast)
    x = G(g["x"], grad=True)
    y = run(x)
    check(y, g["y"], 2e-3, "y")
    y.backward(G(g["g"]))
    check(x.grad, g["dx"], 5e-3, "dx")
    for prefix, mod in (("layers.", layers), ("norm_f.", norm_f)):
        for k, p in mod.named_parameters():
            check(p.grad, g["grad." + prefix + k], 1e-2, "grad " + prefix + k)


def _vim_torch_reference(m, h):
    """The ViM block written with torch ops only (F.conv1d, F.linear, selective_scan_ref's L-step recurrence), the way
    the reference's slow path composes it (mamba_simple.py:201-290 with use_fast_path=False and the flips of :244,
    :258): the full-size check's independent statement of the block."""
    from mamba_ssm.ops.selective_scan_interface import selective_scan_ref
    import torch.nn.functional as F
    Bt, L, _ = h.shape
    d, R, N = m.d_inner, m.dt_rank, m.d_state
    xz = F.linear(h, m.in_proj.weight, m.in_proj.bias).transpose(1, 2)

    def direction(xz_, sfx):
        conv, xp, dtp = getattr(m, "conv1d" + sfx), getattr(m, "x_proj" + sfx), getattr(m, "dt_proj" + sfx)
        x, z = xz_.chunk(2, dim=1)
        x = F.silu(F.conv1d(x, conv.weight, conv.bias, padding=m.d_conv - 1, groups=d)[..., :L])
        x_dbl = F.linear(x.transpose(1, 2), xp.weight)
        dt, Bm, Cm = torch.split(x_dbl, [R, N, N], dim=-1)
        delta = F.linear(dt, dtp.weight).transpose(1, 2)
        A = -torch.exp(getattr(m, "A" + sfx + "_log").float())
        return selective_scan_ref(x, delta, A, Bm.transpose(1, 2).contiguous(), Cm.transpose(1, 2).contiguous(),
                                  getattr(m, "D" + sfx).float(), z=z, delta_bias=dtp.bias.float(), delta_softplus=True)
    y = direction(xz, "") + direction(xz.flip(-1), "_b").flip(-1)
    return F.linear(y.transpose(1, 2), m.out_proj.weight, m.out_proj.bias)


def test_block_L8192_fp32_vs_torch_reference():
    """The benchmark's block shape (d_model 1024, expand 1, L = 8192; batch 2 to bound the reference's L-step
    autograd graph) in fp32 against the torch-only statement of the block: output and every gradient at the fp32 bar."""
    from mamba_ssm.modules.mamba_simple import Mamba
    torch.manual_seed(0)
    m = Mamba(1024, expand=1, bimamba_type="v2").to(DEV)
    h = torch.randn(2, 8192, 1024, device=DEV, requires_grad=True)
    g = torch.randn(2, 8192, 1024, device=DEV)
    y = m(h)
    got = torch.autograd.grad(y, [h] + list(m.parameters()), g)
    yr = _vim_torch_reference(m, h)
    want = torch.autograd.grad(yr, [h] + list(m.parameters()), g)
    check(y, yr, 1e-3, "y")
    names = ["dh"] + [k for k, _ in m.named_parameters()]
    for k, a, b_ in zip(names, got, want):
        check(a, b_.cpu().numpy(), 1e-3 if k == "dh" else 5e-3, "grad " + k)


def _vim_full_size(oracle, monkeypatch, d_model, b, L, wrap_block=False, batches=None):
    """A ViM ("v2") mixer at full size under autocast(bf16), forward + backward, checked three ways:
    (a) the one-node reverse-kernel path == the reference's form (flipped copies through the causal ops, conv_out /
    delta recomputed in backward: VMS_NO_REVERSE=1, VMS_CHECKPOINT_LVL=1); (b) the stages of the path, run with the
    public ops on the block's own tensors, against the oracle on sampled rows: conv (both directions) and the scan given
    (delta, B, C) (both directions); (c) the composition of those ops == the fused block output.
    wrap_block: the mixer sits in a Block(Add -> RMSNorm -> mixer) with the fused add+norm kernel and an fp32 residual
    stream, as the suite's backbones run it (configs[2])."""
    from functools import partial
    import mamba_ssm.modules._core as core
    from mamba_ssm.modules.mamba_simple import Block, Mamba
    from mamba_ssm.ops.triton.layernorm import RMSNorm, rms_norm_fn
    import causal_conv1d_cuda
    import selective_scan_cuda
    torch.manual_seed(0)
    if wrap_block:
        top = Block(d_model, partial(Mamba, expand=1, bimamba_type="v2"), norm_cls=partial(RMSNorm, eps=1e-5),
                    fused_add_norm=True, residual_in_fp32=True).to(DEV)
        m = top.mixer
    else:
        top = m = Mamba(d_model, expand=1, bimamba_type="v2").to(DEV)
    with torch.no_grad():  # move A / D off their deterministic init
        for k, p in m.named_parameters():
            if k.endswith("_log"):
                p.add_(0.3 * torch.randn_like(p))
            elif k in ("D", "D_b"):
                p.add_(0.5 * torch.randn_like(p))
    h = torch.randn(b, L, d_model, device=DEV, dtype=torch.bfloat16, requires_grad=True)
    gout = torch.randn(b, L, d_model, device=DEV, dtype=torch.bfloat16)
    params = list(top.parameters())
    names = ["dh"] + [k for k, _ in top.named_parameters()]

    def run():
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = top(h)[0] if wrap_block else top(h)
        return y, torch.autograd.grad(y, [h] + params, gout)
    assert core._USE_REVERSE_KERNELS and core._CHECKPOINT_LVL == 0
    y1, g1 = run()
    monkeypatch.setattr(core, "_USE_REVERSE_KERNELS", False)
    monkeypatch.setattr(core, "_CHECKPOINT_LVL", 1)
    y2, g2 = run()
    monkeypatch.undo()
    check(y1, y2, 1e-2, "y: one node vs reference form")
    for k, a, b_ in zip(names, g1, g2):
        check(a, b_, 1e-2 if k == "dh" else 2e-2, f"grad {k}: one node vs reference form")

    # (b) stage by stage against the oracle, on the tensors of this very block
    f = lambda t: t.detach().float().cpu().numpy()
    d, R, N = m.d_inner, m.dt_rank, m.d_state
    rows = sorted({0, 1, d // 2 - 1, d // 2, d - 2, d - 1})
    batches = batches if batches is not None else sorted({0, b - 1})
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        hin = h
        if wrap_block:
            hin, _ = rms_norm_fn(h, top.norm.weight, top.norm.bias, residual=None, prenorm=True, residual_in_fp32=True,
                                 eps=top.norm.eps)
        xz = m._in_projection(hin)
        x, z = xz[:, :d], xz[:, d:]
        ys = []
        for sfx, rev in (("", False), ("_b", True)):
            fl = (lambda a: np.ascontiguousarray(a[..., ::-1])) if rev else (lambda a: a)
            conv, xp, dtp = getattr(m, "conv1d" + sfx), getattr(m, "x_proj" + sfx), getattr(m, "dt_proj" + sfx)
            cw, cb = conv.weight.squeeze(1).float(), conv.bias.float()
            xc = causal_conv1d_cuda.causal_conv1d_fwd(x, cw, cb, True, rev)
            for bi in batches:
                o = oracle.conv_fwd(fl(f(x[bi:bi + 1, rows])), f(cw[rows]), f(cb[rows]), True, prec="f64")
                check(xc[bi:bi + 1, rows], fl(o), 1e-2, f"conv{sfx} rows batch {bi}")
            x_dbl = torch.nn.functional.linear(xc.transpose(1, 2), xp.weight.to(torch.bfloat16))     # (b, l, R + 2N)
            delta = torch.nn.functional.linear(x_dbl[..., :R], dtp.weight.to(torch.bfloat16)).transpose(1, 2).contiguous()
            Bm = x_dbl[..., R:R + N].transpose(1, 2).contiguous()[:, None]
            Cm = x_dbl[..., R + N:].transpose(1, 2).contiguous()[:, None]
            A = -torch.exp(getattr(m, "A" + sfx + "_log").float())
            Dv, bias = getattr(m, "D" + sfx).float(), dtp.bias.float()
            out, _, out_z = selective_scan_cuda.fwd(xc, delta, A, Bm, Cm, Dv, z, bias, True, rev)
            for bi in batches:
                sl = (slice(bi, bi + 1), rows)
                o = oracle.scan_fwd(fl(f(xc[sl])), fl(f(delta[sl])), f(A[rows]), fl(f(Bm[bi:bi + 1])), fl(f(Cm[bi:bi + 1])),
                                    f(Dv[rows]), fl(f(z[sl])), f(bias[rows]), True, prec="f64")
                check(out_z[sl], fl(o["out_z"]), 1e-2, f"scan{sfx} rows batch {bi}")
            ys.append(out_z)
        yc = torch.nn.functional.linear((ys[0].float() + ys[1].float()).to(torch.bfloat16).transpose(1, 2),
                                        m.out_proj.weight.to(torch.bfloat16))
    check(y1, yc, 1e-2, "fused block vs composition of the checked stages")


def test_block_full_size_bf16_bench_shape(oracle, monkeypatch):
    """configs[1] exactly as bench.py runs it: Mamba(1024, expand=1, "v2") at (8, 8192) under autocast(bf16)."""
    _vim_full_size(oracle, monkeypatch, 1024, 8, 8192)


def test_block_full_size_configs2_block(oracle, monkeypatch):
    """configs[2]: one Block(Add -> RMSNorm -> ViM 768) of the 12-layer stack at (8, 3136): fused add+norm kernel in
    front, 192 backward-scan workgroups for 256 CUs."""
    _vim_full_size(oracle, monkeypatch, 768, 8, 3136, wrap_block=True)


def test_block_full_size_configs4_long_video(oracle, monkeypatch):
    """configs[4]: the ViM block at (1, 65536, 768): both scans split into ranges of chunks INSIDE the block (24
    workgroups otherwise; round 4: the forward as scan_fwd_sg_kernel, one pass), conv rows of 65,536 elements."""
    import vms_hip
    _vim_full_size(oracle, monkeypatch, 768, 1, 65536)
    import selective_scan_cuda
    # the forward scan of this shape really is the split one (the check above covered its values)
    u = torch.randn(1, 768, 65536, device=DEV, dtype=torch.bfloat16)
    dl = torch.rand(1, 768, 65536, device=DEV).to(torch.bfloat16)
    Bm = torch.randn(1, 1, 16, 65536, device=DEV, dtype=torch.bfloat16)
    selective_scan_cuda.fwd(u, dl, -torch.rand(768, 16, device=DEV), Bm, Bm, None, None, None, True)
    # 768 rows for 1,024 SIMDs: one pass with a row's states over the four waves of a workgroup (round 3: ranges of chunks)
    assert vms_hip.last_kernel() == "scan_fwd_sg"


def test_block_full_size_configs3_dbm(oracle, monkeypatch):
    """configs[3]: the DBM block (shared weights, second half of the channels scanned right-to-left, mamba_new.py:168-229)
    at (2, 2304, 512) under autocast(bf16), forward + backward: (a) reverse-kernel path == the reference's form (the
    flipped half stacked on the batch axis, recompute in backward); (b) conv + scan stages of both halves against the
    oracle on sampled rows; (c) the composition of the checked stages == the block output."""
    import mamba_ssm.modules._core as core
    from mamba_ssm.modules.mamba_new import Mamba as DBM
    import causal_conv1d_cuda
    import selective_scan_cuda
    torch.manual_seed(0)
    d_model, b, L = 512, 2, 2304
    m = DBM(d_model, expand=1).to(DEV)
    with torch.no_grad():
        m.A_log.add_(0.3 * torch.randn_like(m.A_log))
        m.D.add_(0.5 * torch.randn_like(m.D))
    h = torch.randn(b, L, d_model, device=DEV, dtype=torch.bfloat16, requires_grad=True)
    gout = torch.randn(b, L, d_model, device=DEV, dtype=torch.bfloat16)
    params = list(m.parameters())
    names = ["dh"] + [k for k, _ in m.named_parameters()]

    def run():
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = m(h)
        return y, torch.autograd.grad(y, [h] + params, gout)
    y1, g1 = run()
    monkeypatch.setattr(core, "_USE_REVERSE_KERNELS", False)
    monkeypatch.setattr(core, "_CHECKPOINT_LVL", 1)
    y2, g2 = run()
    monkeypatch.undo()
    check(y1, y2, 1e-2, "y: reverse kernels vs stacked flipped copies")
    for k, a, b_ in zip(names, g1, g2):
        check(a, b_, 1e-2 if k == "dh" else 2e-2, f"grad {k}: reverse kernels vs stacked flipped copies")

    f = lambda t: t.detach().float().cpu().numpy()
    d, R, N = m.d_inner, m.dt_rank, m.d_state
    rows = sorted({0, 1, d // 2, d - 1})
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        xz4 = m._in_projection(h)                       # (b, 4 d, L): (x, z) forward half, (x, z) reversed half
        cw, cb = m.conv1d.weight.squeeze(1).float(), m.conv1d.bias.float()
        A = -torch.exp(m.A_log.float())
        Dv, bias = m.D.float(), m.dt_proj.bias.float()
        ys = []
        for half, rev in ((0, False), (1, True)):
            fl = (lambda a: np.ascontiguousarray(a[..., ::-1])) if rev else (lambda a: a)
            xz = xz4[:, 2 * d * half:2 * d * (half + 1)]
            x, z = xz[:, :d], xz[:, d:]
            xc = causal_conv1d_cuda.causal_conv1d_fwd(x, cw, cb, True, rev)
            for bi in range(b):
                o = oracle.conv_fwd(fl(f(x[bi:bi + 1, rows])), f(cw[rows]), f(cb[rows]), True, prec="f64")
                check(xc[bi:bi + 1, rows], fl(o), 1e-2, f"conv half {half} batch {bi}")
            x_dbl = torch.nn.functional.linear(xc.transpose(1, 2), m.x_proj.weight.to(torch.bfloat16))
            delta = torch.nn.functional.linear(x_dbl[..., :R], m.dt_proj.weight.to(torch.bfloat16)).transpose(1, 2).contiguous()
            Bm = x_dbl[..., R:R + N].transpose(1, 2).contiguous()[:, None]
            Cm = x_dbl[..., R + N:].transpose(1, 2).contiguous()[:, None]
            out, _, out_z = selective_scan_cuda.fwd(xc, delta, A, Bm, Cm, Dv, z, bias, True, rev)
            for bi in range(b):
                sl = (slice(bi, bi + 1), rows)
                o = oracle.scan_fwd(fl(f(xc[sl])), fl(f(delta[sl])), f(A[rows]), fl(f(Bm[bi:bi + 1])), fl(f(Cm[bi:bi + 1])),
                                    f(Dv[rows]), fl(f(z[sl])), f(bias[rows]), True, prec="f64")
                check(out_z[sl], fl(o["out_z"]), 1e-2, f"scan half {half} batch {bi}")
            ys.append(out_z)
        yc = torch.nn.functional.linear(torch.cat(ys, dim=1).transpose(1, 2), m.out_proj.weight.to(torch.bfloat16))
    check(y1, yc, 1e-2, "DBM block vs composition of the checked stages")


def test_block_bf16_autocast_runs_and_matches_fp32():
    """The suite trains under autocast(bf16) (run_class_finetuning.py:570-582): same module, bf16
    autocast vs fp32, agree to bf16 accuracy."""
    from mamba_ssm.modules.mamba_simple import Mamba
    torch.manual_seed(0)
    m = Mamba(128, expand=1, bimamba_type="v2").to(DEV)
    x = torch.randn(2, 300, 128, device=DEV, requires_grad=True)
    y32 = m(x)
    y32.float().square().mean().backward()
    g32 = {k: p.grad.clone() for k, p in m.named_parameters()}
    m.zero_grad()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y16 = m(x)
        loss = y16.float().square().mean()
    loss.backward()
    assert y16.dtype == torch.bfloat16
    check(y16, y32.detach().float().cpu().numpy(), 3e-2, "autocast output")
    for k, p in m.named_parameters():
        check(p.grad, g32[k].float().cpu().numpy(), 8e-2, "autocast grad " + k)


# =================================================================================================
# fused residual-add + LayerNorm / RMSNorm
# =================================================================================================
def _norm_run(g, itype, res_fp32=False):
    import mamba_ssm.ops.triton.layernorm as lnm
    x, w = G(g["x"], itype, True), G(g["weight"], grad=True)
    b = G(g["bias"], grad=True) if "bias" in g else None
    res = G(g["residual"], torch.float32 if res_fp32 else itype, True) if "residual" in g else None
    prenorm = bool(g["prenorm"])
    fn = lnm.rms_norm_fn if g["is_rms"] else lnm.layer_norm_fn
    out = fn(x, w, b, residual=res, eps=float(g["eps"]), prenorm=prenorm, residual_in_fp32=res_fp32)
    y, pre = out if prenorm else (out, None)
    loss = (y.float() * G(g["g"])).sum()
    if prenorm:
        loss = loss + (pre.float() * G(g["gpre"])).sum()
    loss.backward()
    return dict(y=y, pre=pre, dx=x.grad, dweight=w.grad, dbias=b.grad if b is not None else None,
                dresidual=res.grad if res is not None else None), (x, w, b, res)


@pytest.mark.parametrize("name", golden_names("norm_"))
def test_norm_vs_oracle_and_golden(oracle, name):
    """layer_norm_fn / rms_norm_fn through the HIP kernels vs the fixtures produced by the reference's
    layer_norm_ref / rms_norm_ref (+ autograd) and vs the oracle on the values the kernel saw."""
    g = load_golden(name)
    itype = itype_of(g)
    tol = TOL[itype]
    got, (x, w, b, res) = _norm_run(g, itype)
    N = g["x"].shape[-1]
    rows = g["x"].size // N
    f = lambda t: None if t is None else t.detach().float().cpu().numpy().reshape(-1, t.shape[-1]) if t.dim() > 1 else t.detach().float().cpu().numpy()
    o = oracle.norm_fwd(f(x), f(w), f(b), f(res), float(g["eps"]), bool(g["is_rms"]), prec="f64")
    ob = oracle.norm_bwd(o["res_out"], f(w), o["mean"], o["rstd"], g["g"].reshape(-1, N),
                         g["gpre"].reshape(-1, N) if "gpre" in g else None, bool(g["is_rms"]), has_bias="bias" in g,
                         prec="f64")
    check(got["y"], o["y"].reshape(g["y"].shape), tol, "y vs oracle")
    check(got["y"], g["y"], tol * 2, "y vs golden")
    if got["pre"] is not None:
        check(got["pre"], g["pre"], tol * 2, "prenorm sum vs golden")
    check(got["dx"], ob["ds"].reshape(g["dx"].shape), tol * 2, "dx vs oracle")
    check(got["dx"], g["dx"], tol * FACTOR["grad"]["golden"], "dx vs golden")
    if got["dresidual"] is not None:
        check(got["dresidual"], g["dresidual"], tol * FACTOR["grad"]["golden"], "dresidual vs golden")
    check(got["dweight"], ob["dw"], tol * 5, "dweight vs oracle")
    check(got["dweight"], g["dweight"], tol * FACTOR["batch_sum"]["golden"], "dweight vs golden")
    if got["dbias"] is not None:
        check(got["dbias"], ob["db"], tol * 5, "dbias vs oracle")


@pytest.mark.parametrize("itype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("N", [8, 64, 200, 384, 1000, 1024, 2048, 4096, 5000])
@pytest.mark.parametrize("is_rms", [False, True])
@pytest.mark.parametrize("mode", ["plain", "residual", "residual_fp32_prenorm", "fp32_prenorm_no_residual"])
def test_norm_shapes_vs_oracle(oracle, N, itype, is_rms, mode):
    """Every kernel variant (register-resident 1/2/4/8 pieces, element-wise rows, residual dtypes, prenorm)
    against the oracle; many more rows than resident waves so the persistent row loop and the dw / db
    partial rows are exercised."""
    torch.manual_seed(N)
    rows = 37 if N >= 2048 else 3000
    res_fp32 = "fp32" in mode
    g = dict(x=torch.randn(rows, N).numpy(), weight=(1 + 0.5 * torch.randn(N)).numpy(), bias=(0.5 * torch.randn(N)).numpy(),
             g=torch.randn(rows, N).numpy(), is_rms=int(is_rms), eps=1e-5, prenorm=int("prenorm" in mode))
    if "residual" in mode and "no_residual" not in mode:
        g["residual"] = torch.randn(rows, N).numpy()
    if g["prenorm"]:
        g["gpre"] = torch.randn(rows, N).numpy()
    tol = TOL[itype]
    got, (x, w, b, res) = _norm_run(g, itype, res_fp32)
    if res_fp32 and g["prenorm"]:
        assert got["pre"].dtype == torch.float32
    f = lambda t: None if t is None else t.detach().float().cpu().numpy()
    o = oracle.norm_fwd(f(x), f(w), f(b), f(res), 1e-5, is_rms, prec="f64")
    ob = oracle.norm_bwd(o["res_out"], f(w), o["mean"], o["rstd"], g["g"], g.get("gpre"), is_rms, prec="f64")
    check(got["y"], o["y"], tol, "y")
    if got["pre"] is not None:
        check(got["pre"], o["res_out"], tol, "prenorm sum")
    check(got["dx"], ob["ds"], tol * 2, "dx")
    if got["dresidual"] is not None:
        check(got["dresidual"], ob["ds"], tol * 2, "dresidual")
    check(got["dweight"], ob["dw"], tol * 5, "dweight")
    check(got["dbias"], ob["db"], tol * 5, "dbias")


@pytest.mark.parametrize("pair", [(torch.float32, torch.bfloat16), (torch.float32, torch.float16),
                                  (torch.float16, torch.bfloat16), (torch.bfloat16, torch.float16)])
@pytest.mark.parametrize("N", [64, 1000, 1024, 5000])
@pytest.mark.parametrize("is_rms", [False, True])
@pytest.mark.parametrize("prenorm", [False, True])
def test_norm_any_residual_dtype(oracle, pair, N, is_rms, prenorm):
    """The residual stream in a dtype that is neither x's nor fp32 (the reference's Triton kernels take any pair,
    layernorm.py:122-173): y in x's dtype, the pre-norm sum and dresidual in the residual's."""
    import mamba_ssm.ops.triton.layernorm as lnm
    itype, rtype = pair
    torch.manual_seed(N + 1)
    rows = 37 if N >= 2048 else 700
    x, res = G(torch.randn(rows, N).numpy(), itype, True), G(torch.randn(rows, N).numpy(), rtype, True)
    w, b = G((1 + 0.5 * torch.randn(N)).numpy(), grad=True), G((0.5 * torch.randn(N)).numpy(), grad=True)
    gy, gpre = torch.randn(rows, N), torch.randn(rows, N)
    fn = lnm.rms_norm_fn if is_rms else lnm.layer_norm_fn
    out = fn(x, w, b, residual=res, eps=1e-5, prenorm=prenorm)
    y, pre = out if prenorm else (out, None)
    assert y.dtype == itype and (pre is None or pre.dtype == rtype)
    loss = (y.float() * G(gy.numpy())).sum()
    if prenorm:
        loss = loss + (pre.float() * G(gpre.numpy())).sum()
    loss.backward()
    assert x.grad.dtype == itype and res.grad.dtype == rtype
    f = lambda t: None if t is None else t.detach().float().cpu().numpy()
    o = oracle.norm_fwd(f(x), f(w), f(b), f(res), 1e-5, is_rms, prec="f64")
    ob = oracle.norm_bwd(o["res_out"], f(w), o["mean"], o["rstd"], gy.numpy(), gpre.numpy() if prenorm else None, is_rms,
                         prec="f64")
    # the kernel normalises the unrounded fp32 sum but the backward re-reads the stored (rounded) one
    tol = max(TOL[itype], TOL[rtype])
    check(y, o["y"], TOL[itype], "y")
    if pre is not None:
        check(pre, o["res_out"], TOL[rtype], "prenorm sum")
    check(x.grad, ob["ds"], tol * 2, "dx")
    check(res.grad, ob["ds"], tol * 2, "dresidual")
    check(w.grad, ob["dw"], tol * 5, "dweight")
    check(b.grad, ob["db"], tol * 5, "dbias")


@pytest.mark.parametrize("shape", [(2048, 768), (2048, 1024), (37, 384), (1, 8), (500, 2052)])
@pytest.mark.parametrize("odt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("has_db", [True, False])
def test_norm_bwd_finish(shape, odt, has_db):
    """vms_layer_norm_bwd_finish: the sum over the backward's partial rows, both arrays in one launch, vs a float64 sum"""
    import vms_hip
    torch.manual_seed(shape[0])
    dwp = torch.randn(*shape, device=DEV)
    dbp = torch.randn(*shape, device=DEV) if has_db else None
    dw = torch.full((shape[1],), float("nan"), device=DEV, dtype=odt)
    db = torch.full((shape[1],), float("nan"), device=DEV, dtype=odt) if has_db else None
    vms_hip.norm_bwd_finish(dwp, dbp, dw, db)
    for got, src in ((dw, dwp), (db, dbp)):
        if got is None:
            continue
        want = src.double().sum(0)
        tol = (2.0 ** -8 if odt == torch.bfloat16 else 1e-5) * max(want.abs().max().item(), 1.0) + 1e-6 * shape[0]
        assert (got.double() - want).abs().max().item() <= tol


def test_norm_extension_errors():
    import layer_norm_cuda
    x = torch.randn(4, 64, device=DEV)
    w = torch.ones(64, device=DEV)
    with pytest.raises(RuntimeError):  # CPU tensors: no CPU path
        layer_norm_cuda.fwd(x.cpu(), w.cpu(), None, 1e-5)
    with pytest.raises(RuntimeError):  # weight shape
        layer_norm_cuda.fwd(x, torch.ones(32, device=DEV), None, 1e-5)
    with pytest.raises(RuntimeError):  # feature dim >= 64 KB (layernorm.py:150-153)
        layer_norm_cuda.fwd(torch.randn(2, 20000, device=DEV), torch.ones(20000, device=DEV), None, 1e-5)


# =================================================================================================
# single-token SSM step
# =================================================================================================
@pytest.mark.parametrize("name", golden_names("ssu_"))
@pytest.mark.parametrize("itype", [torch.float32, torch.bfloat16])
def test_state_update_vs_oracle_and_golden(oracle, name, itype):
    from mamba_ssm.ops.triton.selective_state_update import selective_state_update
    g = load_golden(name)
    t = lambda k, dt=itype: G(g[k], dt) if k in g else None
    state = G(g["state_in"], torch.float32)
    x, dt, z = t("x"), t("dt"), t("z")
    A, D, bias = t("A", torch.float32), t("D", torch.float32), t("dt_bias", torch.float32)
    B, C = t("B"), t("C")
    out = selective_state_update(state, x, dt, A, B, C, D, z=z, dt_bias=bias, dt_softplus=bool(g["softplus"]))
    f = lambda a: None if a is None else a.detach().float().cpu().numpy()
    o_out, o_st = oracle.state_update(g["state_in"], f(x), f(dt), f(A), f(B), f(C), f(D), f(z), f(bias),
                                      bool(g["softplus"]), prec="f64")
    tol = TOL[itype]
    check(out, o_out, tol, "out vs oracle")
    check(state, o_st, 1e-5, "state vs oracle (updated in place)")
    if itype == torch.float32:
        check(out, g["out"], 1e-4, "out vs golden")
        check(state, g["state_out"], 1e-5, "state vs golden")


@pytest.mark.parametrize("itype", [torch.float32, torch.float16, torch.bfloat16])
@pytest.mark.parametrize("has_z", [False, True])
@pytest.mark.parametrize("dstate", [16, 32, 64])
@pytest.mark.parametrize("dim", [2048, 2048 + 16, 4096])
def test_state_update_reference_grid(oracle, dim, dstate, has_z, itype):
    """The reference's own test of the single-token step, as it runs it (mamba/tests/ops/triton/test_selective_state_update.py:12-52:
    batch 2, dt_bias = U - 4, A = -U - 1, B / C / D fp32, softplus; kernel against selective_state_update_ref -- the PyTorch statement
    whose contraction sees the STORED state -- under its element-wise rtol / atol), plus, where the state is fp32, the f64 oracle."""
    from mamba_ssm.ops.triton.selective_state_update import selective_state_update, selective_state_update_ref
    rtol, atol = (3e-4, 1e-3) if itype == torch.float32 else (5e-3, 1e-2)
    if itype == torch.bfloat16:
        rtol, atol = 1e-2, 5e-2
    torch.manual_seed(0)
    b = 2
    state = torch.randn(b, dim, dstate, dtype=itype, device=DEV)
    x = torch.randn(b, dim, device=DEV, dtype=itype)
    dt = torch.randn(b, dim, device=DEV, dtype=itype)
    dt_bias = torch.rand(dim, device=DEV) - 4.0
    A = -torch.rand(dim, dstate, device=DEV) - 1.0
    Bm, Cm = torch.randn(b, dstate, device=DEV), torch.randn(b, dstate, device=DEV)
    D = torch.randn(dim, device=DEV)
    z = torch.randn_like(x) if has_z else None
    f = lambda a: None if a is None else a.detach().float().cpu().numpy()
    st0 = f(state)
    state_ref = state.detach().clone()
    out = selective_state_update(state, x, dt, A, Bm, Cm, D=D, z=z, dt_bias=dt_bias, dt_softplus=True)
    out_ref = selective_state_update_ref(state_ref, x, dt, A, Bm, Cm, D=D, z=z, dt_bias=dt_bias, dt_softplus=True)
    assert out.dtype == out_ref.dtype and out.shape == out_ref.shape
    assert torch.allclose(state.float(), state_ref.float(), rtol=rtol, atol=atol), (state.float() - state_ref.float()).abs().max().item()
    assert torch.allclose(out.float(), out_ref.float(), rtol=rtol, atol=atol), (out.float() - out_ref.float()).abs().max().item()
    if itype == torch.float32:
        o_out, o_st = oracle.state_update(st0, f(x), f(dt), f(A), f(Bm), f(Cm), f(D), f(z), f(dt_bias), True, prec="f64")
        for got, want, what in ((out, o_out, "out"), (state, o_st, "state")):
            got = f(got).astype(np.float64)
            bad = np.abs(got - want) > atol + rtol * np.abs(want)
            assert not bad.any(), f"{what} vs oracle: {bad.sum()} elements outside rtol {rtol} / atol {atol}, worst {np.abs(got - want).max():.3e}"


def test_state_update_keeps_dt_and_z_precision(oracle):
    """bf16 x with fp32 dt / z (Mamba.step under bf16 activations keeps dt in fp32): dt and z are loaded in their own
    dtypes (vms_hip.h dt_dtype / z_dtype), as the reference's kernel does -- the result matches the oracle fed the
    unrounded dt at the fp32 bar for the state."""
    from mamba_ssm.ops.triton.selective_state_update import selective_state_update
    torch.manual_seed(3)
    b, d, N = 2, 80, 16
    x = torch.randn(b, d, device=DEV, dtype=torch.bfloat16)
    dt = torch.rand(b, d, device=DEV) * 0.3 + 1e-3          # fp32: bf16 would lose 16 bits of it
    z = torch.randn(b, d, device=DEV)
    A = -torch.rand(d, N, device=DEV) - 0.5
    Bm, Cm = torch.randn(b, N, device=DEV), torch.randn(b, N, device=DEV)
    D, bias = torch.randn(d, device=DEV), torch.rand(d, device=DEV) * 0.1
    state = torch.randn(b, d, N, device=DEV)
    st0 = state.cpu().numpy()
    out = selective_state_update(state, x, dt, A, Bm, Cm, D, z=z, dt_bias=bias, dt_softplus=True)
    f = lambda a: a.detach().float().cpu().numpy()
    o_out, o_st = oracle.state_update(st0, f(x), f(dt), f(A), f(Bm), f(Cm), f(D), f(z), f(bias), True, prec="f64")
    check(state, o_st, 1e-5, "state (fp32 dt)")
    check(out, o_out, 1e-2, "out (bf16)")


def test_state_update_strided_and_low_precision_state(oracle):
    """x / z are halves of one projection output (Mamba.step: xz.chunk(2, dim=-1)), the state is bf16."""
    from mamba_ssm.ops.triton.selective_state_update import selective_state_update
    torch.manual_seed(0)
    b, d, N = 3, 96, 16
    xz = torch.randn(b, 2 * d, device=DEV, dtype=torch.bfloat16)
    x, z = xz.chunk(2, dim=-1)
    dt = torch.rand(b, d, device=DEV, dtype=torch.bfloat16)
    A = -torch.rand(d, N, device=DEV)
    Bm, Cm = torch.randn(b, N, device=DEV, dtype=torch.bfloat16), torch.randn(b, N, device=DEV, dtype=torch.bfloat16)
    D, bias = torch.randn(d, device=DEV), torch.rand(d, device=DEV)
    state = torch.randn(b, d, N, device=DEV).to(torch.bfloat16)
    st0 = state.float().cpu().numpy()
    out = selective_state_update(state, x, dt, A, Bm, Cm, D, z=z, dt_bias=bias, dt_softplus=True)
    f = lambda a: a.detach().float().cpu().numpy()
    o_out, o_st = oracle.state_update(st0, f(x), f(dt), f(A), f(Bm), f(Cm), f(D), f(z), f(bias), True, prec="f64")
    check(state, o_st, 1e-2, "bf16 state")
    check(out, o_out, 2e-2, "out")


def test_block_step_matches_full_forward():
    """Mamba.step (conv update + SSM step kernels) token by token == the fused forward's causal direction;
    checked on the unidirectional pieces the step API covers (reference mamba_simple.py:292-340)."""
    from mamba_ssm.modules.mamba_simple import Mamba
    from mamba_ssm.ops.selective_scan_interface import mamba_inner_fn
    torch.manual_seed(0)
    m = Mamba(64, d_state=16, expand=2, bimamba_type="v2").to(DEV)
    b, L = 2, 12
    h = torch.randn(b, L, 64, device=DEV)
    with torch.no_grad():
        xz = m._in_projection(h)
        A = -torch.exp(m.A_log.float())
        full = mamba_inner_fn(xz, m.conv1d.weight, m.conv1d.bias, m.x_proj.weight, m.dt_proj.weight, m.out_proj.weight,
                              m.out_proj.bias, A, None, None, m.D.float(), delta_bias=m.dt_proj.bias.float(),
                              delta_softplus=True)
        conv_state, ssm_state = m.allocate_inference_cache(b, L)
        outs = []
        for t in range(L):
            o, conv_state, ssm_state = m.step(h[:, t:t + 1], conv_state, ssm_state)
            outs.append(o)
        stepped = torch.cat(outs, dim=1)
    check(stepped, full.float().cpu().numpy(), 1e-3, "step-by-step vs fused forward")


@pytest.mark.parametrize("seqlen", [304, 301])
def test_block_step_is_hip_graph_capturable(seqlen):
    """Forward + backward of the ViM block captured into a HIP graph (torch.cuda.CUDAGraph) and replayed: the path does
    no host synchronisation, allocates only through torch's caching allocator and sets its launch attributes at
    warm-up, so small / launch-bound problems can run as one graph launch.  Replayed results == eager results."""
    from mamba_ssm.modules.mamba_simple import Mamba
    torch.manual_seed(0)
    m = Mamba(128, expand=1, bimamba_type="v2").to(DEV)
    params = list(m.parameters())
    x = torch.randn(2, seqlen, 128, device=DEV, dtype=torch.bfloat16, requires_grad=True)
    g = torch.randn(2, seqlen, 128, device=DEV, dtype=torch.bfloat16)

    def step():
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = m(x)
        return (y,) + torch.autograd.grad(y, [x] + params, g)
    eager = [t.detach().clone() for t in step()]
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        outs = step()
    x0 = x.detach().clone()
    x.data.copy_(torch.randn_like(x))          # another input in the captured buffer: the replay recomputes
    graph.replay()
    assert not torch.equal(outs[0], eager[0])
    x.data.copy_(x0)
    graph.replay()
    torch.cuda.synchronize()
    for i, (a, b_) in enumerate(zip(outs, eager)):
        check(a, b_.float().cpu().numpy(), 2e-2, f"graph output {i}")


# =================================================================================================
# per-step parameter preparation (vms_param_prep, csrc/param_prep.hip)
# =================================================================================================
@pytest.mark.parametrize("dst_dtype", [torch.bfloat16, torch.float16, torch.float32])
def test_param_prep_jobs_vs_torch(dst_dtype):
    """cast / cast + transpose / -exp of up to 8 matrices in one launch: bit-exact against torch's copies (round to nearest
    even), exp within 2 ulp; ragged tile edges, one-row jobs, strided rows."""
    import vms_hip
    torch.manual_seed(0)
    mk = lambda *s: torch.randn(*s, device=DEV)
    wide = mk(70, 200)
    srcs = [mk(2048, 1024), mk(96, 1024), mk(1024, 64), mk(33, 65), mk(1, 130), wide[:, 3:190], mk(1024, 16), mk(5, 7)]
    ops = [vms_hip.PREP_CAST_T, vms_hip.PREP_CAST, vms_hip.PREP_CAST, vms_hip.PREP_CAST_T, vms_hip.PREP_CAST,
           vms_hip.PREP_CAST_T, vms_hip.PREP_NEG_EXP, vms_hip.PREP_NEG_EXP]
    dsts = []
    for s, op in zip(srcs, ops):
        shape = tuple(s.shape[::-1]) if op == vms_hip.PREP_CAST_T else tuple(s.shape)
        dsts.append(torch.full(shape, float("nan"), device=DEV, dtype=torch.float32 if op == vms_hip.PREP_NEG_EXP else dst_dtype))
    vms_hip.param_prep(list(zip(srcs, dsts, ops)))
    assert vms_hip.last_kernel() == "param_prep"
    for s, d, op in zip(srcs, dsts, ops):
        if op == vms_hip.PREP_NEG_EXP:
            want = -torch.exp(s)
            assert ((d - want).abs() <= 3e-7 * want.abs()).all()
        else:
            want = (s.t() if op == vms_hip.PREP_CAST_T else s).to(dst_dtype)
            assert torch.equal(d, want)
    with pytest.raises(RuntimeError):   # shapes must match the op
        vms_hip.param_prep([(srcs[0], dsts[1], vms_hip.PREP_CAST)])


@pytest.mark.parametrize("d_model,b,L,kw", [(64, 2, 257, {}), (192, 1, 512, {"if_devide_out": True}), (1024, 2, 1024, {})])
def test_block_param_prep_equals_per_node_casts(monkeypatch, d_model, b, L, kw):
    """The ViM block with its parameters prepared by one launch (default) against the same block with every node casting
    its own (VMS_NO_PARAM_PREP=1): the casts are bit-identical, -exp differs by at most an ulp of A."""
    import mamba_ssm.modules._core as core
    from mamba_ssm.modules.mamba_simple import Mamba
    torch.manual_seed(0)
    m = Mamba(d_model, expand=1, bimamba_type="v2", **kw).to(DEV)
    h = torch.randn(b, L, d_model, device=DEV, dtype=torch.bfloat16, requires_grad=True)

    def step():
        h.grad = None
        m.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            out = m(h)
            out.float().square().mean().backward()
        return [out.detach().clone(), h.grad.clone()] + [p.grad.clone() for p in m.parameters()]

    assert core._PARAM_PREP
    got = step()
    monkeypatch.setattr(core, "_PARAM_PREP", False)
    want = step()
    names = ["out", "dh"] + [n for n, _ in m.named_parameters()]
    for n, a, w in zip(names, got, want):
        assert a.dtype == w.dtype and a.shape == w.shape, n
        check(a, w, 2e-3, f"param prep vs per-node casts: {n}")
    # fp32 (no autocast): the preparation does not apply, the block runs as before
    monkeypatch.setattr(core, "_PARAM_PREP", True)
    h32 = h.detach().float().requires_grad_()
    m(h32).square().mean().backward()
    assert torch.isfinite(h32.grad).all()


def test_prep_plan_rekeys_on_geometry_and_runs_from_two_threads():
    """ADVICE r4 on the cached per-module preparation plan (vms_hip.PrepPlan): (a) a source re-viewed with another shape at the
    SAME address must not reuse the cached rows / strides (the key is (data_ptr, shape, stride, dtype)); (b) run() builds its own
    parameter block per call: two threads preparing into different buffers at once each get their own result."""
    import threading
    import vms_hip
    torch.manual_seed(0)
    w = torch.randn(64, 48, device=DEV)
    plan = vms_hip.PrepPlan([(w, 0, 0, (64, 48), (48, 1), torch.bfloat16, vms_hip.PREP_CAST)])
    assert plan.matches([w])
    assert not plan.matches([w.view(48, 64)])          # same address, other geometry
    assert not plan.matches([w.view(32, 96)[:, :48]])  # same address, other strides
    outs = [torch.zeros(64 * 48, device=DEV, dtype=torch.bfloat16) for _ in range(2)]
    errs = []

    def work(k):
        try:
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                for _ in range(200):
                    plan.run((outs[k].data_ptr(),), outs[k])
            s.synchronize()
        except Exception as e:   # noqa: BLE001
            errs.append(e)
    ts = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs
    for o in outs:
        assert torch.equal(o.view(64, 48), w.to(torch.bfloat16))


# =================================================================================================
# both conv directions in one pass (vms_causal_conv1d_fwd_dual)
# =================================================================================================
@pytest.mark.parametrize("itype", [torch.bfloat16, torch.float16, torch.float32])
@pytest.mark.parametrize("shape", [(2, 6, 8), (2, 5, 151), (1, 3, 1569), (2, 4, 6144), (1, 2, 8191), (1, 3, 8192), (1, 2, 65536)])
@pytest.mark.parametrize("width", [2, 3, 4])
@pytest.mark.parametrize("bias,silu", [(True, True), (False, False)])
def test_conv_fwd_dual_equals_two_calls(shape, width, itype, bias, silu):
    """out == the causal call, out_b == the `reverse` call, bit for bit (same tap order); rows as channel halves of an xz
    buffer, ragged / unaligned lengths, one strip and four strips per wave"""
    import vms_hip
    b, d, L = shape
    torch.manual_seed(0)
    xz = torch.randn(b, 2 * d, L, device=DEV).to(itype)
    x = xz[:, :d]
    w, wb = torch.randn(d, width, device=DEV), torch.randn(d, width, device=DEV)
    cb, cbb = (torch.randn(d, device=DEV), torch.randn(d, device=DEV)) if bias else (None, None)
    o1, o2 = torch.empty(b, d, L, device=DEV, dtype=itype), torch.empty(b, d, L, device=DEV, dtype=itype)
    vms_hip.conv_fwd(x, w, cb, o1, silu)
    vms_hip.conv_fwd(x, wb, cbb, o2, silu, reverse=True)
    d1 = torch.full_like(o1, float("nan"))
    d2full = torch.full((b, d + 1, L), float("nan"), device=DEV, dtype=itype)   # out_b as a strided view
    d2 = d2full[:, :d]
    vms_hip.conv_fwd_dual(x, w, cb, d1, wb, cbb, d2, silu)
    assert vms_hip.last_kernel().startswith("conv_fwd_dual")
    # every dtype, with and without SiLU: both kernels narrow the ROUNDED fp32 result (round 3's fp16 exception -- hipcc folding
    # the last fma into the fp16 conversion in one kernel only -- is closed by pinning the fp32 value in both)
    assert torch.equal(d1, o1) and torch.equal(d2, o2)
    assert torch.isnan(d2full[:, d]).all()


@pytest.mark.parametrize("d_model,b,L", [(64, 2, 257), (256, 1, 1569), (512, 2, 2048)])
def test_block_dual_conv_equals_one_conv_per_direction(monkeypatch, d_model, b, L):
    """The ViM block with both directions' conv1d from one pass over x (default) against one conv1d launch per direction
    (VMS_NO_DUAL_CONV=1): the conv outputs are bit-identical, so the block output is; gradients differ by the order of the
    kernels' atomics only."""
    import mamba_ssm.ops.selective_scan_interface as ssi
    from mamba_ssm.modules.mamba_simple import Mamba
    torch.manual_seed(0)
    m = Mamba(d_model, expand=1, bimamba_type="v2").to(DEV)
    h = torch.randn(b, L, d_model, device=DEV, dtype=torch.bfloat16, requires_grad=True)

    def step():
        h.grad = None
        m.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            out = m(h)
            out.float().square().mean().backward()
        return [out.detach().clone(), h.grad.clone()] + [p.grad.clone() for p in m.parameters()]

    assert ssi._DUAL_CONV
    got = step()
    monkeypatch.setattr(ssi, "_DUAL_CONV", False)
    want = step()
    assert torch.equal(got[0], want[0])
    for n, a, w in zip(["dh"] + [n for n, _ in m.named_parameters()], got[1:], want[1:]):
        check(a, w, 2e-3, f"dual conv vs one conv per direction: {n}")


@pytest.mark.parametrize("which", ["dbm", "vim"])
def test_graphed_step_equals_eager(which):
    """mamba_ssm.utils.hip_graph.GraphedStep: forward + backward of a block recorded once as a HIP graph; replays on new
    inputs == the eager step (output, input gradient, every parameter gradient)."""
    from mamba_ssm.utils.hip_graph import GraphedStep
    torch.manual_seed(0)
    if which == "dbm":
        from mamba_ssm.modules.mamba_new import Mamba as M
        m, shape = M(128, expand=1).to(DEV), (2, 576, 128)
    else:
        from mamba_ssm.modules.mamba_simple import Mamba as M
        m, shape = M(128, expand=1, bimamba_type="v2").to(DEV), (2, 512, 128)
    gs = GraphedStep(m, torch.randn(*shape, device=DEV, dtype=torch.bfloat16))
    params = list(m.parameters())
    for seed in (1, 2):
        torch.manual_seed(seed)
        x = torch.randn(*shape, device=DEV, dtype=torch.bfloat16)
        g = torch.randn(*shape, device=DEV, dtype=torch.bfloat16)
        out, dx = gs(x, g)
        got = [out.clone(), dx.clone()] + [p.grad.clone() for p in params]
        xe = x.clone().requires_grad_()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            ye = m(xe)
        want = (ye,) + torch.autograd.grad(ye, [xe] + params, g)
        for i, (a, w) in enumerate(zip(got, want)):
            check(a, w, 2e-2, f"graphed step vs eager, tensor {i}, seed {seed}")

