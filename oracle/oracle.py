"""ctypes front-end of the CPU oracle (oracle/vms_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  Nothing under video-mamba-suite_amd/ may import
this module (tests/test_layout.py enforces it).

All functions take / return numpy float32 arrays (dense, row-major); inputs in a
16-bit I/O type are widened by the caller, outputs are rounded by the caller.
`prec` selects the arithmetic width of the restatement: "f32" mirrors the
reference's `.float()` arithmetic, "f64" is the tighter truth.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "build", "libvms_oracle.so")
_lib = None

_F = ctypes.POINTER(ctypes.c_float)


def build(force=False):
    """Compile the C restatement with gcc (oracle/Makefile)."""
    if force or not os.path.exists(_LIB_PATH) or (
        os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "vms_oracle.c"))
    ):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB_PATH)
    return _lib


def _p(a):
    if a is None:
        return ctypes.cast(None, _F)
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"], (a.dtype, a.flags)
    return a.ctypes.data_as(_F)


def _c(a):
    return None if a is None else np.ascontiguousarray(np.asarray(a, dtype=np.float32))


def _bc4(B, var):
    """(batch, N, L) -> (batch, 1, N, L) as SelectiveScanFn does (SSI:31-36)."""
    if var and B.ndim == 3:
        B = B[:, None]
    return B


def scan_fwd(u, delta, A, B, C, D=None, z=None, delta_bias=None, delta_softplus=False, prec="f32",
             threads=None):
    """Returns dict(out=pre-gate y+uD, out_z (or None), x=(b,d,n_chunks,2N) checkpoints,
    last_state=(b,d,N))."""
    u, delta, A, B, C, D, z, delta_bias = map(_c, (u, delta, A, B, C, D, z, delta_bias))
    batch, dim, L = u.shape
    N = A.shape[1]
    var_B, var_C = B.ndim >= 3, C.ndim >= 3
    B, C = _c(_bc4(B, var_B)), _c(_bc4(C, var_C))
    G = B.shape[1] if var_B else (C.shape[1] if var_C else 1)
    n_chunks = (L + 2047) // 2048
    out = np.empty_like(u)
    out_z = np.empty_like(u) if z is not None else None
    x = np.zeros((batch, dim, n_chunks, 2 * N), np.float32)
    last = np.empty((batch, dim, N), np.float32)
    if threads:
        os.environ["OMP_NUM_THREADS"] = str(threads)
    fn = getattr(lib(), "vms_oracle_scan_fwd_" + prec)
    fn.restype = None
    fn(ctypes.c_int(batch), ctypes.c_int(dim), ctypes.c_int(L), ctypes.c_int(N), ctypes.c_int(G),
       _p(u), _p(delta), _p(A), _p(B), _p(C), _p(D), _p(z), _p(delta_bias),
       ctypes.c_int(var_B), ctypes.c_int(var_C), ctypes.c_int(bool(delta_softplus)),
       _p(out), _p(out_z), _p(x), _p(last))
    return dict(out=out, out_z=out_z, x=x, last_state=last)


def scan_bwd(u, delta, A, B, C, D, z, delta_bias, dout, delta_softplus=False, prec="f32"):
    """Gradients of the final output (out_z if z is given else out) w.r.t. every input."""
    u, delta, A, B, C, D, z, delta_bias, dout = map(_c, (u, delta, A, B, C, D, z, delta_bias, dout))
    batch, dim, L = u.shape
    N = A.shape[1]
    var_B, var_C = B.ndim >= 3, C.ndim >= 3
    sqB, sqC = var_B and B.ndim == 3, var_C and C.ndim == 3
    B, C = _c(_bc4(B, var_B)), _c(_bc4(C, var_C))
    G = B.shape[1] if var_B else (C.shape[1] if var_C else 1)
    du, ddelta = np.empty_like(u), np.empty_like(u)
    dA = np.zeros_like(A)
    dB, dC = np.zeros_like(B), np.zeros_like(C)
    dD = np.zeros(dim, np.float32) if D is not None else None
    dz = np.empty_like(u) if z is not None else None
    dbias = np.zeros(dim, np.float32) if delta_bias is not None else None
    fn = getattr(lib(), "vms_oracle_scan_bwd_" + prec)
    fn.restype = None
    fn(ctypes.c_int(batch), ctypes.c_int(dim), ctypes.c_int(L), ctypes.c_int(N), ctypes.c_int(G),
       _p(u), _p(delta), _p(A), _p(B), _p(C), _p(D), _p(z), _p(delta_bias), _p(dout),
       ctypes.c_int(var_B), ctypes.c_int(var_C), ctypes.c_int(bool(delta_softplus)),
       _p(du), _p(ddelta), _p(dA), _p(dB), _p(dC), _p(dD), _p(dz), _p(dbias))
    if sqB:
        dB = dB[:, 0]
    if sqC:
        dC = dC[:, 0]
    return dict(du=du, ddelta=ddelta, dA=dA, dB=dB, dC=dC, dD=dD, dz=dz, ddelta_bias=dbias)


def conv_fwd(x, weight, bias=None, silu=False, prec="f32"):
    x, weight, bias = map(_c, (x, weight, bias))
    batch, dim, L = x.shape
    W = weight.shape[1]
    out = np.empty_like(x)
    fn = getattr(lib(), "vms_oracle_conv_fwd_" + prec)
    fn.restype = None
    fn(ctypes.c_int(batch), ctypes.c_int(dim), ctypes.c_int(L), ctypes.c_int(W), _p(x), _p(weight),
       _p(bias), ctypes.c_int(bool(silu)), _p(out))
    return out


def conv_bwd(x, weight, bias, dout, silu=False, prec="f32"):
    x, weight, bias, dout = map(_c, (x, weight, bias, dout))
    batch, dim, L = x.shape
    W = weight.shape[1]
    dx = np.empty_like(x)
    dw = np.zeros_like(weight)
    db = np.zeros(dim, np.float32) if bias is not None else None
    fn = getattr(lib(), "vms_oracle_conv_bwd_" + prec)
    fn.restype = None
    fn(ctypes.c_int(batch), ctypes.c_int(dim), ctypes.c_int(L), ctypes.c_int(W), _p(x), _p(weight),
       _p(bias), _p(dout), ctypes.c_int(bool(silu)), _p(dx), _p(dw), _p(db))
    return dict(dx=dx, dweight=dw, dbias=db)


def conv_update(x, conv_state, weight, bias=None, silu=False, prec="f32"):
    """Returns (out, new_conv_state); conv_state is not modified in place."""
    x, weight, bias = map(_c, (x, weight, bias))
    cs = np.array(conv_state, dtype=np.float32, order="C", copy=True)
    batch, dim = x.shape
    W = weight.shape[1]
    out = np.empty_like(x)
    fn = getattr(lib(), "vms_oracle_conv_update_" + prec)
    fn.restype = None
    fn(ctypes.c_int(batch), ctypes.c_int(dim), ctypes.c_int(W), _p(x), _p(cs), _p(weight), _p(bias),
       ctypes.c_int(bool(silu)), _p(out))
    return out, cs


def norm_fwd(x, weight, bias=None, residual=None, eps=1e-6, is_rms=False, prec="f32"):
    """-> dict(y, res_out (= x + residual), mean (None for rms), rstd); x: (rows, cols)."""
    x, weight, bias, residual = map(_c, (x, weight, bias, residual))
    rows, cols = x.shape
    y, res_out = np.empty_like(x), np.empty_like(x)
    mean = None if is_rms else np.empty(rows, np.float32)
    rstd = np.empty(rows, np.float32)
    fn = getattr(lib(), "vms_oracle_norm_fwd_" + prec)
    fn.restype = None
    fn(ctypes.c_int(rows), ctypes.c_int(cols), _p(x), _p(residual), _p(weight), _p(bias), ctypes.c_float(eps),
       ctypes.c_int(bool(is_rms)), _p(y), _p(res_out), _p(mean), _p(rstd))
    return dict(y=y, res_out=res_out, mean=mean, rstd=rstd)


def norm_bwd(s, weight, mean, rstd, dy, dres_out=None, is_rms=False, has_bias=True, prec="f32"):
    """s: the pre-norm sum x + residual saved by the forward.  -> dict(ds, dw, db)."""
    s, weight, mean, rstd, dy, dres_out = map(_c, (s, weight, mean, rstd, dy, dres_out))
    rows, cols = s.shape
    ds, dw = np.empty_like(s), np.empty(cols, np.float32)
    db = np.empty(cols, np.float32) if has_bias else None
    fn = getattr(lib(), "vms_oracle_norm_bwd_" + prec)
    fn.restype = None
    fn(ctypes.c_int(rows), ctypes.c_int(cols), _p(s), _p(weight), _p(mean), _p(rstd), _p(dy), _p(dres_out),
       ctypes.c_int(bool(is_rms)), _p(ds), _p(dw), _p(db))
    return dict(ds=ds, dw=dw, db=db)


def state_update(state, x, dt, A, B, C, D=None, z=None, dt_bias=None, dt_softplus=False, prec="f32"):
    """Returns (out, new_state); `state` is not modified in place."""
    x, dt, A, B, C, D, z, dt_bias = map(_c, (x, dt, A, B, C, D, z, dt_bias))
    st = np.array(state, dtype=np.float32, order="C", copy=True)
    batch, dim, N = st.shape
    out = np.empty_like(x)
    fn = getattr(lib(), "vms_oracle_state_update_" + prec)
    fn.restype = None
    fn(ctypes.c_int(batch), ctypes.c_int(dim), ctypes.c_int(N), _p(st), _p(x), _p(dt), _p(A), _p(B), _p(C), _p(D),
       _p(z), _p(dt_bias), ctypes.c_int(bool(dt_softplus)), _p(out))
    return out, st


# ---- complex A (SSI:111-116, 144-145; selective_scan.cpp:282-287) ------------------------------------------------
def _cpairs(a):
    """complex (..,) -> float32 (.., 2) pairs (contiguous)."""
    a = np.ascontiguousarray(np.asarray(a, dtype=np.complex64))
    return a.view(np.float32).reshape(*a.shape, 2)


def _cscan_args(u, delta, A, B, C):
    u, delta = _c(u), _c(delta)
    batch, dim, L = u.shape
    N = A.shape[1]
    var_B, var_C = not np.iscomplexobj(B), not np.iscomplexobj(C)   # variable B / C arrive as real (.., 2L) tensors
    sqB, sqC = var_B and B.ndim == 3, var_C and C.ndim == 3
    Bp = _c(_bc4(np.asarray(B), True)) if var_B else _cpairs(B)
    Cp = _c(_bc4(np.asarray(C), True)) if var_C else _cpairs(C)
    if var_B:
        assert Bp.shape[-1] == 2 * L, "variable B of a complex scan is (batch, [G,] N, 2L)"
    if var_C:
        assert Cp.shape[-1] == 2 * L, "variable C of a complex scan is (batch, [G,] N, 2L)"
    G = Bp.shape[1] if var_B else (Cp.shape[1] if var_C else 1)
    return u, delta, _cpairs(A), Bp, Cp, batch, dim, L, N, G, var_B, var_C, sqB, sqC


def cscan_fwd(u, delta, A, B, C, D=None, z=None, delta_bias=None, delta_softplus=False, prec="f32"):
    """Complex A.  -> dict(out, out_z, x=(b,d,n_chunks,2N) complex64 checkpoints, last_state=(b,d,N) complex64)."""
    u, delta, Ap, Bp, Cp, batch, dim, L, N, G, var_B, var_C, _, _ = _cscan_args(u, delta, A, B, C)
    D, z, delta_bias = map(_c, (D, z, delta_bias))
    n_chunks = (L + 2047) // 2048
    out = np.empty_like(u)
    out_z = np.empty_like(u) if z is not None else None
    x = np.zeros((batch, dim, n_chunks, 2 * N, 2), np.float32)
    last = np.empty((batch, dim, N, 2), np.float32)
    fn = getattr(lib(), "vms_oracle_cscan_fwd_" + prec)
    fn.restype = None
    fn(ctypes.c_int(batch), ctypes.c_int(dim), ctypes.c_int(L), ctypes.c_int(N), ctypes.c_int(G),
       _p(u), _p(delta), _p(Ap), _p(Bp), _p(Cp), _p(D), _p(z), _p(delta_bias),
       ctypes.c_int(var_B), ctypes.c_int(var_C), ctypes.c_int(bool(delta_softplus)),
       _p(out), _p(out_z), _p(x), _p(last))
    return dict(out=out, out_z=out_z, x=x.view(np.complex64)[..., 0], last_state=last.view(np.complex64)[..., 0])


def cscan_bwd(u, delta, A, B, C, D, z, delta_bias, dout, delta_softplus=False, prec="f32"):
    """Complex A: gradients in PyTorch's convention; dA and constant dB / dC complex64, variable dB / dC real (.., 2L)."""
    u, delta, Ap, Bp, Cp, batch, dim, L, N, G, var_B, var_C, sqB, sqC = _cscan_args(u, delta, A, B, C)
    D, z, delta_bias, dout = map(_c, (D, z, delta_bias, dout))
    du, ddelta = np.empty_like(u), np.empty_like(u)
    dA = np.zeros_like(Ap)
    dB, dC = np.zeros_like(Bp), np.zeros_like(Cp)
    dD = np.zeros(dim, np.float32) if D is not None else None
    dz = np.empty_like(u) if z is not None else None
    dbias = np.zeros(dim, np.float32) if delta_bias is not None else None
    fn = getattr(lib(), "vms_oracle_cscan_bwd_" + prec)
    fn.restype = None
    fn(ctypes.c_int(batch), ctypes.c_int(dim), ctypes.c_int(L), ctypes.c_int(N), ctypes.c_int(G),
       _p(u), _p(delta), _p(Ap), _p(Bp), _p(Cp), _p(D), _p(z), _p(delta_bias), _p(dout),
       ctypes.c_int(var_B), ctypes.c_int(var_C), ctypes.c_int(bool(delta_softplus)),
       _p(du), _p(ddelta), _p(dA), _p(dB), _p(dC), _p(dD), _p(dz), _p(dbias))
    dB = (dB[:, 0] if sqB else dB) if var_B else dB.view(np.complex64)[..., 0]
    dC = (dC[:, 0] if sqC else dC) if var_C else dC.view(np.complex64)[..., 0]
    return dict(du=du, ddelta=ddelta, dA=dA.view(np.complex64)[..., 0], dB=dB, dC=dC, dD=dD, dz=dz, ddelta_bias=dbias)
