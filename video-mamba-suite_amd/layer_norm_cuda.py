"""Extension-level mirror of the reference's fused add + norm launchers `_layer_norm_fwd` / `_layer_norm_bwd`
(mamba/mamba_ssm/ops/triton/layernorm.py:122-173, 291-377): same arguments, same returned tuples, same
allocation rules; the work is done by the gfx950 kernels behind the C ABI (include/vms_hip.h,
csrc/layer_norm.hip).  No CPU path: CPU tensors raise, a missing library is an ImportError."""
import torch

import vms_hip as _k

_lib = _k.lib()


def _check(cond, msg):
    if not cond:
        raise RuntimeError(msg)


def fwd(x, weight, bias, eps, residual=None, out_dtype=None, residual_dtype=None, is_rms_norm=False):
    """-> (y, mean, rstd, residual_out)   (layernorm.py:122-173)"""
    if residual is not None:
        residual_dtype = residual.dtype
    _check(x.dim() == 2 and x.stride(-1) == 1, "x must be (M, N) with a unit last stride")
    M, N = x.shape
    if residual is not None:
        _check(residual.stride(-1) == 1 and tuple(residual.shape) == (M, N), "residual must match x")
    _check(tuple(weight.shape) == (N,) and weight.stride(-1) == 1, "weight must have shape (N,)")
    if bias is not None:
        _check(tuple(bias.shape) == (N,) and bias.stride(-1) == 1, "bias must have shape (N,)")
    _check(out_dtype is None or out_dtype == x.dtype, "out_dtype other than the input dtype is not supported")
    _check(N * x.element_size() < 65536, "This layer norm doesn't support feature dim >= 64KB.")
    y = torch.empty_like(x)
    if residual is not None or (residual_dtype is not None and residual_dtype != x.dtype):
        residual_out = torch.empty(M, N, device=x.device, dtype=residual_dtype)
    else:
        residual_out = None
    mean = torch.empty((M,), dtype=torch.float32, device=x.device) if not is_rms_norm else None
    rstd = torch.empty((M,), dtype=torch.float32, device=x.device)
    _k.norm_fwd(x, residual, weight.float(), bias.float() if bias is not None else None, y, residual_out, mean,
                rstd, eps, is_rms_norm)
    return y, mean, rstd, residual_out


def bwd(dy, x, weight, bias, eps, mean, rstd, dresidual=None, has_residual=False, is_rms_norm=False,
        x_dtype=None):
    """x: the forward's saved pre-norm sum (residual_out, or x).  -> (dx, dw, db, dresidual_in)
    (layernorm.py:291-377)"""
    M, N = x.shape
    _check(x.stride(-1) == 1 and dy.stride(-1) == 1 and tuple(dy.shape) == (M, N), "dy must match x")
    if dresidual is not None:
        _check(dresidual.stride(-1) == 1 and tuple(dresidual.shape) == (M, N), "dresidual must match x")
        _check(dresidual.dtype == x.dtype, "dresidual must have the dtype of the saved pre-norm sum")
    dx = torch.empty_like(x) if x_dtype is None else torch.empty(M, N, dtype=x_dtype, device=x.device)
    _check(dy.dtype == dx.dtype, "dy must have the dtype of dx")
    dresidual_in = torch.empty_like(x) if has_residual and dx.dtype != x.dtype else None
    n_part = _k.norm_bwd_partials(M, N)
    dw_p = torch.empty((n_part, N), dtype=torch.float32, device=x.device)
    db_p = torch.empty((n_part, N), dtype=torch.float32, device=x.device) if bias is not None else None
    _k.norm_bwd(x, dy, weight.float(), mean, rstd, dresidual, dx, dresidual_in, dw_p, db_p, is_rms_norm)
    if N % 4 == 0 and (bias is None or bias.dtype == weight.dtype):   # one launch sums both arrays (vms_layer_norm_bwd_finish)
        dw = torch.empty(N, dtype=weight.dtype, device=x.device)
        db = torch.empty(N, dtype=bias.dtype, device=x.device) if bias is not None else None
        _k.norm_bwd_finish(dw_p, db_p, dw, db)
    else:
        dw = dw_p.sum(0).to(weight.dtype)
        db = db_p.sum(0).to(bias.dtype) if bias is not None else None
    if has_residual and dx.dtype == x.dtype:  # no separate tensor needed in this case (layernorm.py:373-375)
        dresidual_in = dx
    return dx, dw, db, dresidual_in
