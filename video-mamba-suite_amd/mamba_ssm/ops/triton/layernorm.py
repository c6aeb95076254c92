"""mamba_ssm.ops.triton.layernorm over the gfx950 kernels (reference: Triton kernels,
mamba/mamba_ssm/ops/triton/layernorm.py:51-503; the module path is kept because task code imports it).

  y = norm(x + residual) * weight + bias ; with prenorm=True also the pre-norm sum, kept in fp32 when
  residual_in_fp32.  layer_norm_fn / rms_norm_fn / RMSNorm run csrc/layer_norm.hip through layer_norm_cuda
  (no CPU path, like the reference's Triton kernels); layer_norm_ref / rms_norm_ref are the reference's
  pure-PyTorch functions (:19-48) and run anywhere.
"""
import torch
import torch.nn.functional as F

import layer_norm_cuda


def layer_norm_ref(x, weight, bias, residual=None, eps=1e-6, prenorm=False, upcast=False):
    dtype = x.dtype
    if upcast:
        weight = weight.float()
        bias = bias.float() if bias is not None else None
        x = x.float()
        residual = residual.float() if residual is not None else residual
    if residual is not None:
        x = (x + residual).to(x.dtype)
    out = F.layer_norm(x.to(weight.dtype), x.shape[-1:], weight=weight, bias=bias, eps=eps).to(dtype)
    return out if not prenorm else (out, x)


def rms_norm_ref(x, weight, bias, residual=None, eps=1e-6, prenorm=False, upcast=False):
    dtype = x.dtype
    if upcast:
        weight = weight.float()
        bias = bias.float() if bias is not None else None
        x = x.float()
        residual = residual.float() if residual is not None else residual
    if residual is not None:
        x = (x + residual).to(x.dtype)
    rstd = torch.rsqrt(x.square().mean(dim=-1, keepdim=True) + eps)
    out = x * rstd * weight
    if bias is not None:
        out = out + bias
    out = out.to(dtype)
    return out if not prenorm else (out, x)


def _rows(t):
    """(..., N) -> (M, N) with a unit column stride (a copy only when the last dim is strided)."""
    t2 = t.reshape(-1, t.shape[-1])
    return t2 if t2.stride(-1) == 1 else t2.contiguous()


class LayerNormFn(torch.autograd.Function):
    """Autograd node of the fused add + norm: same inputs, outputs and gradient slots as the reference's
    LayerNormFn (layernorm.py:380-461); the kernels live in csrc/layer_norm.hip."""

    @staticmethod
    def forward(ctx, x, weight, bias, residual=None, eps=1e-6, prenorm=False, residual_in_fp32=False,
                is_rms_norm=False):
        shape = x.shape
        if residual is not None and residual.shape != shape:
            raise ValueError("residual must have the shape of x")
        x2 = _rows(x)
        res2 = _rows(residual) if residual is not None else None
        # the pre-norm sum is stored in the residual's dtype, or in fp32 on request when there is no residual
        sum_dtype = res2.dtype if res2 is not None else (torch.float32 if residual_in_fp32 else None)
        y2, mean, rstd, presum = layer_norm_cuda.fwd(x2, weight.contiguous(), bias.contiguous() if bias is not None else None,
                                                     eps, res2, residual_dtype=sum_dtype, is_rms_norm=is_rms_norm)
        if presum is None:  # nothing added, no dtype change: the sum is x itself
            presum = x2
        ctx.save_for_backward(presum, weight, bias, mean, rstd)
        ctx.meta = (shape, eps, is_rms_norm, residual is not None, prenorm, x2.dtype)
        y = y2.reshape(shape)
        return (y, presum.reshape(shape)) if prenorm else y

    @staticmethod
    def backward(ctx, dy, *more):
        presum, weight, bias, mean, rstd = ctx.saved_tensors
        shape, eps, is_rms, had_residual, prenorm, x_dtype = ctx.meta
        dy2 = _rows(dy)
        dsum = _rows(more[0]) if prenorm else None  # gradient that arrived through the prenorm output
        dx, dw, db, dres = layer_norm_cuda.bwd(dy2, presum, weight, bias, eps, mean, rstd, dsum, had_residual, is_rms,
                                               x_dtype=x_dtype)
        return (dx.reshape(shape), dw, db, dres.reshape(shape) if had_residual else None, None, None, None, None)


def layer_norm_fn(x, weight, bias, residual=None, eps=1e-6, prenorm=False, residual_in_fp32=False,
                  is_rms_norm=False):
    return LayerNormFn.apply(x, weight, bias, residual, eps, prenorm, residual_in_fp32, is_rms_norm)


def rms_norm_fn(x, weight, bias, residual=None, prenorm=False, residual_in_fp32=False, eps=1e-6,
                is_rms_norm=True):
    return LayerNormFn.apply(x, weight, bias, residual, eps, prenorm, residual_in_fp32, is_rms_norm)


class RMSNorm(torch.nn.Module):
    def __init__(self, hidden_size, eps=1e-5, device=None, dtype=None):
        factory_kwargs = {"device": device, "dtype": dtype}
        super().__init__()
        self.eps = eps
        self.weight = torch.nn.Parameter(torch.empty(hidden_size, **factory_kwargs))
        self.register_parameter("bias", None)
        self.reset_parameters()

    def reset_parameters(self):
        torch.nn.init.ones_(self.weight)

    def forward(self, x, residual=None, prenorm=False, residual_in_fp32=False):
        return rms_norm_fn(x, self.weight, self.bias, residual=residual, eps=self.eps, prenorm=prenorm,
                           residual_in_fp32=residual_in_fp32, is_rms_norm=True)
