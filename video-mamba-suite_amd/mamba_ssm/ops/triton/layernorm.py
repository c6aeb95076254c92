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


class LayerNormFn(torch.autograd.Function):
    """Host logic of the reference's LayerNormFn (layernorm.py:380-461) over the HIP kernels."""

    @staticmethod
    def forward(ctx, x, weight, bias, residual=None, eps=1e-6, prenorm=False, residual_in_fp32=False,
                is_rms_norm=False):
        x_shape_og = x.shape
        x = x.reshape(-1, x.shape[-1])
        if x.stride(-1) != 1:
            x = x.contiguous()
        if residual is not None:
            assert residual.shape == x_shape_og
            residual = residual.reshape(-1, residual.shape[-1])
            if residual.stride(-1) != 1:
                residual = residual.contiguous()
        weight = weight.contiguous()
        if bias is not None:
            bias = bias.contiguous()
        residual_dtype = residual.dtype if residual is not None else (torch.float32 if residual_in_fp32 else None)
        y, mean, rstd, residual_out = layer_norm_cuda.fwd(x, weight, bias, eps, residual,
                                                          residual_dtype=residual_dtype, is_rms_norm=is_rms_norm)
        if residual_out is None:
            residual_out = x  # nothing was added and no dtype change: the pre-norm sum is x itself
        ctx.save_for_backward(residual_out, weight, bias, mean, rstd)
        ctx.x_shape_og = x_shape_og
        ctx.eps = eps
        ctx.is_rms_norm = is_rms_norm
        ctx.has_residual = residual is not None
        ctx.prenorm = prenorm
        ctx.x_dtype = x.dtype
        y = y.reshape(x_shape_og)
        return y if not prenorm else (y, residual_out.reshape(x_shape_og))

    @staticmethod
    def backward(ctx, dy, *args):
        x, weight, bias, mean, rstd = ctx.saved_tensors
        dy = dy.reshape(-1, dy.shape[-1])
        if dy.stride(-1) != 1:
            dy = dy.contiguous()
        assert dy.shape == x.shape
        dresidual = None
        if ctx.prenorm:
            dresidual = args[0].reshape(-1, args[0].shape[-1])
            if dresidual.stride(-1) != 1:
                dresidual = dresidual.contiguous()
            assert dresidual.shape == x.shape
        dx, dw, db, dresidual_in = layer_norm_cuda.bwd(dy, x, weight, bias, ctx.eps, mean, rstd, dresidual,
                                                       ctx.has_residual, ctx.is_rms_norm, x_dtype=ctx.x_dtype)
        return (dx.reshape(ctx.x_shape_og), dw, db,
                dresidual_in.reshape(ctx.x_shape_og) if ctx.has_residual else None, None, None, None, None)


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
