"""Stand-in for mamba/mamba_ssm/ops/triton/layernorm.py (reference: Triton kernels, :51-503).

The build excludes Triton (BASELINE.json north_star), and the fused add+norm is a "next" row of
the scope table, so the same public names are provided here in plain PyTorch
with the reference's semantics (layer_norm_ref / rms_norm_ref, layernorm.py:19-48):
  y = norm(x + residual) * weight + bias ; optionally also return the pre-norm sum (prenorm=True),
  kept in fp32 when residual_in_fp32.
"""
import torch
import torch.nn.functional as F


def _add_residual(x, residual, residual_in_fp32):
    if residual is not None:
        x = (x.float() + residual.float()) if (residual_in_fp32 or x.dtype != residual.dtype) else x + residual
    res_out = x.float() if residual_in_fp32 else x
    return x, res_out


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


def _norm_fn(x, weight, bias, residual, eps, prenorm, residual_in_fp32, is_rms):
    out_dtype = x.dtype
    xs, res_out = _add_residual(x, residual, residual_in_fp32)
    xf = xs.float()
    if is_rms:
        y = xf * torch.rsqrt(xf.square().mean(dim=-1, keepdim=True) + eps) * weight.float()
        if bias is not None:
            y = y + bias.float()
    else:
        y = F.layer_norm(xf, xf.shape[-1:], weight.float(), bias.float() if bias is not None else None, eps)
    y = y.to(out_dtype)
    return (y, res_out) if prenorm else y


def layer_norm_fn(x, weight, bias, residual=None, eps=1e-6, prenorm=False, residual_in_fp32=False,
                  is_rms_norm=False):
    return _norm_fn(x, weight, bias, residual, eps, prenorm, residual_in_fp32, is_rms_norm)


def rms_norm_fn(x, weight, bias, residual=None, prenorm=False, residual_in_fp32=False, eps=1e-6):
    return _norm_fn(x, weight, bias, residual, eps, prenorm, residual_in_fp32, True)


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
                           residual_in_fp32=residual_in_fp32)
