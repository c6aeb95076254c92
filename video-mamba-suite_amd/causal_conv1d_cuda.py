"""Drop-in for the reference's pybind extension `causal_conv1d_cuda`
(causal-conv1d/csrc/causal_conv1d.cpp:329-333): same three entry points, argument order, return
values and checks.  Work is done by the gfx950 kernels behind include/vms_hip.h."""
import torch

import vms_hip as _k

_lib = _k.lib()

_TYPES = (torch.float32, torch.float16, torch.bfloat16)


def _check(cond, msg):
    if not cond:
        raise RuntimeError(msg)


def _common(x, weight, bias_):
    _check(x.dtype in _TYPES, f"causal_conv1d: input dtype {x.dtype} not supported")
    _check(weight.dtype in _TYPES, f"causal_conv1d: weight dtype {weight.dtype} not supported")
    _check(x.is_cuda, "x.is_cuda()")
    _check(weight.is_cuda, "weight.is_cuda()")
    width = weight.shape[-1]
    _check(tuple(weight.shape) == (x.shape[1], width), "weight must have shape (dim, width)")
    _check(2 <= width <= 4, "causal_conv1d only supports width between 2 and 4")
    if bias_ is not None:
        _check(bias_.dtype == weight.dtype, "bias.scalar_type() == weight_type")
        _check(bias_.is_cuda, "bias.is_cuda()")
        _check(bias_.stride(-1) == 1, "bias.stride(-1) == 1")
        _check(tuple(bias_.shape) == (x.shape[1],), "bias must have shape (dim,)")


def causal_conv1d_fwd(x, weight, bias_, silu_activation, reverse=False, reverse_from=0):
    """-> out   (causal_conv1d.cpp:130-189)
    reverse / reverse_from (extensions, vms_hip.h): anti-causal for every batch entry / for the entries >= reverse_from."""
    ext = _k.ext()
    if ext is not None:   # compiled binding: same checks / allocations / launch in C++
        return ext.conv_fwd(x, weight, bias_, bool(silu_activation), bool(reverse), int(reverse_from))
    _check(x.dim() == 3, "x must be (batch, dim, seqlen)")
    _common(x, weight, bias_)
    _check(x.stride(2) == 1 or x.stride(1) == 1, "x.stride(2) == 1 || x.stride(1) == 1")
    if x.stride(1) == 1 and x.stride(2) > 1:
        _check(x.shape[1] % 8 == 0, "causal_conv1d only supports channel dimension divisible by 8 for now")
    out = torch.empty_like(x)  # preserve_format keeps the unit-stride axis of x
    if (x.stride(1) == 1 and x.stride(2) > 1) and out.stride(1) != 1:
        out = torch.empty(x.shape[0], x.shape[2], x.shape[1], dtype=x.dtype, device=x.device).transpose(1, 2)
    _k.conv_fwd(x, weight, bias_, out, silu_activation, reverse, reverse_from)
    return out


def causal_conv1d_bwd(x, weight, bias_, dout, dx_, silu_activation, reverse=False, zeroed=None, accumulate_dx=False,
                      reverse_from=0):
    """-> [dx, dweight, dbias]   (causal_conv1d.cpp:191-268)"""
    ext = _k.ext()
    if ext is not None:
        return ext.conv_bwd(x, weight, bias_, dout, dx_, bool(silu_activation), bool(reverse), zeroed, bool(accumulate_dx),
                            int(reverse_from))
    _check(x.dim() == 3, "x must be (batch, dim, seqlen)")
    _common(x, weight, bias_)
    _check(dout.is_cuda, "dout.is_cuda()")
    _check(tuple(dout.shape) == tuple(x.shape), "dout must have the shape of x")
    _check(x.stride(2) == 1 or x.stride(1) == 1, "x.stride(2) == 1 || x.stride(1) == 1")
    channel_last = x.stride(1) == 1 and x.stride(2) > 1
    if not channel_last and dout.stride(2) != 1:
        dout = dout.contiguous()
    if channel_last and dout.stride(1) != 1:
        dout = dout.transpose(-1, -2).contiguous().transpose(-1, -2)
    if dx_ is not None:
        dx = dx_
        _check(dx.dtype == x.dtype, "dx.scalar_type() == input_type")
        _check(dx.is_cuda, "dx.is_cuda()")
        _check(tuple(dx.shape) == tuple(x.shape), "dx must have the shape of x")
        _check(dx.stride(1) == 1 if channel_last else dx.stride(2) == 1, "dx must have x's unit-stride axis")
    else:
        dx = torch.empty_like(x)
        if channel_last and dx.stride(1) != 1:
            dx = torch.empty(x.shape[0], x.shape[2], x.shape[1], dtype=x.dtype, device=x.device).transpose(1, 2)
    if zeroed is not None:  # extension: the caller's zero fp32 scratch holds the two atomics targets
        nw, nb = weight.numel(), bias_.numel() if bias_ is not None else 0
        _check(zeroed.dtype == torch.float32 and zeroed.is_cuda and zeroed.dim() == 1 and zeroed.is_contiguous()
               and zeroed.numel() >= nw + nb, "zeroed must be a flat float32 tensor of weight.numel() + bias.numel() elements")
        dweight = zeroed[:nw].view(weight.shape)
        dbias = zeroed[nw:nw + nb] if bias_ is not None else None
    else:
        dweight = torch.zeros_like(weight, dtype=torch.float32)
        dbias = torch.zeros_like(bias_, dtype=torch.float32) if bias_ is not None else None
    _check(not accumulate_dx or dx_ is not None, "accumulate_dx needs the dx tensor to add to")
    _k.conv_bwd(x, weight, bias_, dout, dx, dweight, dbias, silu_activation, reverse, bool(accumulate_dx), reverse_from)
    return [dx, dweight.to(weight.dtype), dbias.to(bias_.dtype) if bias_ is not None else None]


def causal_conv1d_update(x, conv_state, weight, bias_, silu_activation):
    """-> out; conv_state is updated in place   (causal_conv1d.cpp:270-327)"""
    ext = _k.ext()
    if ext is not None:
        return ext.conv_update(x, conv_state, weight, bias_, bool(silu_activation))
    _check(x.dim() == 2, "x must be (batch, dim)")
    _common(x, weight, bias_)
    _check(conv_state.dtype == x.dtype, "conv_state.scalar_type() == input_type")
    _check(conv_state.is_cuda, "conv_state.is_cuda()")
    _check(tuple(conv_state.shape) == (x.shape[0], x.shape[1], weight.shape[-1]),
           "conv_state must have shape (batch, dim, width)")
    out = torch.empty_like(x)
    _k.conv_update(x, conv_state, weight, bias_, out, silu_activation)
    return out
