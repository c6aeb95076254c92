"""`causal_conv1d` public interface on top of the MI355X HIP kernels.

Mirrors causal-conv1d/causal_conv1d/causal_conv1d_interface.py of the reference: the autograd
node CausalConv1dFn / causal_conv1d_fn (:10-46), the decode step causal_conv1d_update (:68-84)
and the pure-PyTorch references causal_conv1d_ref (:49-65) / causal_conv1d_update_ref (:87-104)
that the reference ships as part of its API (they run on any device and are what its tests
compare against).
"""
import torch
import torch.nn.functional as F

import causal_conv1d_cuda

_ACTIVATIONS = (None, "silu", "swish")


def _act_flag(activation):
    if activation not in _ACTIVATIONS:
        raise NotImplementedError("activation must be None, silu, or swish")
    return activation is not None


class CausalConv1dFn(torch.autograd.Function):
    """out = act(conv1d_causal_depthwise(x, weight) + bias); x may be (B, D, L) with unit L stride
    or channel-last (unit D stride)."""

    @staticmethod
    def forward(ctx, x, weight, bias=None, activation=None):
        silu = _act_flag(activation)
        if x.stride(2) != 1 and x.stride(1) != 1:
            x = x.contiguous()
        if bias is not None:
            bias = bias.contiguous()
        ctx.save_for_backward(x, weight, bias)
        ctx.activation = silu
        return causal_conv1d_cuda.causal_conv1d_fwd(x, weight, bias, silu)

    @staticmethod
    def backward(ctx, dout):
        x, weight, bias = ctx.saved_tensors
        if dout.stride(2) != 1 and dout.stride(1) != 1:
            dout = dout.contiguous()
        # dx_ = None: the extension allocates dx (a caller may pre-allocate it to write into a
        # slice of a larger gradient, as the fused Mamba node does)
        dx, dweight, dbias = causal_conv1d_cuda.causal_conv1d_bwd(x, weight, bias, dout, None, ctx.activation)
        return dx, dweight, (dbias if bias is not None else None), None


def causal_conv1d_fn(x, weight, bias=None, activation=None):
    """
    x: (batch, dim, seqlen)
    weight: (dim, width)
    bias: (dim,)
    activation: either None or "silu" or "swish"

    out: (batch, dim, seqlen)
    """
    return CausalConv1dFn.apply(x, weight, bias, activation)


def causal_conv1d_ref(x, weight, bias=None, activation=None):
    """Pure-PyTorch statement of the op (any device): left-padded grouped conv, cut to seqlen."""
    silu = _act_flag(activation)
    in_dtype = x.dtype
    dim, width = weight.shape
    seqlen = x.shape[-1]
    y = F.conv1d(x.to(weight.dtype), weight[:, None, :], bias, padding=width - 1, groups=dim)[..., :seqlen]
    if silu:
        y = F.silu(y)
    return y.to(dtype=in_dtype)


def causal_conv1d_update(x, conv_state, weight, bias=None, activation=None):
    """
    x: (batch, dim)
    conv_state: (batch, dim, width), updated in place
    weight: (dim, width)
    bias: (dim,)

    out: (batch, dim)
    """
    return causal_conv1d_cuda.causal_conv1d_update(x, conv_state, weight, bias, _act_flag(activation))


def causal_conv1d_update_ref(x, conv_state, weight, bias=None, activation=None):
    """Pure-PyTorch decode step: shift the window left by one, append x, dot with the taps."""
    silu = _act_flag(activation)
    in_dtype = x.dtype
    batch, dim = x.shape
    width = weight.shape[1]
    assert conv_state.shape == (batch, dim, width)
    assert weight.shape == (dim, width)
    conv_state.copy_(torch.cat([conv_state[:, :, 1:], x[:, :, None].to(conv_state.dtype)], dim=-1))
    y = (conv_state * weight).sum(dim=-1)
    if bias is not None:
        y = y + bias
    if silu:
        y = F.silu(y)
    return y.to(dtype=in_dtype)
