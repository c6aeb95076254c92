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


def _seq_or_channel_major(t):
    """The kernels take (B, D, L) with a unit stride along L or along D; anything else is compacted."""
    return t if (t.stride(2) == 1 or t.stride(1) == 1) else t.contiguous()


class CausalConv1dFn(torch.autograd.Function):
    """out = act(conv1d_causal_depthwise(x, weight) + bias); x may be (B, D, L) with unit L stride
    or channel-last (unit D stride)."""

    @staticmethod
    def forward(ctx, x, weight, bias=None, activation=None):
        ctx.silu = _act_flag(activation)
        x = _seq_or_channel_major(x)
        bias = None if bias is None else bias.contiguous()
        ctx.save_for_backward(x, weight, bias)
        return causal_conv1d_cuda.causal_conv1d_fwd(x, weight, bias, ctx.silu)

    @staticmethod
    def backward(ctx, dout):
        x, weight, bias = ctx.saved_tensors
        # no pre-allocated dx here (the fused Mamba node passes one to write into a slice of dxz)
        dx, dweight, dbias = causal_conv1d_cuda.causal_conv1d_bwd(x, weight, bias, _seq_or_channel_major(dout), None,
                                                                  ctx.silu)
        return dx, dweight, dbias, None


def causal_conv1d_fn(x, weight, bias=None, activation=None):
    """Depthwise causal conv1d (+ bias, + SiLU) with autograd.
    x (batch, dim, seqlen), weight (dim, width), bias (dim,) or None, activation None | "silu" | "swish"
    -> (batch, dim, seqlen)"""
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
    """Decode step: x (batch, dim), conv_state (batch, dim, width) rolled in place, weight (dim, width),
    bias (dim,) or None -> (batch, dim)"""
    return causal_conv1d_cuda.causal_conv1d_update(x, conv_state, weight, bias, _act_flag(activation))


def causal_conv1d_update_ref(x, conv_state, weight, bias=None, activation=None):
    """Pure-PyTorch decode step: shift the window left by one, append x, dot with the taps."""
    silu = _act_flag(activation)
    if conv_state.shape != (*x.shape, weight.shape[1]) or weight.shape[0] != x.shape[1]:
        raise ValueError("conv_state must be (batch, dim, width) and weight (dim, width)")
    window = torch.cat([conv_state[:, :, 1:], x[:, :, None].to(conv_state.dtype)], dim=-1)
    conv_state.copy_(window)
    y = (window * weight).sum(dim=-1)
    y = y if bias is None else y + bias
    return (F.silu(y) if silu else y).to(dtype=x.dtype)
