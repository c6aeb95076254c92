"""The Mamba block's two large projections as autograd nodes that keep the kernels' layouts.

The scan / conv kernels want (batch, channels, seqlen) with a unit seqlen stride.  The reference gets there
with `rearrange(in_proj.weight @ rearrange(hidden, "b l d -> d (b l)"), "d (b l) -> b d l")`
(mamba_ssm/modules/mamba_simple.py:144-149) and leaves both backward GEMMs to autograd, which runs the
weight gradient as ONE (channels x d_model) GEMM with K = batch * seqlen -- 32 output tiles for 256 CUs.
Here the weight gradients are split along K into a batched GEMM plus a sum (2.4x faster at
(8, 8192, 1024) on MI355X: tools/gemm_wgrad.py, tools/gemm_outproj.py), and out_proj consumes and
produces (batch, channels, seqlen) directly, so that no transpose copy sits between it and the scan.
Only library GEMMs (hipBLASLt through torch) are used; autocast behaves as for nn.Linear.

Round 3 (tools/gemm_layouts.py, profiles/r03_gemm_layouts.md): with the activations' layouts fixed by the kernels, the one
free choice is how the WEIGHT is stored.  hipBLASLt is 9 % (forward) / 21 % (input gradient) faster at these shapes when
in_proj's weight arrives as a (d_model, channels) matrix -- K-contiguous next to the K-contiguous activation operand --
so InProjFn makes that copy once per step (one 4 MB cast + transpose kernel, which autocast's weight cast cost anyway)
and uses it for both GEMMs.
"""
import torch
import torch.nn.functional as F

from mamba_ssm.ops.selective_scan_interface import custom_bwd, custom_fwd


def _autocast_dtype():
    if not torch.is_autocast_enabled():
        return None
    return torch.get_autocast_dtype("cuda") if hasattr(torch, "get_autocast_dtype") else torch.get_autocast_gpu_dtype()


def _k_splits(k_total, target=8192, most=16):
    """Number of K slices for a weight-gradient GEMM whose output is small: ~target rows per slice."""
    s = max(1, min(most, k_total // target))
    while s > 1 and k_total % s:
        s -= 1
    return s


class InProjFn(torch.autograd.Function):
    """hidden (B, L, d_model), weight (C, d_model), bias (C,) | None -> xz (B, C, L), C-slowest in memory."""

    @staticmethod
    @custom_fwd
    def forward(ctx, hidden, weight, bias):
        batch, seqlen, d_model = hidden.shape
        x2 = hidden.reshape(batch * seqlen, d_model)
        if hidden.is_cuda:
            dt = _autocast_dtype() or weight.dtype
            # W^T as its own (d_model, channels) matrix in the compute dtype: cast and transpose in one copy kernel
            wt = torch.empty(d_model, weight.shape[0], dtype=dt, device=weight.device).copy_(weight.t())
            if x2.dtype != dt:
                x2 = x2.to(dt)
            xz = (wt.t() @ x2.t()).view(weight.shape[0], batch, seqlen).permute(1, 0, 2)
        else:  # CPU (tests with checker-backed extensions): the plain formulation
            wt = weight
            xz = (weight @ x2.t()).view(weight.shape[0], batch, seqlen).permute(1, 0, 2)
        if bias is not None:
            xz = xz + bias.to(dtype=xz.dtype)[:, None]
        ctx.save_for_backward(hidden, wt)
        ctx.transposed = hidden.is_cuda
        ctx.has_bias = bias is not None
        return xz

    @staticmethod
    @custom_bwd
    def backward(ctx, dxz):
        hidden, wt = ctx.saved_tensors
        batch, seqlen, d_model = hidden.shape
        rows = batch * seqlen
        channels = wt.shape[1] if ctx.transposed else wt.shape[0]
        g2 = dxz.permute(1, 0, 2).reshape(channels, rows)       # a view when dxz has xz's layout
        x2 = hidden.reshape(rows, d_model)
        dhidden = dweight = dbias = None
        if ctx.needs_input_grad[0]:
            if ctx.transposed:
                dhidden = (g2.to(wt.dtype).t() @ wt.t()).view(batch, seqlen, d_model).to(hidden.dtype)
            else:
                dhidden = (g2.t() @ wt).view(batch, seqlen, d_model)
        if ctx.needs_input_grad[1]:
            s = _k_splits(rows)
            dweight = torch.bmm(g2.view(channels, s, rows // s).permute(1, 0, 2), x2.view(s, rows // s, d_model)).sum(0)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            dbias = g2.sum(dim=1)
        return dhidden, dweight, dbias


class OutProjFn(torch.autograd.Function):
    """y (B, C, L) with unit seqlen stride, weight (d_model, C), bias | None -> (B, L, d_model)."""

    @staticmethod
    @custom_fwd
    def forward(ctx, y, weight, bias):
        ctx.save_for_backward(y, weight)
        ctx.has_bias = bias is not None
        return F.linear(y.transpose(1, 2), weight, bias)

    @staticmethod
    @custom_bwd
    def backward(ctx, dout):
        y, weight = ctx.saved_tensors
        dy = dweight = dbias = None
        if ctx.needs_input_grad[0]:
            dy = torch.matmul(weight.t(), dout.transpose(1, 2))  # (B, C, L): the layout the scan backward reads
        if ctx.needs_input_grad[1]:
            dweight = torch.bmm(y, dout).sum(0).t()              # one K slice per batch entry
        if ctx.has_bias and ctx.needs_input_grad[2]:
            dbias = dout.sum(dim=(0, 1))
        return dy, dweight, dbias


def in_proj_fn(hidden, weight, bias=None):
    return InProjFn.apply(hidden, weight, bias)


def out_proj_fn(y, weight, bias=None):
    return OutProjFn.apply(y, weight, bias)
