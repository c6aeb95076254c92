"""`mamba_ssm.ops.selective_scan_interface` on top of the MI355X HIP kernels.

Public surface (signatures, argument meaning, returned gradients) mirrors the reference file
mamba/mamba_ssm/ops/selective_scan_interface.py:
  selective_scan_fn / SelectiveScanFn                      (:14-83)
  mamba_inner_fn_no_out_proj / MambaInnerFnNoOutProj       (:155-289, 627-633)  <- what the suite runs
  mamba_inner_fn / MambaInnerFn                            (:292-434, 606-614)
  bimamba_inner_fn / BiMambaInnerFn                        (:437-603, 616-624)
  selective_scan_ref, mamba_inner_ref, bimamba_inner_ref   (:86-152, 636-709)   pure PyTorch, any device

The three fused nodes share one implementation here (`_inner_forward` / `_inner_backward`):
conv1d+SiLU -> x_proj GEMM -> dt_proj GEMM -> selective scan (+z gate) [-> out_proj], with the
reference's recompute policy (checkpoint_lvl=1: conv output and delta are rebuilt in backward,
SSI:218-219, 238-241) and its layout contract (delta and the scan output are "d-slowest",
dx/dz are written straight into the halves of one dxz buffer, SSI:244-248, 281-283).
"""
import os

import torch
import torch.nn.functional as F

import causal_conv1d_cuda
import selective_scan_cuda
import vms_hip as _vms
from causal_conv1d import causal_conv1d_fn

try:  # torch >= 2.4
    from torch.amp import custom_bwd as _custom_bwd, custom_fwd as _custom_fwd

    def custom_fwd(fn):
        return _custom_fwd(fn, device_type="cuda")

    def custom_bwd(fn):
        return _custom_bwd(fn, device_type="cuda")
except ImportError:  # pragma: no cover
    from torch.cuda.amp import custom_bwd, custom_fwd


def _last_dim_contiguous(t):
    if t is None or t.stride(-1) == 1:
        return t
    if t.shape[-1] == 1:   # a size-1 axis may carry any stride (and .contiguous() keeps it): restate it as 1
        return t.as_strided(t.shape, t.stride()[:-1] + (1,), t.storage_offset())
    return t.contiguous()


# =================================================================================================
# selective_scan_fn
# =================================================================================================
class SelectiveScanFn(torch.autograd.Function):

    @staticmethod
    def forward(ctx, u, delta, A, B, C, D=None, z=None, delta_bias=None, delta_softplus=False,
                return_last_state=False):
        u, delta, B, C, z = map(_last_dim_contiguous, (u, delta, B, C, z))
        if D is not None:
            D = D.contiguous()
        # (batch, dstate, L) means one group: promote to (batch, 1, dstate, L) for the kernel
        ctx.squeeze_B = B.dim() == 3
        ctx.squeeze_C = C.dim() == 3
        if ctx.squeeze_B:
            B = B.unsqueeze(1)
        if ctx.squeeze_C:
            C = C.unsqueeze(1)
        out, x, *rest = selective_scan_cuda.fwd(u, delta, A, B, C, D, z, delta_bias, delta_softplus,
                                                for_backward=any(ctx.needs_input_grad))
        ctx.delta_softplus = delta_softplus
        ctx.has_z = z is not None
        ctx.has_D = D is not None
        ctx.has_bias = delta_bias is not None
        ctx.save_for_backward(u, delta, A, B, C, D, z, delta_bias, x, out if ctx.has_z else None)
        result = rest[0] if ctx.has_z else out
        if not return_last_state:
            return result
        last_state = x[:, :, -1, 1::2]  # (batch, dim, dstate): state after the last chunk
        ctx.mark_non_differentiable(last_state)
        return result, last_state

    @staticmethod
    def backward(ctx, dout, *unused):
        u, delta, A, B, C, D, z, delta_bias, x, out = ctx.saved_tensors
        dout = _last_dim_contiguous(dout)
        du, ddelta, dA, dB, dC, dD, ddelta_bias, *rest = selective_scan_cuda.bwd(
            u, delta, A, B, C, D, z, delta_bias, dout, x, out, None, ctx.delta_softplus, False)
        dz = rest[0] if ctx.has_z else None
        if ctx.squeeze_B:
            dB = dB.squeeze(1)
        if ctx.squeeze_C:
            dC = dC.squeeze(1)
        return (du, ddelta, dA, dB, dC, dD if ctx.has_D else None, dz,
                ddelta_bias if ctx.has_bias else None, None, None)


def selective_scan_fn(u, delta, A, B, C, D=None, z=None, delta_bias=None, delta_softplus=False,
                      return_last_state=False):
    """if return_last_state is True, returns (out, last_state)
    last_state has shape (batch, dim, dstate). Note that the gradient of the last state is
    not considered in the backward pass.
    """
    # complex A (the reference's weight_t = complex<float> instantiations, selective_scan.cpp:282-287): its own HIP kernels
    # behind the same extension entry points (csrc/selective_scan_complex.hip)
    return SelectiveScanFn.apply(u, delta, A, B, C, D, z, delta_bias, delta_softplus, return_last_state)


def selective_scan_ref(u, delta, A, B, C, D=None, z=None, delta_bias=None, delta_softplus=False,
                       return_last_state=False):
    """Pure-PyTorch statement of the selective scan (runs on any device, differentiable).

    u, delta, z: (B, D, L);  A: (D, N) real or complex;  D, delta_bias: (D,)
    B, C: (D, N) constant, or input dependent (B, N, L) / (B, G, N, L) (last dim 2L if A is complex)
    returns out (B, D, L) in u's dtype [, last_state (B, D, N)]
    """
    in_dtype = u.dtype
    u = u.float()
    dt = delta.float()
    if delta_bias is not None:
        dt = dt + delta_bias.float()[:, None]
    if delta_softplus:
        dt = F.softplus(dt)
    batch, dim, seqlen = u.shape
    dstate = A.shape[1]
    var_B, var_C = B.dim() >= 3, C.dim() >= 3

    def per_dim(M, var):
        """-> accessor l -> (batch|1, dim, dstate) slice of B or C at position l"""
        if not var:
            Mc = M if A.is_complex() else M.float()
            return lambda l: Mc[None]
        M = M.float()
        if A.is_complex():  # interleaved (re, im) pairs along L
            M = torch.view_as_complex(M.reshape(*M.shape[:-1], seqlen, 2).contiguous())
        if M.dim() == 3:
            return lambda l: M[:, None, :, l]
        rep = dim // M.shape[1]
        return lambda l: M[:, :, :, l].repeat_interleave(rep, dim=1)

    B_at, C_at = per_dim(B, var_B), per_dim(C, var_C)
    state = A.new_zeros((batch, dim, dstate))
    ys = []
    for l in range(seqlen):
        dt_l = dt[:, :, l, None]
        state = torch.exp(dt_l * A) * state + (dt_l * u[:, :, l, None]) * B_at(l)
        y = (state * C_at(l)).sum(dim=-1)
        ys.append(2 * y.real if y.is_complex() else y)
    y = torch.stack(ys, dim=2)
    out = y if D is None else y + u * D[:, None]
    if z is not None:
        out = out * F.silu(z.float())
    out = out.to(dtype=in_dtype)
    return (out, state) if return_last_state else out


# =================================================================================================
# fused Mamba inner nodes
# =================================================================================================
def _inner_ext_module():
    return _vms.ext()


def _inner_ext(xz, out_proj, A_b, B, C, B_proj_bias, C_proj_bias):
    """The compiled one-call form of the node (vms_torch.cpp inner_fwd / inner_bwd) when it applies: GPU tensors, the
    binding built, input-dependent B / C without projection biases, one direction, no fused out_proj -- what every module
    of the suite runs.  debug.no_inner_ext keeps the Python statement of the node (tests compare the two)."""
    if (out_proj is not None or A_b is not None or B is not None or C is not None or B_proj_bias is not None
            or C_proj_bias is not None or not xz.is_cuda or _vms.debug.no_inner_ext):
        return None
    ext = _vms.ext()
    if ext is None or not hasattr(ext, "inner_fwd"):
        return None
    return ext


# Bit 16 of proj_flags: x_dbl = x_proj.weight @ conv1d_out and dx_dbl[:R] = dt_proj.weight^T @ ddelta -- the node's two products
# that contract over the channels -- as one streaming pass on the matrix cores (vms_proj_kred) instead of library GEMMs.
_PROJ_KRED = not _vms.debug.no_proj_kred


def _mfma_proj(d_inner=None, dt_rank=None):
    """proj_flags of the compiled node.  Bit 1: the node's small dt_proj products -- delta = W_dt x_dbl[:R] and its weight
    gradient -- on the hand-written matrix-core kernels (csrc/inner_proj.hip) instead of the library's GEMMs.  At the benchmark
    shape (d_inner 1024, dt_rank 64) the two are at parity (4.20-4.25 vs 4.26 ms per block step, profiles/r03_small_gemms.md, r04y);
    where dt_rank is not a multiple of 64 or d_inner not one of 256 -- every d_model 768 / 512 / 384 config of the suite -- the
    library picks poor tiles (38 and 36 us per call at (8, 768, 3136) for 1.9 GFLOP over 39 MB) and the hand-written ones win:
    12-layer stack 20.3 -> 19.5 ms, long video 3.90 -> 3.88 (profiles/r04_mfma_proj_ab.txt).  debug.mfma_proj = True / False forces.
    Bit 2 (default on; debug.no_fused_tail clears it): the backward's tail -- dx_proj.weight, dconv1d_out += W_x^T dx_dbl and the
    conv1d backward -- as ONE pass over the activations (vms_proj_conv_bwd) instead of three kernels and seven."""
    if _vms.debug.mfma_proj is not None:
        mfma = bool(_vms.debug.mfma_proj)
    else:
        mfma = d_inner is not None and (d_inner % 256 != 0 or dt_rank % 64 != 0)
    return (1 if mfma else 0) | (0 if _vms.debug.no_fused_tail else 2) | (16 if _PROJ_KRED else 0)


def _for_backward(ctx):
    """Will this node's backward run?  (then the forward scan leaves 8-element checkpoints for it, vms_hip.h x_has_sub == 3:
    537 MB at (8, 1024, 8192) instead of 34; an inference forward keeps the small layout.)  Contexts without the flag -- the
    per-direction stand-ins of the bidirectional node -- inherit `for_backward` from their parent."""
    nig = getattr(ctx, "needs_input_grad", None)
    return getattr(ctx, "for_backward", True) if nig is None else any(nig)


def _x_flags(ctx, xz, d_state):
    """inner_fwd proj_flags bits 4 / 8: no backward / keep the 128-element checkpoints (the checkpoint policy of
    vms_hip.x_mode_for_shape: VMS_X_LAYOUT, set_x_layout_policy, the modules' scan_checkpoints=, or memory-aware "auto")."""
    fb = _for_backward(ctx)
    coarse = fb and _vms.x_mode_for_shape(xz.shape[0], xz.shape[1] // 2, xz.shape[2], d_state, xz.device) == 1
    return (0 if fb else 4) | (8 if coarse else 0)


def _autocast_weights(*ws):
    if not torch.is_autocast_enabled():
        return ws
    dt = torch.get_autocast_dtype("cuda") if hasattr(torch, "get_autocast_dtype") else torch.get_autocast_gpu_dtype()
    return tuple(w.to(dtype=dt) if w is not None else None for w in ws)


def _conv_and_projections(xz, conv_w, conv_b, x_proj_w, dt_proj_w, reverse=False):
    """conv1d+SiLU on the x half of xz, then the two small GEMMs.
    Returns conv_out (b, d, l), x_dbl (b*l, R+2N), delta (b, d, l) with d as the slowest axis.
    reverse: anti-causal conv (the kernels' right-to-left mode; the GEMMs are position-wise)."""
    batch, _, L = xz.shape
    d_inner = conv_w.shape[0]
    R = dt_proj_w.shape[1]
    conv_out = causal_conv1d_cuda.causal_conv1d_fwd(xz[:, :d_inner], conv_w, conv_b, True, reverse)
    x_dbl = F.linear(conv_out.transpose(1, 2).reshape(batch * L, d_inner), x_proj_w)
    delta = _delta_from(x_dbl, dt_proj_w, batch, L)
    return conv_out, x_dbl, delta


def _delta_from(x_dbl, dt_proj_w, batch, L):
    R = dt_proj_w.shape[1]
    # (d, b*l) GEMM output viewed as (b, d, l): strides (l, b*l, 1) -- no transpose copy (SSI:178-182)
    return (dt_proj_w @ x_dbl[:, :R].t()).view(dt_proj_w.shape[0], batch, L).permute(1, 0, 2)


def _bc_from_x_dbl(x_dbl, lo, hi, bias, batch, L, is_complex):
    """x_dbl[:, lo:hi] (b*l, n) -> (b, 1, n, l) contiguous (complex: (b, 1, n/2, 2l))."""
    M = x_dbl[:, lo:hi]
    if bias is not None:
        M = M + bias.to(dtype=M.dtype)
    n = hi - lo
    if not is_complex:
        return M.view(batch, L, n).permute(0, 2, 1).contiguous().unsqueeze(1)
    return M.view(batch, L, n // 2, 2).permute(0, 2, 1, 3).reshape(batch, 1, n // 2, 2 * L).contiguous()


def _bc_grad_to_x_dbl(dM, batch, L, is_complex):
    """inverse layout of _bc_from_x_dbl for a gradient: (b, 1, n, l) -> (b*l, n)."""
    if not is_complex:
        return dM.squeeze(1).permute(0, 2, 1).reshape(batch * L, -1)
    n2 = dM.shape[2]
    return dM.view(batch, n2, L, 2).permute(0, 2, 1, 3).reshape(batch * L, 2 * n2)


def _flip_l(t):
    return t.flip([-1])


# ---- projections in the sequence-major layout ---------------------------------------------------------
# The reference runs x_proj / dt_proj on "(b l) d" matrices (SSI:175-182), which turns conv_out into a
# transposed GEMM operand and leaves B / C needing "(b l) n -> b n l" copies.  The same products taken per
# batch on the (d, l) matrices the kernels already hold are plain row-major GEMMs whose outputs ARE the
# layouts the scan wants:  x_dblT = W_x @ conv_out -> (b, R+2N, l): rows R..R+N are B, the last N are C
# (unit seqlen stride, no copy);  delta = W_dt @ x_dblT[:, :R] -> (b, d, l).  Same arithmetic, different
# summation grouping inside the GEMMs only.
def _proj_T(conv_out, x_proj_w, dt_proj_w):
    R = dt_proj_w.shape[1]
    x_dblT = torch.matmul(x_proj_w, conv_out)           # (b, R+2N, l)
    delta = torch.matmul(dt_proj_w, x_dblT[:, :R])      # (b, d, l)
    return x_dblT, delta


def _bc_from_x_dblT(x_dblT, lo, hi, bias):
    M = x_dblT[:, lo:hi]
    if bias is not None:
        M = M + bias.to(dtype=M.dtype)[None, :, None]
    return M.unsqueeze(1)                                # (b, 1, n, l), unit seqlen stride


def _mask_padding(delta, seq_valid, delta_softplus=True):
    """delta (pre-softplus) with -inf behind seq_valid: softplus gives exactly 0 there, so a = exp(0 A) = 1 and b = 0 u B = 0 (the
    recurrence passes its state through unchanged, in either direction), and d softplus = 0 (no gradient reaches the padding's
    delta; du, dB, dC, dA carry the factor delta = 0).  In place: delta is this node's own product."""
    if seq_valid and seq_valid < delta.shape[-1]:
        assert delta_softplus, "sequence padding needs delta_softplus (softplus(-inf) = 0)"
        delta[..., seq_valid:].fill_(float("-inf"))
    return delta


def _inner_forward(ctx, xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight,
                   out_proj, A, A_b, B, C, D, delta_bias, B_proj_bias, C_proj_bias, delta_softplus,
                   checkpoint_lvl, reverse=False, out_z_into=None, reverse_from=0, conv_out=None, x_dbl=None, seq_valid=0):
    """out_proj: None (no projection) or (weight, bias).  A_b: None or the reverse-direction A.
    seq_valid > 0 (extension; the mixer's padding of ragged sequences to whole vectors, modules/_core.py): positions >= seq_valid
    of xz are zero padding; the pre-softplus delta is set to -inf there, which makes them identity steps of both directions'
    recurrences and every gradient through them exactly zero (_mask_padding).
    reverse: the whole node runs right-to-left (== flip o node o flip, without the copies).
    reverse_from > 0: the batch entries >= reverse_from run right-to-left, the others left-to-right (vms_hip.h ABI v5).
    conv_out: this direction's conv1d + SiLU output when the caller already has it (both directions of a block from one pass
    over x: vms_causal_conv1d_fwd_dual)."""
    assert checkpoint_lvl in (0, 1)
    batch, _, L = xz.shape
    R = delta_proj_weight.shape[1]
    is_complex = A.is_complex()
    d_state = A.shape[-1] * (2 if is_complex else 1)
    out_w = out_bias = None
    # the weight gradients' partial sums over the batch are added in the PARAMETERS' dtype (fp32 under autocast): autograd
    # then has nothing to cast
    ctx.w_dtype = getattr(ctx, "w_dtype_override", None) or x_proj_weight.dtype
    if out_proj is not None:
        x_proj_weight, delta_proj_weight, out_w, out_bias = _autocast_weights(
            x_proj_weight, delta_proj_weight, out_proj[0], out_proj[1])
    else:
        x_proj_weight, delta_proj_weight = _autocast_weights(x_proj_weight, delta_proj_weight)
    xz = _last_dim_contiguous(xz)
    conv_w = conv1d_weight.squeeze(1)  # (d, 1, w) -> (d, w)
    conv_b = conv1d_bias.contiguous() if conv1d_bias is not None else None
    d_inner = conv_w.shape[0]
    z = xz[:, d_inner:]
    if is_complex:   # the public functions route a complex A to the composition of the ops (_complex_inner)
        raise RuntimeError("the fused inner node is built for a real A")
    ctx.is_variable_B, ctx.is_variable_C = B is None, C is None
    ctx.has_B_proj_bias, ctx.has_C_proj_bias = B_proj_bias is not None, C_proj_bias is not None
    ext = _inner_ext(xz, out_proj, A_b, B, C, B_proj_bias, C_proj_bias)
    ctx.fast = ext is not None
    if ext is not None:
        # the whole node in one host call (csrc/torch_binding/vms_torch.cpp inner_fwd): same ops, same order
        D = D.contiguous() if D is not None else None
        out_z, conv_out, x_dbl, delta, ckpt, out = ext.inner_fwd(
            xz, conv_w, conv_b, x_proj_weight, delta_proj_weight, A, D, delta_bias, bool(delta_softplus), bool(reverse),
            out_z_into, _vms.scan_impl_from_env(), _vms.segments_from_env("VMS_FWD_SEGMENTS"), int(reverse_from), _mfma_proj(d_inner, R) | _x_flags(ctx, xz, d_state),
            conv_out, x_dbl, int(seq_valid))
        ctx.reverse_from = int(reverse_from)
        ctx.seq_valid = int(seq_valid)
        ctx.delta_softplus, ctx.checkpoint_lvl = delta_softplus, checkpoint_lvl
        ctx.has_D, ctx.has_delta_bias = D is not None, delta_bias is not None
        ctx.has_out_proj = ctx.has_out_proj_bias = ctx.bidirectional = False
        ctx.reverse = bool(reverse)
        if checkpoint_lvl >= 1:
            conv_out, delta = None, None
        ctx.save_for_backward(xz, conv_w, conv_b, x_dbl, x_proj_weight, delta_proj_weight, None, conv_out, delta, A, None, None,
                              D, delta_bias, ckpt, out, None, None, None)
        return out_z
    ctx.reverse_from = int(reverse_from)
    ctx.seq_valid = int(seq_valid)
    rf = {"reverse_from": int(reverse_from)} if reverse_from else {}
    if conv_out is None:
        conv_out = causal_conv1d_cuda.causal_conv1d_fwd(xz[:, :d_inner], conv_w, conv_b, True, reverse, **rf)
    if x_dbl is None:
        x_dbl, delta = _proj_T(conv_out, x_proj_weight, delta_proj_weight)   # x_dbl: (b, R+2N, l)
    else:
        delta = torch.matmul(delta_proj_weight, x_dbl[:, :R])
    # (a one-element sequence: the size-1 axis may carry any stride)
    conv_out, x_dbl = _last_dim_contiguous(conv_out), _last_dim_contiguous(x_dbl)
    delta = _mask_padding(_last_dim_contiguous(delta), seq_valid, delta_softplus)

    if B is None:
        B = _bc_from_x_dblT(x_dbl, R, R + d_state, B_proj_bias)
    else:
        B = _last_dim_contiguous(B)
    if C is None:
        C = _bc_from_x_dblT(x_dbl, x_dbl.shape[1] - d_state, x_dbl.shape[1], C_proj_bias)
    else:
        C = _last_dim_contiguous(C)
    if D is not None:
        D = D.contiguous()

    # out_z_into: the gated output is added to that tensor by the kernel (second direction of a bidirectional node)
    out, ckpt, out_z = selective_scan_cuda.fwd(conv_out, delta, A, B, C, D, z, delta_bias, delta_softplus, reverse,
                                               **({} if out_z_into is None else {"out_z_into": out_z_into}), **rf,
                                               for_backward=_for_backward(ctx))
    saved_b = (None, None, None)
    if A_b is not None:
        assert not A_b.is_complex(), "A should not be complex!!"
        # the second scan reads the same tensors in the opposite direction (the reference flips copies of
        # all of them, SSI:499-507); its outputs come back in the original order
        # (A CONSTANT B / C is (dim, dstate): the reference's `.flip([-1])` of every scan input, SSI:506, reverses its STATES for the
        # second direction -- state n of that scan meets B[:, dstate - 1 - n].  Odd, but it is what the function returns; the drop-in
        # does the same, test_inner768_vs_reference_fixtures[inner768_bi_B0C0_f32].)
        B_b = B if ctx.is_variable_B else B.flip(-1).contiguous()
        C_b = C if ctx.is_variable_C else C.flip(-1).contiguous()
        out_b, ckpt_b, out_z_b = selective_scan_cuda.fwd(
            conv_out, delta, A_b, B_b, C_b, D, z, delta_bias, delta_softplus, not reverse, for_backward=_for_backward(ctx))
        out_z = out_z + out_z_b
        saved_b = (A_b, ckpt_b, out_b)

    ctx.delta_softplus = delta_softplus
    ctx.checkpoint_lvl = checkpoint_lvl
    ctx.has_D, ctx.has_delta_bias = D is not None, delta_bias is not None
    ctx.has_out_proj = out_proj is not None
    ctx.has_out_proj_bias = out_bias is not None
    ctx.bidirectional = A_b is not None
    ctx.reverse = bool(reverse)
    if checkpoint_lvl >= 1:  # rebuilt in backward from xz and x_dbl
        conv_out, delta = None, None
    ctx.save_for_backward(xz, conv_w, conv_b, x_dbl, x_proj_weight, delta_proj_weight, out_w,
                          conv_out, delta, A, B, C, D, delta_bias, ckpt, out, *saved_b)
    if out_proj is None:
        return out_z  # (b, d, l), d-slowest like delta
    return F.linear(out_z.transpose(1, 2), out_w, out_bias)


def _inner_backward(ctx, dout, dxz_into=None):
    """dxz_into: a dxz that already holds the gradient xz received through another node (the other direction of
    a bidirectional block); this node's dx / dz are ADDED to it by the kernels instead of by a separate pass."""
    (xz, conv_w, conv_b, x_dbl, x_proj_weight, delta_proj_weight, out_w, conv_out, delta,
     A, B, C, D, delta_bias, ckpt, out, A_b, ckpt_b, out_b) = ctx.saved_tensors
    batch, _, L = xz.shape
    R = delta_proj_weight.shape[1]
    is_complex = A.is_complex()
    d_state = A.shape[-1] * (2 if is_complex else 1)
    d_inner = conv_w.shape[0]
    x, z = xz[:, :d_inner], xz[:, d_inner:]
    dout = _last_dim_contiguous(dout)
    rf = {"reverse_from": ctx.reverse_from} if getattr(ctx, "reverse_from", 0) else {}
    if ctx.checkpoint_lvl == 1:
        conv_out = causal_conv1d_cuda.causal_conv1d_fwd(x, conv_w, conv_b, True, ctx.reverse, **rf)
        delta = _mask_padding(_last_dim_contiguous(torch.matmul(delta_proj_weight, x_dbl[:, :R])), getattr(ctx, "seq_valid", 0),
                              ctx.delta_softplus)
    if getattr(ctx, "fast", False):
        dxz, dconv_w, dconv_b, dx_proj_weight, ddelta_proj_weight, dA, dD, ddelta_bias = _inner_ext_module().inner_bwd(
            dout, xz, conv_w, conv_b, x_proj_weight, delta_proj_weight, A, D, delta_bias, conv_out, x_dbl, delta, ckpt, out,
            bool(ctx.delta_softplus), ctx.reverse, dxz_into, _vms.scan_impl_from_env(), _vms.segments_from_env("VMS_BWD_SEGMENTS"),
            getattr(ctx, "reverse_from", 0), ctx.w_dtype == torch.float32, _mfma_proj(d_inner, R))
        return dict(dxz=dxz, dconv_w=dconv_w.unsqueeze(1), dconv_b=dconv_b, dx_proj_weight=dx_proj_weight,
                    ddelta_proj_weight=ddelta_proj_weight, dout_proj_weight=None, dout_proj_bias=None, dA=dA, dA_b=None,
                    dB=None, dC=None, dD=dD, ddelta_bias=ddelta_bias, dB_proj_bias=None, dC_proj_bias=None)
    dxz = torch.empty_like(xz) if dxz_into is None else dxz_into
    acc = dxz_into is not None
    dx, dz = dxz[:, :d_inner], dxz[:, d_inner:]

    dout_2d = None
    if ctx.has_out_proj:  # dout: (b, l, e)
        dout_2d = dout.reshape(batch * L, -1).t()                       # (e, b*l)
        dy = (out_w.t() @ dout_2d).view(d_inner, batch, L).permute(1, 0, 2)  # (b, d, l) d-slowest
    else:
        dy = dout
    # one zero fill for every fp32 atomics target of this node (scan: dA, dB, dC, dD, ddelta_bias; conv: dweight, dbias)
    n_scan = selective_scan_cuda.bwd_accumulator_elems(A, B, C, D, delta_bias)
    n_conv = conv_w.numel() + (conv_b.numel() if conv_b is not None else 0)
    zeros = torch.zeros(n_scan * (2 if ctx.bidirectional else 1) + n_conv, dtype=torch.float32, device=xz.device)
    # out_z is rebuilt only where something reads it: the fused out_proj's weight gradient.  (The reference asks for it
    # in every node, SSI:247-251, and drops it in the ..NoOutProj ones: one full-tensor store per scan for nothing.)
    want_out_z = ctx.has_out_proj
    dconv_out, ddelta, dA, dB, dC, dD, ddelta_bias, dz, *out_z = selective_scan_cuda.bwd(
        conv_out, delta, A, B, C, D, z, delta_bias, dy, ckpt, out, dz, ctx.delta_softplus, want_out_z, ctx.reverse,
        zeroed=zeros[:n_scan], keep_fp32=True, accumulate_dz=acc, **rf)
    out_z = out_z[0] if out_z else None
    dA_b = None
    if ctx.bidirectional:
        B_b = B if ctx.is_variable_B else B.flip(-1).contiguous()   # (constant B / C: the second scan saw the states reversed, forward)
        C_b = C if ctx.is_variable_C else C.flip(-1).contiguous()
        dconv_b, ddelta_b, dA_b, dB_b, dC_b, dD_b, ddelta_bias_b, dz_b, *out_z_b = selective_scan_cuda.bwd(
            conv_out, delta, A_b, B_b, C_b, D, z, delta_bias, dy,
            ckpt_b, out_b, dz, ctx.delta_softplus, want_out_z, not ctx.reverse,
            zeroed=zeros[n_scan:2 * n_scan], keep_fp32=True, accumulate_dz=True)
        dconv_out = dconv_out + dconv_b
        ddelta = ddelta + ddelta_b
        dB = dB + (dB_b if ctx.is_variable_B else dB_b.flip(-1))
        dC = dC + (dC_b if ctx.is_variable_C else dC_b.flip(-1))
        if dD is not None:
            dD = dD + dD_b
        if ddelta_bias is not None:
            ddelta_bias = ddelta_bias + ddelta_bias_b
        if want_out_z:
            out_z = out_z + out_z_b[0]

    dout_proj_weight = dout_proj_bias = None
    if ctx.has_out_proj:
        dout_proj_weight = dout_2d @ out_z.transpose(1, 2).reshape(batch * L, d_inner)   # (e, d)
        if ctx.has_out_proj_bias:
            dout_proj_bias = dout.sum(dim=(0, 1))

    dx_dbl = torch.empty_like(x_dbl)                                       # (b, R+2N, l)
    dB_proj_bias = dC_proj_bias = None
    nx = x_dbl.shape[1]
    if ctx.is_variable_B:
        dB2 = dB.squeeze(1)                                                # (b, n, l)
        if ctx.has_B_proj_bias:
            dB_proj_bias = dB2.sum(dim=(0, 2))
        dx_dbl[:, R:R + d_state] = dB2
        dB = None
    if ctx.is_variable_C:
        dC2 = dC.squeeze(1)
        if ctx.has_C_proj_bias:
            dC_proj_bias = dC2.sum(dim=(0, 2))
        dx_dbl[:, nx - d_state:] = dC2
        dC = None
    ddelta_proj_weight = torch.matmul(ddelta, x_dbl[:, :R].transpose(1, 2)).sum(0, dtype=ctx.w_dtype)      # (d, R)
    dx_dbl[:, :R] = torch.matmul(delta_proj_weight.t(), ddelta)                          # (b, R, l)
    dx_proj_weight = torch.matmul(dx_dbl, conv_out.transpose(1, 2)).sum(0, dtype=ctx.w_dtype)              # (R+2N, d)
    # in place: dconv_out is this node's own buffer (the scan's du); out-of-place baddbmm copies it first
    dconv_out.baddbmm_(x_proj_weight.t().expand(batch, -1, -1), dx_dbl)                 # + W_x^T dx_dbl
    _, dconv_w, dconv_b = causal_conv1d_cuda.causal_conv1d_bwd(x, conv_w, conv_b, dconv_out, dx, True, ctx.reverse,
                                                               zeroed=zeros[zeros.numel() - n_conv:], accumulate_dx=acc, **rf)
    return dict(dxz=dxz, dconv_w=dconv_w.unsqueeze(1), dconv_b=dconv_b if conv_b is not None else None,
                dx_proj_weight=dx_proj_weight, ddelta_proj_weight=ddelta_proj_weight,
                dout_proj_weight=dout_proj_weight, dout_proj_bias=dout_proj_bias,
                dA=dA, dA_b=dA_b, dB=dB, dC=dC, dD=dD if ctx.has_D else None,
                ddelta_bias=ddelta_bias if ctx.has_delta_bias else None,
                dB_proj_bias=dB_proj_bias, dC_proj_bias=dC_proj_bias)


_DUAL_BWD = not _vms.debug.no_dual_bwd   # debug.no_dual_bwd: one backward-scan launch per direction (A/B, tests)


def _inner_backward_dual(first, second, dout):
    """Both directions' backward with their two scans as ONE call (vms_torch.cpp inner_bwd_dual -> vms_selective_scan_bwd_dual:
    one grid when the pair qualifies -- the suite's (8, 768, 3136) direction is 192 workgroups for 256 CUs).
    -> (g1, g2) as _inner_backward returns them, or (None, None) when the compiled one-call nodes are not in use."""
    if not (_DUAL_BWD and getattr(first, "fast", False) and getattr(second, "fast", False)):
        return None, None
    ext = _inner_ext_module()
    if ext is None or not hasattr(ext, "inner_bwd_dual") or first.reverse or not second.reverse:
        return None, None
    dout = _last_dim_contiguous(dout)
    packs = []
    for sub in (first, second):
        (xz, conv_w, conv_b, x_dbl, x_proj_weight, delta_proj_weight, _, conv_out, delta, A, _, _, D, delta_bias, ckpt, out,
         _, _, _) = sub.saved_tensors
        if sub.checkpoint_lvl == 1:
            R = delta_proj_weight.shape[1]
            conv_out = causal_conv1d_cuda.causal_conv1d_fwd(xz[:, :conv_w.shape[0]], conv_w, conv_b, True, sub.reverse)
            delta = _mask_padding(_last_dim_contiguous(torch.matmul(delta_proj_weight, x_dbl[:, :R])), getattr(sub, "seq_valid", 0),
                                  sub.delta_softplus)
        packs.append([conv_w, conv_b, x_proj_weight, delta_proj_weight, A, D, delta_bias, conv_out, x_dbl, delta, ckpt, out])
    r = ext.inner_bwd_dual(dout, xz, packs[0], packs[1], bool(first.delta_softplus), _vms.scan_impl_from_env(),
                           _vms.segments_from_env("VMS_BWD_SEGMENTS"), first.w_dtype == torch.float32,
                           _mfma_proj(packs[0][0].shape[0], packs[0][3].shape[1]))
    dxz = r[0]

    def as_dict(v):
        dconv_w, dconv_b, dx_proj_weight, ddelta_proj_weight, dA, dD, ddelta_bias = v
        return dict(dxz=dxz, dconv_w=dconv_w.unsqueeze(1), dconv_b=dconv_b, dx_proj_weight=dx_proj_weight,
                    ddelta_proj_weight=ddelta_proj_weight, dout_proj_weight=None, dout_proj_bias=None, dA=dA, dA_b=None,
                    dB=None, dC=None, dD=dD, ddelta_bias=ddelta_bias, dB_proj_bias=None, dC_proj_bias=None)
    return as_dict(r[1:8]), as_dict(r[8:15])


class MambaInnerFnNoOutProj(torch.autograd.Function):

    @staticmethod
    @custom_fwd
    def forward(ctx, xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight,
                A, B=None, C=None, D=None, delta_bias=None, B_proj_bias=None,
                C_proj_bias=None, delta_softplus=True, checkpoint_lvl=1, reverse=False, reverse_from=0,
                x_proj_prepared=None, delta_proj_prepared=None, seq_valid=0):
        """xz: (batch, 2*dim, seqlen) -> out_z: (batch, dim, seqlen)
        x_proj_prepared / delta_proj_prepared (extension): the two projection weights already in the autocast dtype (the block's
        one-launch parameter preparation); used instead of casting here, no gradient (it goes to the parameters).
        seq_valid (extension): 0, or the number of real positions of a zero-padded xz (_inner_forward)."""
        if x_proj_prepared is not None:
            ctx.w_dtype_override = x_proj_weight.dtype      # the PARAMETERS' dtype, not that of the prepared copies
            x_proj_weight, delta_proj_weight = x_proj_prepared, delta_proj_prepared
        return _inner_forward(ctx, xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight,
                              None, A, None, B, C, D, delta_bias, B_proj_bias, C_proj_bias,
                              delta_softplus, checkpoint_lvl, reverse, reverse_from=reverse_from, seq_valid=int(seq_valid))

    @staticmethod
    @custom_bwd
    def backward(ctx, dout):
        g = _inner_backward(ctx, dout)
        return (g["dxz"], g["dconv_w"], g["dconv_b"], g["dx_proj_weight"], g["ddelta_proj_weight"],
                g["dA"], g["dB"], g["dC"], g["dD"], g["ddelta_bias"], g["dB_proj_bias"], g["dC_proj_bias"],
                None, None, None, None, None, None, None)


class NegExpPairFn(torch.autograd.Function):
    """(A_log, A_b_log) -> (-exp(A_log), -exp(A_b_log)) in fp32 as multi-tensor kernels: 2 launches forward and 1
    backward for both directions of a block, where `-torch.exp(x.float())` twice costs 4 + 4 (each ~5 us on an
    otherwise idle GPU queue: they sit between the big kernels of every step)."""

    @staticmethod
    def forward(ctx, a_log, b_log, a_prepared=None, b_prepared=None):
        """a_prepared, b_prepared: the two results, already computed by the block's one-launch parameter preparation
        (vms_param_prep VMS_PREP_NEG_EXP); this node then only routes the gradient."""
        if a_prepared is not None:
            outs = [a_prepared, b_prepared]
        else:
            outs = torch._foreach_exp([a_log.float(), b_log.float()])
            torch._foreach_neg_(outs)
        ctx.save_for_backward(*outs)
        return tuple(o.view_as(o) for o in outs) if a_prepared is not None else tuple(outs)

    @staticmethod
    def backward(ctx, ga, gb):
        a, b = ctx.saved_tensors
        da, db = torch._foreach_mul([ga, gb], [a, b])   # d(-exp(x)) = -exp(x) dx
        return da, db, None, None


class NegExpFn(torch.autograd.Function):
    """A_log -> -exp(A_log.float()) with one launch forward (vms_param_prep VMS_PREP_NEG_EXP) and one backward, where
    `-torch.exp(x.float())` costs 2 + 2 small kernels: the DBM block's step is ~30 kernels of a few microseconds each."""

    @staticmethod
    def forward(ctx, a_log, prepared=None):
        """prepared: -exp(a_log) already computed (the block's one-launch parameter preparation); this node then only routes the gradient"""
        if prepared is not None:
            ctx.save_for_backward(prepared)
            return prepared.view_as(prepared)
        if a_log.is_cuda and a_log.dtype == torch.float32 and a_log.is_contiguous() and a_log.dim() == 2:
            out = torch.empty_like(a_log)
            _vms.param_prep([(a_log.detach(), out, _vms.PREP_NEG_EXP)])
        else:
            out = -torch.exp(a_log.float())
        ctx.save_for_backward(out)
        return out

    @staticmethod
    def backward(ctx, g):
        (a,) = ctx.saved_tensors
        return g * a, None   # d(-exp(x)) = -exp(x) dx


_DUAL_CONV = not _vms.debug.no_dual_conv   # debug.no_dual_conv: one conv1d launch per direction (A/B, tests)


def _dual_conv(xz, conv_w, conv_b, conv_w_b, conv_b_b):
    """-> (conv_out forward direction, conv_out backward direction) or (None, None) when the one-pass kernel does not apply
    (then each direction's node runs its own conv1d)."""
    if not (_DUAL_CONV and xz.is_cuda and xz.stride(-1) == 1 and conv_w.dtype == conv_w_b.dtype and conv_w.shape == conv_w_b.shape
            and (conv_b is None) == (conv_b_b is None)):
        return None, None
    d_inner = conv_w.shape[0]
    x = xz[:, :d_inner]
    w, wb = conv_w.squeeze(1), conv_w_b.squeeze(1)
    cb = conv_b.contiguous() if conv_b is not None else None
    cbb = conv_b_b.contiguous() if conv_b_b is not None else None
    ext = _vms.ext()
    if ext is not None:
        o, ob = ext.conv_fwd_dual(x, w, cb, wb, cbb, True)
        return o, ob
    o, ob = torch.empty(x.shape, dtype=x.dtype, device=x.device), torch.empty(x.shape, dtype=x.dtype, device=x.device)
    _vms.conv_fwd_dual(x, w, cb, o, wb, cbb, ob, True)
    return o, ob


_CONV_XPROJ = not _vms.debug.no_conv_xproj   # debug.no_conv_xproj: conv1d and x_proj of a bidirectional block as separate launches (A/B, tests)


def _conv_xproj_dual(xz, conv_w, conv_b, conv_w_b, conv_b_b, x_proj_w, x_proj_w_b):
    """Both directions' conv1d + SiLU AND both x_proj products from ONE pass over x (vms_conv_xproj_dual: conv1d_out is written but never
    read back) -> [conv_out, conv_out_b, x_dbl, x_dbl_b], or None when the kernel does not apply (then _dual_conv + x_proj_dual)."""
    if not (_CONV_XPROJ and _DUAL_CONV and _PROJ_KRED and xz.is_cuda):
        return None
    ext = _vms.ext()
    if ext is None or not hasattr(ext, "conv_xproj_dual"):
        return None
    d_inner = conv_w.shape[0]
    cb = conv_b.contiguous() if conv_b is not None else None
    cbb = conv_b_b.contiguous() if conv_b_b is not None else None
    r = ext.conv_xproj_dual(xz[:, :d_inner], conv_w.squeeze(1), cb, conv_w_b.squeeze(1), cbb, x_proj_w, x_proj_w_b)
    return r if len(r) == 4 else None


class _SubCtx:
    """What _inner_forward / _inner_backward need from an autograd ctx, for nodes that run them more than once."""

    def save_for_backward(self, *tensors):
        self.saved_tensors = tensors


class BiMambaInnerFnNoOutProj(torch.autograd.Function):
    """Both directions of a ViM block (two parameter sets, the second scanned the other way) as ONE node:
    out = inner(xz; set 1, left-to-right) + inner(xz; set 2, right-to-left).  The reference builds this from
    two MambaInnerFnNoOutProj nodes on xz and xz.flip (mamba_simple.py:234-258); as one node the second
    direction's dx / dz are accumulated into the first's by the kernels (no dxz_1 + dxz_2 pass in autograd)."""

    N_PER_DIR = 7  # conv1d weight, conv1d bias, x_proj weight, dt_proj weight, A, D, dt_proj bias

    @staticmethod
    @custom_fwd
    def forward(ctx, xz, delta_softplus, checkpoint_lvl, seq_valid, *params):
        """params: the 2 x 7 parameters, optionally followed by the four small projection weights (x_proj, dt_proj of both
        directions) already in the autocast dtype -- the block's one-launch parameter preparation -- which are used instead of
        casting here and receive no gradient.  seq_valid: 0, or the number of real positions of a zero-padded xz (_inner_forward)."""
        n = BiMambaInnerFnNoOutProj.N_PER_DIR
        assert len(params) in (2 * n, 2 * n + 4)
        low_given = params[2 * n:]
        params = params[:2 * n]
        ctx.n_extra = len(low_given)
        param_dtype = params[2].dtype
        if low_given:
            params = list(params)
            for i, t in zip([2, 3, n + 2, n + 3], low_given):
                params[i] = t
        elif torch.is_autocast_enabled():  # the four small projection weights of both directions: one cast kernel
            params = list(params)
            idx = [2, 3, n + 2, n + 3]
            dt = torch.get_autocast_dtype("cuda") if hasattr(torch, "get_autocast_dtype") else torch.get_autocast_gpu_dtype()
            if all(params[i].dtype != dt for i in idx):
                low = [torch.empty_like(params[i], dtype=dt) for i in idx]
                torch._foreach_copy_(low, [params[i] for i in idx])
                for i, t in zip(idx, low):
                    params[i] = t
        subs, out = [], None
        # conv1d of both directions from ONE pass over x (vms_causal_conv1d_fwd_dual): the second direction's filter runs
        # anti-causally over the same rows
        fused = _conv_xproj_dual(xz, params[0], params[1], params[n], params[n + 1], params[2], params[n + 2])
        conv_outs = fused[:2] if fused is not None else _dual_conv(xz, params[0], params[1], params[n], params[n + 1])
        # x_proj of both directions right behind it, while both outputs are (partly) in the 256 MB Infinity Cache: the second
        # direction's x_proj otherwise reads its operand from HBM after the first direction's scan (40 instead of 30 us)
        # (both as ONE launch of vms_proj_kred when the compiled binding is loaded)
        ext = _vms.ext() if conv_outs[0] is not None else None
        if fused is not None:
            x_dbls = fused[2:]
        elif ext is not None and hasattr(ext, "x_proj_dual"):
            x_dbls = ext.x_proj_dual(params[2], conv_outs[0], params[n + 2], conv_outs[1], _PROJ_KRED)
        else:
            x_dbls = [torch.matmul(params[i * n + 2], conv_outs[i]) if conv_outs[i] is not None else None for i in range(2)]
        for i in range(2):
            cw, cb, xw, dw, A, D, dbias = params[i * n:(i + 1) * n]
            sub = _SubCtx()
            sub.w_dtype_override = param_dtype   # the PARAMETERS' dtype, not that of the autocast copies made above
            sub.for_backward = any(ctx.needs_input_grad)
            # the second direction's scan adds its gated output to the first's
            out = _inner_forward(sub, xz, cw, cb, xw, dw, None, A, None, None, None, D, dbias, None, None,
                                 delta_softplus, checkpoint_lvl, reverse=(i == 1), out_z_into=out, conv_out=conv_outs[i], x_dbl=x_dbls[i],
                                 seq_valid=seq_valid)
            subs.append(sub)
        ctx.counts = [len(sub.saved_tensors) for sub in subs]
        ctx.save_for_backward(*subs[0].saved_tensors, *subs[1].saved_tensors)
        for sub in subs:
            sub.saved_tensors = None
        ctx.subs = subs
        return out

    @staticmethod
    @custom_bwd
    def backward(ctx, dout):
        saved = ctx.saved_tensors
        first, second = ctx.subs
        first.saved_tensors, second.saved_tensors = saved[:ctx.counts[0]], saved[ctx.counts[0]:]
        g1, g2 = _inner_backward_dual(first, second, dout)
        if g1 is None:
            g2 = _inner_backward(second, dout)
            g1 = _inner_backward(first, dout, dxz_into=g2["dxz"])
        first.saved_tensors = second.saved_tensors = None
        per_dir = lambda g: (g["dconv_w"], g["dconv_b"], g["dx_proj_weight"], g["ddelta_proj_weight"], g["dA"], g["dD"],
                             g["ddelta_bias"])
        return (g1["dxz"], None, None, None) + per_dir(g1) + per_dir(g2) + (None,) * ctx.n_extra


def bimamba_inner_fn_no_out_proj(xz, params, params_b, delta_softplus=True, checkpoint_lvl=1, prepared=None, seq_valid=0):
    """params / params_b: (conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight, A, D, delta_bias) of the
    left-to-right and of the right-to-left direction -> out_z_fwd + out_z_bwd, (batch, dim, seqlen).
    prepared: None, or (x_proj_weight, delta_proj_weight, x_proj_weight_b, delta_proj_weight_b) already in the autocast dtype.
    seq_valid: 0, or the number of real positions when xz[..., seq_valid:] is the caller's zero padding (_inner_forward)."""
    return BiMambaInnerFnNoOutProj.apply(xz, delta_softplus, checkpoint_lvl, int(seq_valid), *params, *params_b, *(prepared or ()))


class MambaInnerFn(torch.autograd.Function):

    @staticmethod
    @custom_fwd
    def forward(ctx, xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight,
                out_proj_weight, out_proj_bias,
                A, B=None, C=None, D=None, delta_bias=None, B_proj_bias=None,
                C_proj_bias=None, delta_softplus=True, checkpoint_lvl=1):
        """xz: (batch, 2*dim, seqlen) -> (batch, seqlen, out_features)"""
        return _inner_forward(ctx, xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight,
                              (out_proj_weight, out_proj_bias), A, None, B, C, D, delta_bias,
                              B_proj_bias, C_proj_bias, delta_softplus, checkpoint_lvl)

    @staticmethod
    @custom_bwd
    def backward(ctx, dout):
        g = _inner_backward(ctx, dout)
        return (g["dxz"], g["dconv_w"], g["dconv_b"], g["dx_proj_weight"], g["ddelta_proj_weight"],
                g["dout_proj_weight"], g["dout_proj_bias"],
                g["dA"], g["dB"], g["dC"], g["dD"], g["ddelta_bias"], g["dB_proj_bias"], g["dC_proj_bias"],
                None, None)


class BiMambaInnerFn(torch.autograd.Function):

    @staticmethod
    @custom_fwd
    def forward(ctx, xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight,
                out_proj_weight, out_proj_bias,
                A, A_b, B=None, C=None, D=None, delta_bias=None, B_proj_bias=None,
                C_proj_bias=None, delta_softplus=True, checkpoint_lvl=1):
        """Two scans (A forward in time, A_b on the flipped sequence) sharing everything else."""
        return _inner_forward(ctx, xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight,
                              (out_proj_weight, out_proj_bias), A, A_b, B, C, D, delta_bias,
                              B_proj_bias, C_proj_bias, delta_softplus, checkpoint_lvl)

    @staticmethod
    @custom_bwd
    def backward(ctx, dout):
        g = _inner_backward(ctx, dout)
        return (g["dxz"], g["dconv_w"], g["dconv_b"], g["dx_proj_weight"], g["ddelta_proj_weight"],
                g["dout_proj_weight"], g["dout_proj_bias"],
                g["dA"], g["dA_b"], g["dB"], g["dC"], g["dD"], g["ddelta_bias"],
                g["dB_proj_bias"], g["dC_proj_bias"], None, None)


def mamba_inner_fn(
    xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight,
    out_proj_weight, out_proj_bias,
    A, B=None, C=None, D=None, delta_bias=None, B_proj_bias=None,
    C_proj_bias=None, delta_softplus=True
):
    if A.is_complex():
        # complex A (tested by the reference, tests/ops/test_selective_scan.py:152-250; no suite model has one): the node as
        # the composition of the differentiable HIP ops -- conv1d, the projections, the complex scan -- instead of the
        # one-call node built for the real case
        # (not mamba_inner_ref: like the reference's, it hard-wires delta_softplus=True, SSI:669 -- MambaInnerFn honours the flag)
        x, z, delta, B, C = _inner_ref_scan_inputs(xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight, A, B, C,
                                                   B_proj_bias, C_proj_bias)
        y = selective_scan_fn(x, delta, A, B, C, D, z=z, delta_bias=delta_bias, delta_softplus=delta_softplus)
        return F.linear(y.transpose(1, 2), out_proj_weight, out_proj_bias)
    return MambaInnerFn.apply(xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight,
                              out_proj_weight, out_proj_bias,
                              A, B, C, D, delta_bias, B_proj_bias, C_proj_bias, delta_softplus)


def bimamba_inner_fn(
    xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight,
    out_proj_weight, out_proj_bias,
    A, A_b, B=None, C=None, D=None, delta_bias=None, B_proj_bias=None,
    C_proj_bias=None, delta_softplus=True
):
    return BiMambaInnerFn.apply(xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight,
                                out_proj_weight, out_proj_bias,
                                A, A_b, B, C, D, delta_bias, B_proj_bias, C_proj_bias, delta_softplus)


def mamba_inner_fn_no_out_proj(
    xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight,
    A, B=None, C=None, D=None, delta_bias=None, B_proj_bias=None,
    C_proj_bias=None, delta_softplus=True, reverse=False, checkpoint_lvl=1, reverse_from=0, prepared=None, seq_valid=0
):
    """prepared (extension): None or (x_proj_weight, delta_proj_weight) already in the autocast dtype.
    seq_valid (extension): 0, or the number of real positions when xz[..., seq_valid:] is the caller's zero padding (_inner_forward).
    reverse (extension, default off): the node runs right-to-left over xz -- the value of
    flip(node(flip(xz))) without the flipped copies the bidirectional blocks otherwise pay for.
    reverse_from (extension): batch entries >= reverse_from run right-to-left, the others left-to-right -- the DBM block's
    two halves (shared weights) as ONE node on a batch of 2 B (mamba_new.py:192-213 stacks a flipped copy instead).
    checkpoint_lvl (extension; the reference hard-wires its default 1 here): 0 keeps conv_out and delta for the
    backward instead of rebuilding them."""
    if A.is_complex():   # as mamba_inner_fn: the composition of the ops, without the output projection
        if reverse or reverse_from:
            raise RuntimeError("reverse / reverse_from are not available with a complex A")
        x, z, delta, B, C = _inner_ref_scan_inputs(xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight, A, B, C,
                                                   B_proj_bias, C_proj_bias)
        return selective_scan_fn(x, delta, A, B, C, D, z=z, delta_bias=delta_bias, delta_softplus=delta_softplus)
    return MambaInnerFnNoOutProj.apply(xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight,
                                       A, B, C, D, delta_bias, B_proj_bias, C_proj_bias, delta_softplus,
                                       checkpoint_lvl, reverse, reverse_from, *(prepared or (None, None)), int(seq_valid))


# ---- unfused references built from the public ops (dispatch to the HIP ops on GPU tensors) --------
def _inner_ref_scan_inputs(xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight, A, B, C,
                           B_proj_bias, C_proj_bias):
    batch, _, L = xz.shape
    R = delta_proj_weight.shape[1]
    is_complex = A.is_complex()
    d_state = A.shape[-1] * (2 if is_complex else 1)
    x, z = xz.chunk(2, dim=1)
    x = causal_conv1d_fn(x, conv1d_weight.squeeze(1), conv1d_bias, "silu")
    x_dbl = F.linear(x.transpose(1, 2).reshape(batch * L, -1), x_proj_weight)
    delta = _delta_from(x_dbl, delta_proj_weight, batch, L)
    if B is None:
        B = _bc_from_x_dbl(x_dbl, R, R + d_state, B_proj_bias, batch, L, is_complex).squeeze(1)
    if C is None:
        C = _bc_from_x_dbl(x_dbl, x_dbl.shape[1] - d_state, x_dbl.shape[1], C_proj_bias, batch, L,
                           is_complex).squeeze(1)
    return x, z, delta, B, C


def mamba_inner_ref(
    xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight,
    out_proj_weight, out_proj_bias,
    A, B=None, C=None, D=None, delta_bias=None, B_proj_bias=None,
    C_proj_bias=None, delta_softplus=True
):
    x, z, delta, B, C = _inner_ref_scan_inputs(xz, conv1d_weight, conv1d_bias, x_proj_weight,
                                               delta_proj_weight, A, B, C, B_proj_bias, C_proj_bias)
    y = selective_scan_fn(x, delta, A, B, C, D, z=z, delta_bias=delta_bias, delta_softplus=True)
    return F.linear(y.transpose(1, 2), out_proj_weight, out_proj_bias)


def bimamba_inner_ref(
    xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight,
    out_proj_weight, out_proj_bias,
    A, A_b, B=None, C=None, D=None, delta_bias=None, B_proj_bias=None,
    C_proj_bias=None, delta_softplus=True
):
    x, z, delta, B, C = _inner_ref_scan_inputs(xz, conv1d_weight, conv1d_bias, x_proj_weight,
                                               delta_proj_weight, A, B, C, B_proj_bias, C_proj_bias)
    y = selective_scan_fn(x, delta, A, B, C, D, z=z, delta_bias=delta_bias, delta_softplus=True)
    y_b = selective_scan_fn(_flip_l(x), _flip_l(delta), A_b, _flip_l(B), _flip_l(C), D, _flip_l(z),
                            delta_bias, delta_softplus=True)
    return F.linear((y + _flip_l(y_b)).transpose(1, 2), out_proj_weight, out_proj_bias)
