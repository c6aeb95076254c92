"""The Mamba block's two large projections as autograd nodes that keep the kernels' layouts.

The scan / conv kernels want (batch, channels, seqlen) with a unit seqlen stride.  The reference gets there
with `rearrange(in_proj.weight @ rearrange(hidden, "b l d -> d (b l)"), "d (b l) -> b d l")`
(mamba_ssm/modules/mamba_simple.py:144-149) and leaves both backward GEMMs to autograd, which runs the
weight gradient as ONE (channels x d_model) GEMM with K = batch * seqlen -- 32 output tiles for 256 CUs.
Here the weight gradients are split along K into a batched GEMM plus a sum (2.4x faster at
(8, 8192, 1024) on MI355X: tools/gemm_wgrad.py, tools/attic/gemm_outproj.py), and out_proj consumes and
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


_SUM_SLICES = True   # False (tests / A/B: monkeypatch.setattr): torch's reduction instead of vms_sum_slices
_mm_out_dtype_ok = {}   # (device index, dtype) -> does torch.mm(a, b, out_dtype=torch.float32) work for these operands on this build / device?


def _mm_out_dtype(like):
    """Probed ONCE, explicitly, on tiny operands of the caller's dtype and device (outside any stream capture): an unrelated first
    failure of a real call -- an out-of-memory error is a RuntimeError too -- must not flip the kernel path of every later weight
    gradient, and the batched branch of _in_proj_param_grads must not depend on which node ran first."""
    key = (like.device.index, like.dtype)   # per device and operand dtype (one global answer served fp16 after a bf16 probe)
    if key not in _mm_out_dtype_ok:
        if torch.cuda.is_current_stream_capturing():
            # decided at the first call outside a capture; inside one the cast form is always valid.  GraphedStep's warm-up steps run
            # the same nodes eagerly first, so a captured step and the eager steps around it take the same path.
            return False
        try:
            t = torch.ones(16, 16, device=like.device, dtype=like.dtype)
            torch.mm(t, t, out_dtype=torch.float32)
            torch.bmm(t[None], t[None], out_dtype=torch.float32)
            _mm_out_dtype_ok[key] = True
        except torch.cuda.OutOfMemoryError:
            raise
        except (RuntimeError, TypeError):
            _mm_out_dtype_ok[key] = False
    return _mm_out_dtype_ok[key]


def _mm_wgrad(a, b, w_dtype):
    """a (m, k) @ b (k, n) -> (m, n) in the parameter's dtype: an unsliced weight gradient.  With 16-bit operands and an fp32
    parameter the GEMM writes fp32 itself (out_dtype: no cast kernel behind it, no rounding to 16 bits in between) where the
    library offers that; otherwise the product in the operands' dtype, then the cast."""
    if a.is_cuda and w_dtype == torch.float32 and a.dtype in (torch.bfloat16, torch.float16) and _mm_out_dtype(a):
        return torch.mm(a, b, out_dtype=torch.float32)
    return torch.mm(a, b).to(w_dtype)


def _sum_slices(t, w_dtype):
    """sum over the K slices of a batched weight-gradient GEMM, in the parameter's dtype (vms_sum_slices on the GPU: one streaming
    launch instead of torch's multi-block reduction + its semaphore fill)"""
    if t.is_cuda and t.shape[0] > 1 and _SUM_SLICES:
        import vms_hip
        return vms_hip.sum_slices(t, w_dtype)
    return t.sum(0, dtype=w_dtype)


def _k_splits(k_total, rows_out=2048, cols_out=1024, most=32):
    """Number of K slices for a weight-gradient GEMM (K = batch * seqlen, output rows_out x cols_out): the library tiles the
    small output 256 x 256, so slices x tiles should come to about one workgroup per CU -- 8 slices for in_proj at d_model 1024
    (32 tiles), 16 for 1024 x 1024 or 1536 x 768, 14 for the latter at 8 x 3136 rows (tools/gemm_wgrad_split.py,
    profiles/r04_wgrad_splits.txt: (25088, 1536 x 768) 101 us with the 2 slices of the earlier rule, 73 with 14; (65536, 1536 x
    768) 197 -> 153 us) -- with at least 1024 rows per slice, and the count must divide K."""
    if k_total <= 8192:      # a GEMM this small takes the same ~27 us sliced or not ((4608, 2048 x 512): 30.5 / 27.3 / 27.6 us for
        return 1             # 1 / 4 / 8 slices) and the sum of the slices is one more kernel in a launch-bound step (the DBM block)
    tiles = -(-rows_out // 256) * -(-cols_out // 256)
    want = max(1, min(most, round(256 / tiles), k_total // 1024))
    best = 1
    for s in range(1, most + 1):
        if k_total % s == 0 and abs(s - want) < abs(best - want):
            best = s
    return best


def _interleave_halves(t, dim):
    """channels ordered [half][c] along `dim` -> [c][half] (the order in which the two halves of the DBM block's channels
    become the two halves of a stacked BATCH, see InProjFn): a view."""
    n = t.shape[dim]
    shp = list(t.shape)
    v = t.reshape(shp[:dim] + [2, n // 2] + shp[dim + 1:]).transpose(dim, dim + 1)
    return v


class InProjFn(torch.autograd.Function):
    """hidden (B, L, d_model), weight (C, d_model), bias (C,) | None -> xz (B, C, L), C-slowest in memory.

    stack_halves (the DBM block, whose projection emits the (x, z) of BOTH directions, C = 2 * 2 d_inner): the result is
    the (2 B, C / 2, L) tensor whose entries [0, B) are the first half of the channels and [B, 2 B) the second -- what
    `torch.cat(xz.chunk(2, dim=1), dim=0)` would copy together, obtained for free by emitting the GEMM's output rows in
    the order [c][half]: the (C, B L) product, read as [c][half][b][l], IS (C / 2, 2 B, L).  The row order is a
    permutation of the weight's rows, folded into the transposed weight copy this node makes anyway."""

    @staticmethod
    @custom_fwd
    def forward(ctx, hidden, weight, bias, stack_halves=False, wt_prepared=None, param_grads=True):
        """wt_prepared: weight^T already in the compute dtype, (d_model, channels) -- made by the block's one-launch parameter
        preparation (modules/_core.py, vms_param_prep); not differentiated (the gradient goes to `weight`).
        param_grads=False: this node returns the input gradient only; InProjParamGradFn, placed behind it, owns dweight / dbias."""
        ctx.param_grads = param_grads
        batch, seqlen, d_model = hidden.shape
        channels = weight.shape[0]
        x2 = hidden.reshape(batch * seqlen, d_model)
        if stack_halves:
            assert channels % 2 == 0
        dt = (_autocast_dtype() or weight.dtype) if hidden.is_cuda else weight.dtype
        # the K-contiguous copy of the weight pays when a copy is made anyway -- autocast's cast, or the block's prepared one;
        # an fp32 (or already low-precision) weight outside autocast keeps the parameter itself: no copy kernel, no weight-sized
        # saved tensor, autograd's version check on the parameter intact
        transposed = hidden.is_cuda and (wt_prepared is not None or dt != weight.dtype)
        if transposed:
            if wt_prepared is not None:
                # (with stack_halves: the prepared copy already has its columns in the order [c][half])
                assert wt_prepared.dtype == dt and tuple(wt_prepared.shape) == (d_model, channels)
                wt = wt_prepared
            else:
                # W^T as its own (d_model, channels) matrix in the compute dtype: cast and transpose in one copy kernel
                wt = torch.empty(d_model, channels, dtype=dt, device=weight.device)
            if wt_prepared is not None:
                pass
            elif stack_halves:   # column c * 2 + half of wt <- row half * (C / 2) + c of the weight
                wt.view(d_model, channels // 2, 2).copy_(weight.view(2, channels // 2, d_model).permute(2, 1, 0))
            else:
                wt.copy_(weight.t())
            if x2.dtype != dt:
                x2 = x2.to(dt)
            prod = wt.t() @ x2.t()                               # (channels, B L)
        else:  # the plain formulation (also the CPU one, for the tests with checker-backed extensions)
            wt = weight
            w = _interleave_halves(weight, 0).reshape(channels, d_model) if stack_halves else weight
            if x2.dtype != w.dtype:
                x2 = x2.to(w.dtype)
            prod = w @ x2.t()
        if bias is not None:
            b = _interleave_halves(bias, 0).reshape(channels) if stack_halves else bias
            prod = prod + b.to(dtype=prod.dtype)[:, None]
        if stack_halves:
            xz = prod.view(channels // 2, 2 * batch, seqlen).permute(1, 0, 2)
        else:
            xz = prod.view(channels, batch, seqlen).permute(1, 0, 2)
        ctx.save_for_backward(hidden, wt)
        ctx.transposed = transposed
        ctx.w_dtype = weight.dtype   # the K-split partial sums are added in the PARAMETER's dtype: no cast kernel in autograd
        ctx.stack_halves = stack_halves
        ctx.has_bias = bias is not None
        return xz

    @staticmethod
    @custom_bwd
    def backward(ctx, dxz):
        hidden, wt = ctx.saved_tensors
        batch, seqlen, d_model = hidden.shape
        rows = batch * seqlen
        channels = wt.shape[1] if ctx.transposed else wt.shape[0]
        # rows of g2 in the order the forward emitted them ([c][half] when stacked); a view when dxz has xz's layout
        g2 = dxz.permute(1, 0, 2).reshape(channels, rows)
        x2 = hidden.reshape(rows, d_model)
        unstack = (lambda t: t.reshape((channels // 2, 2) + t.shape[1:]).transpose(0, 1).reshape(t.shape)) if ctx.stack_halves \
            else (lambda t: t)
        dhidden = dweight = dbias = None
        if ctx.needs_input_grad[0]:
            if ctx.transposed:
                dhidden = (g2.to(wt.dtype).t() @ wt.t()).view(batch, seqlen, d_model).to(hidden.dtype)
            else:
                w = _interleave_halves(wt, 0).reshape(channels, d_model) if ctx.stack_halves else wt
                dhidden = (g2.to(w.dtype).t() @ w).view(batch, seqlen, d_model).to(hidden.dtype)
        if ctx.param_grads:
            dweight, dbias = _in_proj_param_grads(g2, x2, channels, d_model, ctx.w_dtype, ctx.stack_halves,
                                                  ctx.needs_input_grad[1], ctx.has_bias and ctx.needs_input_grad[2])
        return dhidden, dweight, dbias, None, None, None


def _in_proj_param_grads(g2, x2, channels, d_model, w_dtype, stack_halves, want_w, want_b):
    """g2 (channels, B L) = the gradient of xz in the order the forward emitted its rows, x2 (B L, d_model) -> dweight, dbias"""
    rows = x2.shape[0]
    unstack = (lambda t: t.reshape((channels // 2, 2) + t.shape[1:]).transpose(0, 1).reshape(t.shape)) if stack_halves else (lambda t: t)
    dweight = dbias = None
    if want_w:
        s = _k_splits(rows, channels, d_model)
        if x2.dtype != g2.dtype:
            x2 = x2.to(g2.dtype)
        if s == 1 and stack_halves and g2.is_cuda and w_dtype == torch.float32 and g2.dtype in (torch.bfloat16, torch.float16) and _mm_out_dtype(g2):
            # the rows of g2 are ordered [c][half]: as the strided (half, c, rows) view the two halves are two batched GEMMs whose
            # (2, C / 2, d_model) result IS the parameter's row order -- no un-stacking copy behind the GEMM
            g3 = g2.view(channels // 2, 2, rows).permute(1, 0, 2)
            dweight = torch.bmm(g3, x2.unsqueeze(0).expand(2, -1, -1), out_dtype=torch.float32).view(channels, d_model)
        elif s == 1:
            dweight = unstack(_mm_wgrad(g2, x2, w_dtype))
        else:
            dweight = unstack(_sum_slices(torch.bmm(g2.view(channels, s, rows // s).permute(1, 0, 2), x2.view(s, rows // s, d_model)), w_dtype))
    if want_b:
        dbias = unstack(g2.sum(dim=1))
    return dweight, dbias


class InProjParamGradFn(torch.autograd.Function):
    """Identity on xz whose backward produces in_proj's weight (and bias) gradient and passes dxz on to InProjFn, which then
    computes the input gradient.  Why two nodes: in_proj's weight gradient is the LAST parameter gradient of a block's backward;
    as one node with the input gradient, DistributedDataParallel's hook for it -- and the all-reduce of the last bucket -- fires
    only after BOTH GEMMs.  The engine runs this node first (created later), so the bucket's all-reduce (8 MB at d_model 1024)
    overlaps the 0.22 ms input-gradient GEMM instead of trailing the step.  Same kernels, same values on one GPU."""

    @staticmethod
    @custom_fwd
    def forward(ctx, xz, hidden, weight, bias, stack_halves):
        ctx.save_for_backward(hidden)
        ctx.channels, ctx.w_dtype, ctx.stack_halves, ctx.has_bias = weight.shape[0], weight.dtype, stack_halves, bias is not None
        return xz.view_as(xz)

    @staticmethod
    @custom_bwd
    def backward(ctx, dxz):
        (hidden,) = ctx.saved_tensors
        batch, seqlen, d_model = hidden.shape
        g2 = dxz.permute(1, 0, 2).reshape(ctx.channels, batch * seqlen)
        dweight, dbias = _in_proj_param_grads(g2, hidden.reshape(batch * seqlen, d_model), ctx.channels, d_model, ctx.w_dtype,
                                              ctx.stack_halves, ctx.needs_input_grad[2], ctx.has_bias and ctx.needs_input_grad[3])
        return dxz, None, dweight, dbias, None


class OutProjFn(torch.autograd.Function):
    """y (B, C, L) with unit seqlen stride, weight (d_model, C), bias | None -> (B, L, d_model).

    stacked_halves (the DBM block): y is the (2 B, C / 2, L) output of the stacked node (entries [0, B): the channels
    [0, C / 2) of the projection's input, entries [B, 2 B): the channels [C / 2, C)), channel-slowest in memory as the scan
    leaves it -- i.e. already the (C, B L) matrix with rows [c][half]; the weight's columns are permuted to match (one
    small copy, with the autocast cast) instead of torch.cat copying the activations."""

    @staticmethod
    @custom_fwd
    def forward(ctx, y, weight, bias, stacked_halves=False, w_prepared=None):
        """w_prepared: weight already in the compute dtype (the block's one-launch parameter preparation); used by the forward
        and the input-gradient GEMM instead of autocast's two casts of `weight`, not differentiated."""
        ctx.has_bias = bias is not None
        ctx.stacked_halves = stacked_halves
        ctx.w_dtype = weight.dtype
        if not stacked_halves:
            w = weight if w_prepared is None else w_prepared
            ctx.save_for_backward(y, w)
            return F.linear(y.transpose(1, 2), w, bias)
        b2, half_c, seqlen = y.shape
        batch, d_model = b2 // 2, weight.shape[0]
        dt = (_autocast_dtype() or weight.dtype) if y.is_cuda else weight.dtype
        if w_prepared is not None:     # the block's one-launch preparation made the permuted copy
            assert w_prepared.dtype == dt and tuple(w_prepared.shape) == (d_model, 2 * half_c)
            wp = w_prepared
        else:
            wp = torch.empty(d_model, 2 * half_c, dtype=dt, device=weight.device)     # column c * 2 + half <- column half * C/2 + c
            wp.view(d_model, half_c, 2).copy_(weight.view(d_model, 2, half_c).transpose(1, 2))
        y2 = y.permute(1, 0, 2).reshape(2 * half_c, batch * seqlen)                 # a view in the scan's layout
        if y2.dtype != dt:
            y2 = y2.to(dt)
        out = (y2.t() @ wp.t()).view(batch, seqlen, d_model)
        if bias is not None:
            out = out + bias.to(out.dtype)
        ctx.save_for_backward(y2, wp)
        ctx.dims = (batch, half_c, seqlen)
        return out

    @staticmethod
    @custom_bwd
    def backward(ctx, dout):
        if not ctx.stacked_halves:
            y, weight = ctx.saved_tensors
            dy = dweight = dbias = None
            if ctx.needs_input_grad[0]:
                if dout.is_contiguous() and dout.shape[2] >= 512:
                    # ONE GEMM over the flattened rows: its (C, B L) result IS the channel-slowest (B, C, L) tensor the scan
                    # backward reads (any batch / channel strides); the library runs it faster than B batched ones at d_model >=
                    # 512 ((8, 3136, 768) 55.9 -> 43.9 us, (8, 8192, 1024) 144 -> 136; d_model 384: 45 vs 49, kept batched;
                    # tools/gemm_outproj_dgrad.py)
                    b_, l_, dm_ = dout.shape
                    dy = torch.matmul(weight.t(), dout.reshape(b_ * l_, dm_).t()).view(weight.shape[1], b_, l_).permute(1, 0, 2)
                else:
                    dy = torch.matmul(weight.t(), dout.transpose(1, 2))  # (B, C, L): the layout the scan backward reads
            if ctx.needs_input_grad[1]:
                # one K slice per batch entry, summed in the parameter's dtype; produced as (d_model, C) = the parameter's own
                # layout (the transposed product's .t() view cost autograd a 10 us copy when it stored the gradient, and this
                # operand order is the library's faster one here: tools/attic/gemm_outproj_wgrad.py, 154 -> 138 us with the sum)
                b_, c_, l_ = y.shape
                if y.stride(2) == 1 and y.stride(0) == l_ and y.stride(1) == b_ * l_ and dout.is_contiguous():
                    # y in the scan's channel-slowest layout IS the (C, B L) matrix: K slices of the flattened rows, as many as
                    # fill the chip (batch 1 at 65,536 positions was ONE 768 x 768 GEMM with K = 65,536: 196 -> 94 us)
                    rows = b_ * l_
                    ks = _k_splits(rows, dout.shape[2], c_)
                    dweight = _sum_slices(torch.bmm(dout.reshape(ks, rows // ks, -1).transpose(1, 2),
                                                    y.permute(1, 0, 2).reshape(c_, ks, rows // ks).permute(1, 2, 0)), ctx.w_dtype)
                else:
                    dweight = _sum_slices(torch.bmm(dout.transpose(1, 2), y.transpose(1, 2)), ctx.w_dtype)
            if ctx.has_bias and ctx.needs_input_grad[2]:
                dbias = dout.sum(dim=(0, 1))
            return dy, dweight, dbias, None, None
        y2, wp = ctx.saved_tensors
        batch, half_c, seqlen = ctx.dims
        d_model = wp.shape[0]
        dout = dout.to(wp.dtype)
        dy = dweight = dbias = None
        if ctx.needs_input_grad[0]:
            dy2 = torch.matmul(wp.t(), dout.reshape(batch * seqlen, d_model).t())        # (C, B L), rows [c][half]
            dy = dy2.view(half_c, 2 * batch, seqlen).permute(1, 0, 2)                      # (2 B, C / 2, L), the scan's layout
        if ctx.needs_input_grad[1]:
            # (d_model, C) permuted: ONE GEMM over the flattened rows (y2 is the (C, B L) matrix), fp32 out of the GEMM where offered
            dwp = _mm_wgrad(dout.reshape(batch * seqlen, d_model).t(), y2.t(), ctx.w_dtype)
            dweight = dwp.view(d_model, half_c, 2).transpose(1, 2).reshape(d_model, 2 * half_c)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            dbias = dout.sum(dim=(0, 1))
        return dy, dweight, dbias, None, None


def in_proj_fn(hidden, weight, bias=None, stack_halves=False, wt_prepared=None):
    if not hidden.requires_grad or not (weight.requires_grad or (bias is not None and bias.requires_grad)) or not torch.is_grad_enabled():
        return InProjFn.apply(hidden, weight, bias, stack_halves, wt_prepared)
    # parameter gradients from their own node, ahead of the input gradient (see InProjParamGradFn)
    # (detached parameters: an edge from InProjFn to their AccumulateGrad nodes, even one that carries None, would make those wait
    # for InProjFn's backward)
    xz = InProjFn.apply(hidden, weight.detach(), bias.detach() if bias is not None else None, stack_halves, wt_prepared, False)
    return InProjParamGradFn.apply(xz, hidden, weight, bias, stack_halves)


def out_proj_fn(y, weight, bias=None, stacked_halves=False, w_prepared=None):
    return OutProjFn.apply(y, weight, bias, stacked_halves, w_prepared)
