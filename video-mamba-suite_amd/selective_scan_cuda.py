"""Drop-in for the reference's pybind extension `selective_scan_cuda`
(mamba/csrc/selective_scan/selective_scan.cpp:494-497): same two entry points, same argument
order, same returned tensor lists, same checks (raised as RuntimeError, as TORCH_CHECK does).
The work is done by the gfx950 kernels behind the C ABI in include/vms_hip.h.

Differences that are visible only to someone poking at the raw extension:
  * complex A (weight_t = complex<float>, selective_scan.cpp:47, 282-287) runs on its own HIP kernels
    (csrc/selective_scan_complex.hip, vms_hip.h is_complex) through the ctypes binding; the extensions of this
    module (reverse, out_z_into, ...) beyond `reverse` are real-A only.
  * x[b, d, c, 2n] holds the state after the first 1024 elements of chunk c instead of the
    running product of exp(delta A); x[b, d, c, 2n+1] (what `last_state` slices) is unchanged.
  * 64-bit strides: no 2^32-element limit on batch_stride * batch.
"""
import torch
import torch.nn.functional as F

import vms_hip as _k

_lib = _k.lib()  # fail at import time if the HIP library is missing, like a missing .so would

_ITYPES = (torch.float32, torch.float16, torch.bfloat16)


def _check(cond, msg):
    if not cond:
        raise RuntimeError(msg)


def _common_checks(u, delta, A, B, C, D_, z_, delta_bias_):
    """selective_scan.cpp:233-305."""
    _check(u.dtype in _ITYPES, f"selective_scan: input dtype {u.dtype} not supported")
    _check(A.dtype in (torch.float32, torch.complex64), f"selective_scan: weight dtype {A.dtype} not supported")
    cl = 2 if A.is_complex() else 1   # variable B / C of a complex A: (.., 2 * seqlen) interleaved pairs (:270, 276)
    var_B, var_C = B.dim() >= 3, C.dim() >= 3
    _check(delta.dtype == u.dtype, "delta.scalar_type() == input_type")
    _check(B.dtype == (u.dtype if var_B else A.dtype), "B.scalar_type() == (!is_variable_B ? weight_type : input_type)")
    _check(C.dtype == (u.dtype if var_C else A.dtype), "C.scalar_type() == (!is_variable_C ? weight_type : input_type)")
    for name, t in (("u", u), ("delta", delta), ("A", A), ("B", B), ("C", C)):
        _check(t.is_cuda, f"{name}.is_cuda()")
    _check(u.dim() == 3, "u must be (batch, dim, seqlen)")
    _check(u.stride(-1) == 1, "u.stride(-1) == 1")
    _check(delta.stride(-1) == 1, "delta.stride(-1) == 1")
    batch, dim, seqlen = u.shape
    dstate = A.shape[1]
    n_groups = B.shape[1] if var_B else 1
    _check(dstate <= 256, "selective_scan only supports state dimension <= 256")
    _check(tuple(delta.shape) == (batch, dim, seqlen), "delta must have shape (batch, dim, seqlen)")
    _check(tuple(A.shape) == (dim, dstate), "A must have shape (dim, dstate)")
    if not var_B:
        _check(tuple(B.shape) == (dim, dstate), "B must have shape (dim, dstate)")
    else:
        _check(B.dim() == 4 and tuple(B.shape) == (batch, n_groups, dstate, seqlen * cl),
               "B must have shape (batch, n_groups, dstate, seqlen)" + (" with seqlen * 2 for a complex A" if cl == 2 else ""))
        _check(B.stride(-1) == 1, "B.stride(-1) == 1")
    if not var_C:
        _check(tuple(C.shape) == (dim, dstate), "C must have shape (dim, dstate)")
    else:
        _check(C.dim() == 4 and tuple(C.shape) == (batch, C.shape[1], dstate, seqlen * cl) and
               C.shape[1] == (n_groups if var_B else C.shape[1]),
               "C must have shape (batch, n_groups, dstate, seqlen)" + (" with seqlen * 2 for a complex A" if cl == 2 else ""))
        _check(C.stride(-1) == 1, "C.stride(-1) == 1")
    for name, t in (("D", D_), ("delta_bias", delta_bias_)):
        if t is not None:
            _check(t.dtype == torch.float32, f"{name} must be float32")
            _check(t.is_cuda, f"{name}.is_cuda()")
            _check(t.stride(-1) == 1, f"{name}.stride(-1) == 1")
            _check(tuple(t.shape) == (dim,), f"{name} must have shape (dim,)")
    if z_ is not None:
        _check(z_.dtype == u.dtype, "z.scalar_type() == input_type")
        _check(z_.is_cuda, "z.is_cuda()")
        _check(z_.stride(-1) == 1, "z.stride(-1) == 1")
        _check(tuple(z_.shape) == (batch, dim, seqlen), "z must have shape (batch, dim, seqlen)")
    return batch, dim, seqlen, dstate, var_B, var_C


def pad_bc(B, C, reverse=False, both=False):
    """Variable B, C (batch, groups, dstate, seqlen) -> (B', C', pad): the same values as views of zero-padded
    copies whose rows stay readable (zeros) for `pad` elements past their logical end -- after the last element, or
    before the first one for a right-to-left scan.  With that guarantee (vms_hip.h bc_pad) the fast kernels, which
    read B / C in 16-byte vectors, also take sequence lengths that are not a multiple of 16.  pad == 0: untouched."""
    seqlen = B.shape[-1]
    pad = (-seqlen) % 16
    if pad == 0 or B.dim() != 4 or C.dim() != 4:
        return B, C, 0
    if both:   # reverse_from: some entries are read past their end, the others before their start
        return F.pad(B, (pad, pad))[..., pad:pad + seqlen], F.pad(C, (pad, pad))[..., pad:pad + seqlen], pad
    if reverse:
        return F.pad(B, (pad, 0))[..., pad:], F.pad(C, (pad, 0))[..., pad:], pad
    return F.pad(B, (0, pad))[..., :seqlen], F.pad(C, (0, pad))[..., :seqlen], pad


def fwd(u, delta, A, B, C, D_, z_, delta_bias_, delta_softplus, reverse=False, out_z_into=None, bc_pad=None, reverse_from=0,
        for_backward=True):
    """-> [out, x, (out_z)]   (selective_scan.cpp:226-336)
    reverse (extension, default off): scan right-to-left == flip(fwd(flip(..))) without copies.
    out_z_into (extension): a (batch, dim, seqlen) tensor the gated output is ADDED to (and that is returned as
    out_z) -- the other direction's output of a bidirectional block.
    bc_pad (extension): None = pad B / C here when the length needs it (pad_bc); an int = the caller already did.
    reverse_from (extension, vms_hip.h ABI v5): batch entries >= reverse_from run right-to-left, the others left-to-right.
    for_backward (extension, ABI v7): False = x will not be handed to bwd (inference): the binding allocates the small
    checkpoint layout instead of the 8-element checkpoints the backward kernel prefers (vms_hip.h x_has_sub == 3)."""
    if A.is_complex():
        return _fwd_complex(u, delta, A, B, C, D_, z_, delta_bias_, delta_softplus, reverse, out_z_into, reverse_from)
    ext = _k.ext()
    impl = _k.scan_impl_from_env()
    if ext is not None:   # compiled binding: same checks / allocations / launch in C++
        if bc_pad is None:
            B, C, bc_pad = pad_bc(B, C, reverse, reverse_from > 0)
        return ext.scan_fwd(u, delta, A, B, C, D_, z_, delta_bias_, bool(delta_softplus), bool(reverse), out_z_into,
                            bc_pad, impl, _k.segments_from_env("VMS_FWD_SEGMENTS"), int(reverse_from),
                            _k.x_mode_for(u, A.shape[1], for_backward))
    batch, dim, seqlen, dstate, _, _ = _common_checks(u, delta, A, B, C, D_, z_, delta_bias_)
    n_chunks = (seqlen + 2047) // 2048
    out = torch.empty_like(delta)  # inherits delta's (d-slowest) layout, selective_scan.cpp:310-311
    out_z = torch.empty_like(z_) if z_ is not None else None
    if out_z_into is not None:
        _check(z_ is not None, "out_z_into needs z")
        _check(out_z_into.dtype == u.dtype and out_z_into.is_cuda and out_z_into.stride(-1) == 1 and
               tuple(out_z_into.shape) == (batch, dim, seqlen),
               "out_z_into must be (batch, dim, seqlen), input dtype, unit last stride")
        out_z = out_z_into
    # x: the reference-shaped (batch, dim, n_chunks, 2*dstate) tensor, allocated by the binding as a view
    # of a larger buffer that also carries the finer checkpoints the backward kernels start from
    if bc_pad is None:
        B, C, bc_pad = pad_bc(B, C, reverse, reverse_from > 0)
    x = _k.scan_fwd(u, delta, A, B, C, D_, z_, delta_bias_, out, out_z, None, delta_softplus, reverse,
                    out_z_into is not None, bc_pad, reverse_from, for_backward)
    return [out, x] + ([out_z] if z_ is not None else [])


def _fwd_complex(u, delta, A, B, C, D_, z_, delta_bias_, delta_softplus, reverse, out_z_into, reverse_from):
    """complex A (selective_scan.cpp:282-287): x is complex64 (batch, dim, n_chunks, 2 * dstate) as the reference allocates it
    (:313), here the view of a (.., 6 * dstate) buffer whose tail holds the state after every 512 elements for bwd."""
    batch, dim, seqlen, dstate, _, _ = _common_checks(u, delta, A, B, C, D_, z_, delta_bias_)
    _check(0 <= reverse_from <= batch and not (reverse and reverse_from), "reverse_from must be in [0, batch] with reverse off")
    n_chunks = (seqlen + 2047) // 2048
    out = torch.empty_like(delta)
    out_z = torch.empty_like(z_) if z_ is not None else None
    if out_z_into is not None:
        _check(z_ is not None, "out_z_into needs z")
        _check(out_z_into.dtype == u.dtype and out_z_into.is_cuda and out_z_into.stride(-1) == 1 and
               tuple(out_z_into.shape) == (batch, dim, seqlen),
               "out_z_into must be (batch, dim, seqlen), input dtype, unit last stride")
        out_z = out_z_into
    x = torch.empty(batch, dim, n_chunks, 6 * dstate, device=u.device, dtype=torch.complex64)[..., :2 * dstate]
    _k.scan_fwd(u, delta, A, B, C, D_, z_, delta_bias_, out, out_z, x, delta_softplus, reverse, out_z_into is not None, 0, int(reverse_from))
    return [out, x] + ([out_z] if z_ is not None else [])


def bwd_accumulator_elems(A, B, C, D_, delta_bias_):
    """fp32 elements of zeroed scratch bwd(..., zeroed=) carves its atomics targets from."""
    return A.numel() + B.numel() + C.numel() + (D_.numel() if D_ is not None else 0) + (
        delta_bias_.numel() if delta_bias_ is not None else 0)


def _carve(flat, offset, like):
    n = like.numel()
    return flat[offset:offset + n].view(like.shape), offset + n


def _bwd_prepare(u, delta, A, B, C, D_, z_, delta_bias_, dout, x_, out_, dz_, delta_softplus, recompute_out_z, reverse,
                 zeroed, keep_fp32, accumulate_dz, bc_pad, reverse_from, no_dz=False):
    """The checks and allocations of selective_scan.cpp:338-492 -> (arguments of vms_hip.scan_bwd, results()).
    no_dz: z is given but this launch produces no dz (the second direction of bwd_dual)."""
    batch, dim, seqlen, dstate, var_B, var_C = _common_checks(u, delta, A, B, C, D_, z_, delta_bias_)
    _check(dout.dtype == u.dtype, "dout.scalar_type() == input_type")
    _check(dout.is_cuda, "dout.is_cuda()")
    _check(dout.stride(-1) == 1, "dout.stride(-1) == 1")
    _check(tuple(dout.shape) == (batch, dim, seqlen), "dout must have shape (batch, dim, seqlen)")
    out = dz = out_z = None
    if z_ is not None:
        _check(out_ is not None, "out_.has_value()")
        out = out_
        _check(out.dtype == u.dtype and out.is_cuda and out.stride(-1) == 1 and
               tuple(out.shape) == (batch, dim, seqlen), "out must be (batch, dim, seqlen), input dtype, unit last stride")
        _check(not accumulate_dz or dz_ is not None, "accumulate_dz needs the dz tensor to add to")
        if dz_ is not None:
            dz = dz_
            _check(dz.dtype == u.dtype and dz.is_cuda and dz.stride(-1) == 1 and
                   tuple(dz.shape) == (batch, dim, seqlen), "dz must be (batch, dim, seqlen), input dtype, unit last stride")
        elif not no_dz:
            dz = torch.empty_like(z_)
        if recompute_out_z:
            out_z = torch.empty_like(out)
    n_chunks = (seqlen + 2047) // 2048
    if n_chunks > 1:
        _check(x_ is not None, "x_.has_value()")
    if x_ is not None:
        _check(x_.dtype == A.dtype and x_.is_cuda and tuple(x_.shape) == (batch, dim, n_chunks, 2 * dstate)
               and x_.stride(3) == 1 and x_.stride(1) == n_chunks * x_.stride(2)
               and x_.stride(0) == dim * x_.stride(1),
               "x must be the (batch, dim, n_chunks, 2*dstate) checkpoint tensor returned by fwd")
    else:
        _check(seqlen <= (512 if A.is_complex() else 1024), "x (the forward's checkpoints) is required for this seqlen")
    du = torch.empty_like(u)
    ddelta = torch.empty_like(delta)
    if bc_pad is None:
        Bk, Ck, bc_pad = pad_bc(B, C, reverse, reverse_from > 0)   # B / C themselves keep their shapes for the gradients below
    else:
        Bk, Ck = B, C
    if zeroed is not None and not A.is_complex():
        _check(zeroed.dtype == torch.float32 and zeroed.is_cuda and zeroed.dim() == 1 and zeroed.is_contiguous()
               and zeroed.numel() >= bwd_accumulator_elems(A, B, C, D_, delta_bias_),
               "zeroed must be a flat float32 tensor of at least bwd_accumulator_elems() elements")
        dA, o = _carve(zeroed, 0, A)
        dB, o = _carve(zeroed, o, B)
        dC, o = _carve(zeroed, o, C)
        dD = ddelta_bias = None
        if D_ is not None:
            dD, o = _carve(zeroed, o, D_)
        if delta_bias_ is not None:
            ddelta_bias, o = _carve(zeroed, o, delta_bias_)
    else:
        dA = torch.zeros_like(A)
        # variable B / C: fp32 accumulators cast on return (:461-462, 488); constant ones in the weight type
        dB = torch.zeros_like(B, dtype=torch.float32 if B.dim() >= 3 else A.dtype)
        dC = torch.zeros_like(C, dtype=torch.float32 if C.dim() >= 3 else A.dtype)
        dD = torch.zeros_like(D_) if D_ is not None else None
        ddelta_bias = torch.zeros_like(delta_bias_) if delta_bias_ is not None else None
    kargs = (u, delta, A, Bk, Ck, D_, z_, delta_bias_, dout, x_, out, out_z, du, ddelta, dA, dB, dC, dD,
             ddelta_bias, dz, delta_softplus, reverse, bool(accumulate_dz), bc_pad, reverse_from)

    def results():
        rB, rC = (dB, dC) if keep_fp32 else (dB.to(B.dtype), dC.to(C.dtype))
        result = [du, ddelta, dA, rB, rC, dD, ddelta_bias]
        if z_ is not None and not no_dz:
            result.append(dz)
        if recompute_out_z:
            result.append(out_z)
        return result
    return kargs, results


def bwd(u, delta, A, B, C, D_, z_, delta_bias_, dout, x_, out_, dz_, delta_softplus, recompute_out_z, reverse=False,
        zeroed=None, keep_fp32=False, accumulate_dz=False, bc_pad=None, reverse_from=0):
    """-> [du, ddelta, dA, dB, dC, dD, ddelta_bias, (dz), (out_z)]   (selective_scan.cpp:338-492)
    zeroed (extension): a flat, ZERO fp32 tensor of >= bwd_accumulator_elems(..) elements to hold dA, dB, dC, dD and
    ddelta_bias (one fill by the caller instead of five here); keep_fp32: return dB / dC as accumulated (fp32);
    accumulate_dz: dz_ += instead of dz_ = (dz_ must be given)."""
    ext = _k.ext() if not A.is_complex() else None   # complex A: the checks / allocations below, launch through ctypes
    impl = _k.scan_impl_from_env()
    if A.is_complex():
        bc_pad = 0
    if ext is not None:
        if bc_pad is None:
            Bk, Ck, bc_pad = pad_bc(B, C, reverse, reverse_from > 0)
        else:
            Bk, Ck = B, C
        return ext.scan_bwd(u, delta, A, Bk, Ck, D_, z_, delta_bias_, dout, x_, out_, dz_, bool(delta_softplus),
                            bool(recompute_out_z), bool(reverse), zeroed, bool(keep_fp32), bool(accumulate_dz), bc_pad, impl,
                            _k.segments_from_env("VMS_BWD_SEGMENTS"), B, C, int(reverse_from))
    kargs, results = _bwd_prepare(u, delta, A, B, C, D_, z_, delta_bias_, dout, x_, out_, dz_, delta_softplus, recompute_out_z,
                                  reverse, zeroed, keep_fp32, accumulate_dz, bc_pad, reverse_from)
    _k.scan_bwd(*kargs)
    return results()


def bwd_dual(dir_a, dir_b, z, dout, dz_, delta_softplus, zeroed_a=None, zeroed_b=None, keep_fp32=False, accumulate_dz=False):
    """The backward scans of BOTH directions of a bidirectional block in one call (vms_hip.h vms_selective_scan_bwd_dual; the
    reference launches selective_scan_cuda.bwd twice, SSI:541-561).  dir_a / dir_b = (u, delta, A, B, C, D_, delta_bias_, x_,
    out_) of the left-to-right / right-to-left direction; z, dout (and dz_) are shared.
    -> ([du, ddelta, dA, dB, dC, dD, ddelta_bias, dz], [du, ddelta, dA, dB, dC, dD, ddelta_bias]): a's list carries the gradient
    z receives through both directions.  One grid when the pair qualifies (vms_scan_bwd_dual_fused), else two launches."""
    ua, da, Aa, Ba, Ca, Da, ba, xa, oa = dir_a
    ub, db, Ab, Bb, Cb, Db, bb, xb, ob = dir_b
    _check(z is not None, "bwd_dual: z is required (without a gate the two directions share nothing: call bwd twice)")
    ext = _k.ext()
    impl = _k.scan_impl_from_env()
    Bka, Cka, pad_a = pad_bc(Ba, Ca, False)
    Bkb, Ckb, pad_b = pad_bc(Bb, Cb, True)
    if ext is not None and hasattr(ext, "scan_bwd_dual"):
        ra, rb = ext.scan_bwd_dual(ua, da, Aa, Bka, Cka, Da, ba, xa, oa, zeroed_a, Ba, Ca,
                                   ub, db, Ab, Bkb, Ckb, Db, bb, xb, ob, zeroed_b, Bb, Cb,
                                   z, dout, dz_, bool(delta_softplus), bool(keep_fp32), bool(accumulate_dz), pad_a, pad_b, impl,
                                   _k.segments_from_env("VMS_BWD_SEGMENTS"))
        return ra, rb
    ka, res_a = _bwd_prepare(ua, da, Aa, Bka, Cka, Da, z, ba, dout, xa, oa, dz_, delta_softplus, False, False, zeroed_a,
                             keep_fp32, accumulate_dz, pad_a, 0)
    kb, res_b = _bwd_prepare(ub, db, Ab, Bkb, Ckb, Db, z, bb, dout, xb, ob, None, delta_softplus, False, True, zeroed_b,
                             keep_fp32, False, pad_b, 0, no_dz=True)
    _k.scan_bwd_dual(ka, kb)
    ra, rb = res_a(), res_b()
    if not keep_fp32:   # _bwd_prepare saw the padded B / C: the gradients take the callers' dtypes (already their shapes)
        ra[3], ra[4], rb[3], rb[4] = ra[3].to(Ba.dtype), ra[4].to(Ca.dtype), rb[3].to(Bb.dtype), rb[4].to(Cb.dtype)
    return ra, rb
