"""CPU stand-ins for the two extension modules (test infrastructure): same argument lists / returns as
selective_scan_cuda / causal_conv1d_cuda, computed with the C oracle.  Used to test the host logic of the
fused autograd nodes and the modules without a GPU (tests/test_cpu_surface.py, tests/ddp_worker.py)."""
import types

import numpy as np
import torch


def make_fakes(oracle):
    def np_(t, rev=False):
        """rev: the extension's right-to-left mode is, by definition, the causal op on flipped copies"""
        if t is None:
            return None
        a = t.detach().float().cpu().numpy()
        return np.ascontiguousarray(a[..., ::-1]) if (rev and a.ndim >= 3) else a

    def npc(t):   # complex tensors keep their dtype (the oracle's complex entry points tell constant from variable B / C by it)
        return t.detach().cpu().numpy() if t.is_complex() else np_(t)

    def un(a, rev):
        return np.ascontiguousarray(a[..., ::-1]) if (rev and a.ndim >= 3) else a

    sl = lambda t, a, b: None if t is None else t[a:b]

    def fwd(u, delta, A, B, C, D_, z_, delta_bias_, delta_softplus, reverse=False, out_z_into=None, reverse_from=0, for_backward=True):
        if reverse_from:   # by definition (vms_hip.h ABI v5): the two sub-batches, the second one right-to-left
            k, n = reverse_from, u.shape[0]
            lo = fwd(u[:k], delta[:k], A, sl(B, 0, k) if B.dim() >= 3 else B, sl(C, 0, k) if C.dim() >= 3 else C, D_, sl(z_, 0, k),
                     delta_bias_, delta_softplus, False, sl(out_z_into, 0, k))
            hi = fwd(u[k:], delta[k:], A, sl(B, k, n) if B.dim() >= 3 else B, sl(C, k, n) if C.dim() >= 3 else C, D_, sl(z_, k, n),
                     delta_bias_, delta_softplus, True, sl(out_z_into, k, n))
            out = torch.empty_like(delta)
            out[:k], out[k:] = lo[0], hi[0]
            res = [out, torch.cat([lo[1], hi[1]], dim=0)]
            if z_ is not None:
                if out_z_into is not None:
                    res.append(out_z_into)
                else:
                    oz = torch.empty_like(z_)
                    oz[:k], oz[k:] = lo[2], hi[2]
                    res.append(oz)
            return res
        rv = reverse
        if A.is_complex():   # the complex kernels' stand-in (no right-to-left mode in this fake)
            assert not rv and out_z_into is None
            r = oracle.cscan_fwd(np_(u), np_(delta), npc(A), npc(B), npc(C), np_(D_), np_(z_), np_(delta_bias_), delta_softplus,
                                 prec="f64")
            res = [torch.empty_like(delta).copy_(torch.from_numpy(r["out"])), torch.from_numpy(r["x"])]
            if z_ is not None:
                res.append(torch.empty_like(z_).copy_(torch.from_numpy(r["out_z"])))
            return res
        r = oracle.scan_fwd(np_(u, rv), np_(delta, rv), np_(A), np_(B, rv), np_(C, rv), np_(D_), np_(z_, rv),
                            np_(delta_bias_), delta_softplus, prec="f64")
        out = torch.empty_like(delta).copy_(torch.from_numpy(un(r["out"], rv)))
        res = [out, torch.from_numpy(r["x"])]
        if z_ is not None:
            oz = torch.from_numpy(un(r["out_z"], rv))
            res.append(torch.empty_like(z_).copy_(oz) if out_z_into is None else out_z_into.add_(oz.to(out_z_into.dtype)))
        return res

    def bwd(u, delta, A, B, C, D_, z_, delta_bias_, dout, x_, out_, dz_, delta_softplus, recompute_out_z,
            reverse=False, zeroed=None, keep_fp32=False, accumulate_dz=False, reverse_from=0):  # the scratch is the real shim's business
        if reverse_from:
            k, n = reverse_from, u.shape[0]
            vb, vc = B.dim() >= 3, C.dim() >= 3
            dz = dz_ if (dz_ is not None or z_ is None) else torch.empty_like(z_)
            parts = []
            for a, b, rv2 in ((0, k, False), (k, n, True)):
                parts.append(bwd(u[a:b], delta[a:b], A, B[a:b] if vb else B, C[a:b] if vc else C, D_, sl(z_, a, b), delta_bias_,
                                 dout[a:b], sl(x_, a, b), sl(out_, a, b), sl(dz, a, b), delta_softplus, recompute_out_z, rv2,
                                 None, keep_fp32, accumulate_dz))
            lo, hi = parts
            cat = lambda i: torch.cat([lo[i], hi[i]], dim=0)
            add = lambda i: None if lo[i] is None else lo[i] + hi[i]
            ddelta = torch.empty_like(delta)
            ddelta[:k], ddelta[k:] = lo[1], hi[1]
            res = [cat(0), ddelta, add(2), cat(3) if vb else add(3), cat(4) if vc else add(4), add(5), add(6)]
            if z_ is not None:
                res.append(dz)
            if recompute_out_z:
                res.append(cat(len(lo) - 1))
            return res
        rv = reverse
        if A.is_complex():
            assert not rv and not recompute_out_z
            r = oracle.cscan_bwd(np_(u), np_(delta), npc(A), npc(B), npc(C), np_(D_), np_(z_), np_(delta_bias_), np_(dout),
                                 delta_softplus, prec="f64")
        else:
            r = oracle.scan_bwd(np_(u, rv), np_(delta, rv), np_(A), np_(B, rv), np_(C, rv), np_(D_), np_(z_, rv),
                                np_(delta_bias_), np_(dout, rv), delta_softplus, prec="f64")
        r = {k: (un(v, rv) if v is not None else None) for k, v in r.items()}
        tt = lambda a, like: torch.from_numpy(a).to(like.dtype)
        res = [tt(r["du"], u), torch.empty_like(delta).copy_(tt(r["ddelta"], delta)), tt(r["dA"], A),
               tt(r["dB"], B), tt(r["dC"], C),
               tt(r["dD"], D_) if D_ is not None else None,
               tt(r["ddelta_bias"], delta_bias_) if delta_bias_ is not None else None]
        if z_ is not None:
            dz = dz_ if dz_ is not None else torch.empty_like(z_)
            if accumulate_dz:
                dz.add_(tt(r["dz"], z_))
            else:
                dz.copy_(tt(r["dz"], z_))
            res.append(dz)
        if recompute_out_z:
            f = oracle.scan_fwd(np_(u, rv), np_(delta, rv), np_(A), np_(B, rv), np_(C, rv), np_(D_), np_(z_, rv),
                                np_(delta_bias_), delta_softplus, prec="f64")
            res.append(torch.from_numpy(un(f["out_z"], rv)).to(u.dtype))
        return res

    def cfwd(x, w, b, silu, reverse=False, reverse_from=0):
        if reverse_from:
            k = reverse_from
            return torch.cat([cfwd(x[:k], w, b, silu, False), cfwd(x[k:], w, b, silu, True)], dim=0)
        return torch.from_numpy(un(oracle.conv_fwd(np_(x, reverse), np_(w), np_(b), silu, prec="f64"), reverse)).to(x.dtype)

    def cbwd(x, w, b, dout, dx_, silu, reverse=False, zeroed=None, accumulate_dx=False, reverse_from=0):
        if reverse_from:
            k = reverse_from
            dx = dx_ if dx_ is not None else torch.empty_like(x)
            lo = cbwd(x[:k], w, b, dout[:k], dx[:k], silu, False, None, accumulate_dx)
            hi = cbwd(x[k:], w, b, dout[k:], dx[k:], silu, True, None, accumulate_dx)
            return [dx, lo[1] + hi[1], lo[2] + hi[2] if b is not None else None]
        r = oracle.conv_bwd(np_(x, reverse), np_(w), np_(b), np_(dout, reverse), silu, prec="f64")
        dx = dx_ if dx_ is not None else torch.empty_like(x)
        if accumulate_dx:
            dx.add_(torch.from_numpy(un(r["dx"], reverse)).to(dx.dtype))
        else:
            dx.copy_(torch.from_numpy(un(r["dx"], reverse)))
        return [dx, torch.from_numpy(r["dweight"]), torch.from_numpy(r["dbias"]) if b is not None else None]


    def n_acc(A, B, C, D_, delta_bias_):
        return A.numel() + B.numel() + C.numel() + (D_.numel() if D_ is not None else 0) + (
            delta_bias_.numel() if delta_bias_ is not None else 0)

    fs = types.SimpleNamespace(fwd=fwd, bwd=bwd, bwd_accumulator_elems=n_acc)
    fc = types.SimpleNamespace(causal_conv1d_fwd=cfwd, causal_conv1d_bwd=cbwd)
    return fs, fc


def make_norm_fake(oracle):
    """layer_norm_cuda stand-in (same signatures as video-mamba-suite_amd/layer_norm_cuda.py)."""
    def np_(t):
        return None if t is None else t.detach().float().cpu().numpy()

    def nfwd(x, weight, bias, eps, residual=None, out_dtype=None, residual_dtype=None, is_rms_norm=False):
        if residual is not None:
            residual_dtype = residual.dtype
        r = oracle.norm_fwd(np_(x), np_(weight), np_(bias), np_(residual), eps, is_rms_norm, prec="f64")
        y = torch.from_numpy(r["y"]).to(x.dtype)
        res_out = None
        if residual is not None or (residual_dtype is not None and residual_dtype != x.dtype):
            res_out = torch.from_numpy(r["res_out"]).to(residual_dtype)
        mean = torch.from_numpy(r["mean"]) if not is_rms_norm else None
        return y, mean, torch.from_numpy(r["rstd"]), res_out

    def nbwd(dy, x, weight, bias, eps, mean, rstd, dresidual=None, has_residual=False, is_rms_norm=False,
             x_dtype=None):
        r = oracle.norm_bwd(np_(x), np_(weight), np_(mean), np_(rstd), np_(dy), np_(dresidual), is_rms_norm,
                            has_bias=bias is not None, prec="f64")
        dx = torch.from_numpy(r["ds"]).to(x_dtype or x.dtype)
        dw = torch.from_numpy(r["dw"]).to(weight.dtype)
        db = torch.from_numpy(r["db"]).to(bias.dtype) if bias is not None else None
        dres_in = None
        if has_residual:
            dres_in = dx if dx.dtype == x.dtype else torch.from_numpy(r["ds"]).to(x.dtype)
        return dx, dw, db, dres_in

    return types.SimpleNamespace(fwd=nfwd, bwd=nbwd)
