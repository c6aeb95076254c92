"""selective_scan_fn against selective_scan_ref the way the reference's own test runs it (mamba/tests/ops/test_selective_scan.py:17-150):
batch 2, dim 4, dstate 8, the inputs of its recipe, kernel vs the PyTorch statement on the GPU under ITS element-wise rtol / atol per
tensor -- over the grid the reference has switched on AND the values it keeps commented out beside it (both weight types, the three
activation dtypes, all 13 sequence lengths, every flag).  The oracle-based tests (test_hip_parity.py, test_parity_hardening.py) are the
tight ones; this file is the literal one."""
import itertools

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"
SEQLENS = [8, 16, 32, 64, 128, 256, 372, 512, 784, 1024, 1134, 2048, 4096]      # test_selective_scan.py:22 (the commented list)
FLAGS = list(itertools.product([False, True], repeat=7))                          # B var, C var, D, z, delta_bias, softplus, last_state


def _tolerances(itype, has_z):
    rtol, atol = (6e-4, 2e-3) if itype == torch.float32 else (3e-3, 5e-3)          # :45-51
    if itype == torch.bfloat16:
        rtol, atol = 3e-2, 5e-2
    rtolw, atolw = 1e-3, 1e-3
    if has_z:
        rtolw, atolw = max(rtolw, rtol), max(atolw, atol)
    return rtol, atol, rtolw, atolw


def _run(seqlen, itype, wtype, groups, var_B, var_C, has_D, has_z, has_bias, softplus, last_state):
    from mamba_ssm.ops.selective_scan_interface import selective_scan_fn, selective_scan_ref
    rtol, atol, rtolw, atolw = _tolerances(itype, has_z)
    torch.manual_seed(0)
    b, dim, N = 2, 4, 8
    cplx = wtype == torch.complex64
    mk = lambda *s, dtype=itype: torch.randn(*s, device=DEV, dtype=dtype)
    bc_len = seqlen * (2 if cplx else 1)                                          # a complex B / C travels as interleaved pairs (:62)
    bc_shape = (b, N, bc_len) if groups == 1 else (b, groups, N, bc_len)
    leaves = dict(
        A=-0.5 * torch.rand(dim, N, device=DEV, dtype=wtype),
        B=mk(*bc_shape) if var_B else mk(dim, N, dtype=wtype),
        C=mk(*bc_shape) if var_C else mk(dim, N, dtype=wtype),
        D=mk(dim, dtype=torch.float32) if has_D else None,
        z=mk(b, dim, seqlen) if has_z else None,
        delta_bias=0.5 * torch.rand(dim, device=DEV) if has_bias else None,
        u=mk(b, dim, seqlen),
        delta=(0.5 * torch.rand(b, dim, seqlen, device=DEV)).to(itype),
    )
    got_in = {k: None if v is None else v.detach().clone().requires_grad_() for k, v in leaves.items()}
    ref_in = {k: None if v is None else v.detach().clone().requires_grad_() for k, v in leaves.items()}

    def call(fn, t):
        r = fn(t["u"], t["delta"], t["A"], t["B"], t["C"], t["D"], z=t["z"], delta_bias=t["delta_bias"], delta_softplus=softplus,
               return_last_state=last_state)
        return r if last_state else (r, None)
    out, state = call(selective_scan_fn, got_in)
    out_ref, state_ref = call(selective_scan_ref, ref_in)
    assert torch.allclose(out, out_ref, rtol=rtol, atol=atol), f"out {(out - out_ref).abs().max().item():.3e}"
    if last_state:
        assert torch.allclose(state, state_ref, rtol=rtol, atol=atol), f"last_state {(state - state_ref).abs().max().item():.3e}"
    g = torch.randn_like(out)
    out_ref.backward(g)
    out.backward(g)
    # per tensor (rtol, atol): test_selective_scan.py:137-149
    bars = dict(u=(2 * rtol, 2 * atol), delta=(5 * rtol, 10 * atol), A=(rtolw, 5 * atolw),
                B=(rtol, atol) if var_B else (rtolw, atolw), C=(rtol, atol) if var_C else (rtolw, atolw),
                D=(rtolw, atolw), z=(rtolw, atolw), delta_bias=(rtolw, atolw))
    for k, (rt, at) in bars.items():
        if got_in[k] is None:
            continue
        a, r = got_in[k].grad, ref_in[k].grad.to(got_in[k].grad.dtype)
        assert a.shape == r.shape, k
        assert torch.allclose(a, r, rtol=rt, atol=at), f"d{k}: {(a - r).abs().max().item():.3e} (rtol {rt}, atol {at})"


@pytest.mark.parametrize("groups", [1, 2])
@pytest.mark.parametrize("wtype", [torch.float32, torch.complex64])
@pytest.mark.parametrize("itype", [torch.float32, torch.float16, torch.bfloat16])
@pytest.mark.parametrize("seqlen", SEQLENS)
def test_selective_scan_every_length_dtype_and_weight_type(seqlen, itype, wtype, groups):
    _run(seqlen, itype, wtype, groups, True, True, True, True, True, True, True)


@pytest.mark.parametrize("wtype", [torch.float32, torch.complex64])
@pytest.mark.parametrize("itype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("flags", FLAGS, ids=lambda f: "".join("1" if v else "0" for v in f))
def test_selective_scan_every_flag_combination(flags, itype, wtype):
    var_B, var_C, has_D, has_z, has_bias, softplus, last_state = flags
    _run(128, itype, wtype, 1, var_B, var_C, has_D, has_z, has_bias, softplus, last_state)
