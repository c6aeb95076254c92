"""mamba_ssm.ops.triton.selective_state_update over the gfx950 kernel (reference: a Triton kernel,
mamba/mamba_ssm/ops/triton/selective_state_update.py:16-154; the module path is kept because the modules
import it).  Single-token SSM step used by Mamba.step (autoregressive decode; no video task runs it --
SURVEY.md 8f-4).  `state` is updated in place.  selective_state_update_ref is the reference's pure-PyTorch
function (:157-192) and runs anywhere; selective_state_update has no CPU path."""
import torch
import torch.nn.functional as F

import vms_hip as _k


def selective_state_update(state, x, dt, A, B, C, D=None, z=None, dt_bias=None, dt_softplus=False):
    """state (batch, dim, dstate) [in/out]; x, dt, z (batch, dim); A (dim, dstate); B, C (batch, dstate);
    D, dt_bias (dim) -> out (batch, dim)"""
    batch, dim, dstate = state.shape
    assert x.shape == (batch, dim) and dt.shape == x.shape and A.shape == (dim, dstate)
    assert B.shape == (batch, dstate) and C.shape == B.shape
    assert D is None or D.shape == (dim,)
    assert z is None or z.shape == x.shape
    assert dt_bias is None or dt_bias.shape == (dim,)
    # dt and z keep their own dtypes (the reference's kernel loads every tensor in its dtype and widens to fp32)
    if C.dtype != B.dtype:
        C = C.to(B.dtype)
    wd = A.dtype
    D = D.to(wd).contiguous() if D is not None else None
    dt_bias = dt_bias.to(wd).contiguous() if dt_bias is not None else None
    out = torch.empty_like(x)
    _k.state_update(state, x, dt, A, B, C, D, z, dt_bias, out, dt_softplus)
    return out


def selective_state_update_ref(state, x, dt, A, B, C, D=None, z=None, dt_bias=None, dt_softplus=False):
    if dt_bias is not None:
        dt = dt + dt_bias
    dt = F.softplus(dt) if dt_softplus else dt
    dA = torch.exp(dt[:, :, None] * A)
    dB = dt[:, :, None] * B[:, None, :]
    state.copy_(state * dA + dB * x[:, :, None])
    out = torch.einsum("bdn,bn->bd", state.to(C.dtype), C)
    if D is not None:
        out += (x * D).to(out.dtype)
    return (out if z is None else out * F.silu(z)).to(x.dtype)
