"""Stand-in for mamba/mamba_ssm/ops/triton/selective_state_update.py (reference: a Triton kernel,
:16-154).  Single-token SSM step used only by Mamba.step (autoregressive decode; no video task
runs it -- SURVEY.md 3.4, 8f-4).  Plain PyTorch with the semantics of the reference's
selective_state_update_ref (:157-192); `state` is updated in place."""
import torch
import torch.nn.functional as F


def selective_state_update(state, x, dt, A, B, C, D=None, z=None, dt_bias=None, dt_softplus=False):
    """state (batch, dim, dstate) [in/out]; x, dt, z (batch, dim); A (dim, dstate); B, C (batch, dstate);
    D, dt_bias (dim) -> out (batch, dim)"""
    if dt_bias is not None:
        dt = dt + dt_bias
    if dt_softplus:
        dt = F.softplus(dt)
    dA = torch.exp(dt[:, :, None] * A)
    dBx = (dt * x)[:, :, None] * B[:, None, :]
    state.copy_(state * dA + dBx)
    out = (state.to(C.dtype) * C[:, None, :]).sum(dim=-1)
    if D is not None:
        out = out + (x * D).to(out.dtype)
    if z is not None:
        out = out * F.silu(z)
    return out.to(x.dtype)


selective_state_update_ref = selective_state_update
