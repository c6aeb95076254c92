"""causal_conv1d_fn against causal_conv1d_ref the way the reference's own test runs it (causal-conv1d/tests/test_causal_conv1d.py:14-75): its
full grid at its own size -- batch 2, dim 4096 + 32 cut out of a wider tensor ("dim not divisible by 64", a non-trivial batch stride),
both memory layouts, widths 2-4, with / without bias and SiLU, the three dtypes, all 14 sequence lengths -- kernel vs the PyTorch
statement on the GPU under ITS rtol / atol.  (tests/test_hip_parity.py::test_conv_reference_grid holds the oracle version at dim 264.)"""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("channel_last", [False, True])
@pytest.mark.parametrize("itype", [torch.float32, torch.float16, torch.bfloat16])
@pytest.mark.parametrize("silu_activation", [False, True])
@pytest.mark.parametrize("has_bias", [False, True])
@pytest.mark.parametrize("width", [2, 3, 4])
@pytest.mark.parametrize("seqlen", [8, 16, 32, 64, 128, 151, 256, 372, 512, 784, 1024, 1134, 2048, 4096])
def test_causal_conv1d_reference_grid_at_reference_size(seqlen, width, has_bias, silu_activation, itype, channel_last):
    from causal_conv1d.causal_conv1d_interface import causal_conv1d_fn, causal_conv1d_ref
    rtol, atol = (3e-4, 1e-3) if itype == torch.float32 else (3e-3, 5e-3)      # :31-34
    if itype == torch.bfloat16:
        rtol, atol = 1e-2, 5e-2
    rtolw, atolw = 1e-3, 1e-3
    torch.manual_seed(0)
    b, dim, lo = 2, 4096 + 32, 4096
    if channel_last:
        x = torch.randn(b, seqlen, lo + dim + 64, device=DEV, dtype=itype)[:, :, lo:lo + dim].transpose(1, 2)
    else:
        x = torch.randn(b, lo + dim + 64, seqlen, device=DEV, dtype=itype)[:, lo:lo + dim, :]
    x = x.requires_grad_()
    weight = torch.randn(dim, width, device=DEV, requires_grad=True)
    bias = torch.randn(dim, device=DEV, requires_grad=True) if has_bias else None
    x_r = x.detach().clone().requires_grad_()
    w_r = weight.detach().clone().requires_grad_()
    b_r = bias.detach().clone().requires_grad_() if has_bias else None
    act = "silu" if silu_activation else None
    out = causal_conv1d_fn(x, weight, bias, activation=act)
    out_ref = causal_conv1d_ref(x_r, w_r, b_r, activation=act)
    assert torch.allclose(out, out_ref, rtol=rtol, atol=atol), f"out {(out - out_ref).abs().max().item():.3e}"
    g = torch.randn_like(out)
    out_ref.backward(g)
    out.backward(g)
    assert torch.allclose(x.grad, x_r.grad.to(itype), rtol=rtol, atol=atol), f"dx {(x.grad - x_r.grad).abs().max().item():.3e}"
    assert torch.allclose(weight.grad, w_r.grad, rtol=rtolw, atol=atolw), f"dweight {(weight.grad - w_r.grad).abs().max().item():.3e}"
    if has_bias:
        assert torch.allclose(bias.grad, b_r.grad, rtol=rtolw, atol=atolw), f"dbias {(bias.grad - b_r.grad).abs().max().item():.3e}"
