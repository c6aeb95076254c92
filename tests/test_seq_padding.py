"""Ragged sequences through the mixer's zero padding (modules/_core.py _SEQ_PAD): the padded run on the whole-vector kernels must
give what the unpadded run on the ragged kernels gives -- outputs, input gradient and every parameter gradient -- because
softplus(-inf) = 0 turns the padding into identity steps of both directions' recurrences (selective_scan_interface._mask_padding).
The unpadded path is the one the golden fixtures and the oracle pin (tests/test_hip_parity.py)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(mod, hidden, gout, autocast):
    mod.zero_grad(set_to_none=True)
    h = hidden.detach().clone().requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
        out = mod(h)
    out.float().backward(gout)
    return out.detach().float(), h.grad.detach().float(), {n: p.grad.detach().float().clone() for n, p in mod.named_parameters()}


def _rel(a, b):
    return (a - b).abs().max().item() / max(b.abs().max().item(), 1e-6)


def _make(variant, d_model, **kw):
    if variant == "dbm":   # round 6: the DBM mixer pads too (its second half is scanned right-to-left by the kernels: padding at the physical end)
        from mamba_ssm.modules.mamba_new import Mamba
        torch.manual_seed(0)
        return Mamba(d_model, d_state=16, d_conv=4, expand=1, **kw).cuda()
    if variant == "vim_norm":
        from mamba_ssm.modules.mamba_simple_scan_norm import Mamba
    else:
        from mamba_ssm.modules.mamba_simple import Mamba
    torch.manual_seed(0)
    return Mamba(d_model, d_state=16, d_conv=4, expand=2, bimamba_type="v2", **kw).cuda()


@pytest.mark.parametrize("checkpoint_lvl", [0, 1])
@pytest.mark.parametrize("variant,kw", [("vim", {}), ("vim", {"if_devide_out": True}), ("vim_norm", {"if_devide_out": True}), ("dbm", {})])
@pytest.mark.parametrize("batch,seqlen,d_model", [(2, 197, 64), (1, 1569, 32), (3, 35, 48), (2, 1, 32), (1, 188, 64), (4, 107, 32)])
def test_padding_fp32_equals_ragged(monkeypatch, batch, seqlen, d_model, variant, kw, checkpoint_lvl):
    """fp32 end to end: the two runs differ by kernel generation only (summation order), so they agree to 1e-4 -- any value leaking
    out of the padding (a state picked up by the right-to-left scan, a gradient through the padding's delta) would show whole"""
    from mamba_ssm.modules import _core
    monkeypatch.setattr(_core, "_CHECKPOINT_LVL", checkpoint_lvl)
    monkeypatch.setattr(_core, "_SEQ_PAD_FP32", True)
    mod = _make(variant, d_model, **kw)
    torch.manual_seed(1)
    hidden = torch.randn(batch, seqlen, d_model, device="cuda")
    gout = torch.randn(batch, seqlen, d_model, device="cuda")
    monkeypatch.setattr(_core, "_SEQ_PAD", 0)
    o0, dh0, g0 = _run(mod, hidden, gout, False)
    monkeypatch.setattr(_core, "_SEQ_PAD", 16)
    assert mod._seq_padding(hidden) == ((-seqlen) % 8 if seqlen <= 16 else (-seqlen) % 16)
    o1, dh1, g1 = _run(mod, hidden, gout, False)
    assert o1.shape == o0.shape and dh1.shape == dh0.shape
    assert _rel(o1, o0) < 1e-4, ("out", _rel(o1, o0))
    assert _rel(dh1, dh0) < 1e-4, ("dhidden", _rel(dh1, dh0))
    for n in g0:
        assert _rel(g1[n], g0[n]) < 2e-4, (n, _rel(g1[n], g0[n]))


@pytest.mark.parametrize("batch,seqlen,d_model", [(2, 197, 64), (2, 1569, 96), (4, 393, 192)])
def test_padding_bf16_autocast_close_to_ragged_and_fp32(monkeypatch, batch, seqlen, d_model):
    """bf16 autocast (how the suite trains): padded and ragged runs are two roundings of the same values -- each must sit as close
    to the fp32 run as the other"""
    from mamba_ssm.modules import _core
    mod = _make("vim", d_model)
    torch.manual_seed(1)
    hidden = torch.randn(batch, seqlen, d_model, device="cuda")
    gout = torch.randn(batch, seqlen, d_model, device="cuda")
    monkeypatch.setattr(_core, "_SEQ_PAD", 0)
    ref = _run(mod, hidden, gout, False)
    rag = _run(mod, hidden, gout, True)
    monkeypatch.setattr(_core, "_SEQ_PAD", 16)
    pad = _run(mod, hidden, gout, True)
    for name, r, a, b in (("out", ref[0], rag[0], pad[0]), ("dhidden", ref[1], rag[1], pad[1])) + tuple(
            (n, ref[2][n], rag[2][n], pad[2][n]) for n in ref[2]):
        e_rag, e_pad = _rel(a, r), _rel(b, r)
        assert e_pad < max(2.0 * e_rag, 2e-2), (name, e_pad, e_rag)


def test_padding_not_applied_when_it_would_be_wrong():
    """an in_proj bias makes the padding's xz nonzero; aligned lengths and CPU tensors need none"""
    from mamba_ssm.modules.mamba_simple import Mamba
    m = Mamba(32, bimamba_type="v2", bias=True).cuda()
    assert m._seq_padding(torch.zeros(1, 197, 32, device="cuda", dtype=torch.bfloat16)) == 0
    m = Mamba(32, bimamba_type="v2").cuda()
    assert m._seq_padding(torch.zeros(1, 197, 32, device="cuda", dtype=torch.bfloat16)) == 11
    assert m._seq_padding(torch.zeros(1, 208, 32, device="cuda", dtype=torch.bfloat16)) == 0
    assert m._seq_padding(torch.zeros(1, 197, 32, device="cuda")) == 0          # fp32 activations, no autocast
    with torch.autocast("cuda", dtype=torch.bfloat16):
        assert m._seq_padding(torch.zeros(1, 197, 32, device="cuda")) == 11
    assert m._seq_padding(torch.zeros(1, 197, 32)) == 0


def test_mask_padding_makes_identity_steps():
    """the node itself: xz zero-padded + seq_valid == the unpadded node on the real positions, and the padding's gated output is 0"""
    from mamba_ssm.ops.selective_scan_interface import bimamba_inner_fn_no_out_proj
    torch.manual_seed(0)
    b, d, L, N, R, pad = 2, 64, 200, 16, 4, 8
    dev = "cuda"

    def params():
        return (torch.randn(d, 1, 4, device=dev) * 0.3, torch.randn(d, device=dev) * 0.3, torch.randn(R + 2 * N, d, device=dev) * d ** -0.5,
                torch.randn(d, R, device=dev) * R ** -0.5, -torch.rand(d, N, device=dev) - 0.2, torch.randn(d, device=dev),
                torch.rand(d, device=dev) - 3.0)
    pa, pb = params(), params()
    xz = torch.randn(b, 2 * d, L, device=dev)
    xzp = torch.nn.functional.pad(xz, (0, pad))
    ref = bimamba_inner_fn_no_out_proj(xz, pa, pb, checkpoint_lvl=0)
    got = bimamba_inner_fn_no_out_proj(xzp, pa, pb, checkpoint_lvl=0, seq_valid=L)
    assert got.shape[-1] == L + pad
    assert got[..., L:].abs().max().item() == 0.0
    assert _rel(got[..., :L].float(), ref.float()) < 1e-4


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_padding_inference_no_grad(monkeypatch, dtype):
    """forward only (torch.no_grad, eval), fp16 and bf16 autocast, an input that needs no gradient: padded == ragged within the
    dtype's rounding, and the output has the caller's shape"""
    from mamba_ssm.modules import _core
    mod = _make("vim", 96).eval()
    torch.manual_seed(2)
    hidden = torch.randn(3, 393, 96, device="cuda")
    outs = []
    for pad in (0, 16):
        monkeypatch.setattr(_core, "_SEQ_PAD", pad)
        with torch.no_grad(), torch.autocast("cuda", dtype=dtype):
            outs.append(mod(hidden).float())
    assert outs[0].shape == outs[1].shape == (3, 393, 96)
    assert torch.isfinite(outs[1]).all()
    assert _rel(outs[1], outs[0]) < (2e-2 if dtype == torch.bfloat16 else 5e-3)


def test_padding_block_stack_matches_ragged(monkeypatch):
    """two Blocks (fused add + RMSNorm -> ViM mixer) on a ragged length, bf16 autocast, forward + backward: the padded stack tracks
    the ragged one (the un-padded slices feed the next block's norm)"""
    from functools import partial
    from mamba_ssm.modules import _core
    from mamba_ssm.modules.mamba_simple import Block, Mamba
    from mamba_ssm.ops.triton.layernorm import RMSNorm
    torch.manual_seed(0)
    d = 128
    blocks = torch.nn.ModuleList([Block(d, partial(Mamba, bimamba_type="v2", layer_idx=i), norm_cls=partial(RMSNorm, eps=1e-5),
                                        fused_add_norm=True, residual_in_fp32=True) for i in range(2)]).cuda()
    x = torch.randn(2, 197, d, device="cuda")
    g = torch.randn(2, 197, d, device="cuda")

    def run():
        blocks.zero_grad(set_to_none=True)
        h = x.clone().requires_grad_(True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            hid, res = h, None
            for blk in blocks:
                hid, res = blk(hid, res)
        (hid.float() * g).sum().backward()
        return hid.detach().float(), h.grad.float(), [p.grad.float().clone() for p in blocks.parameters()]
    monkeypatch.setattr(_core, "_SEQ_PAD", 0)
    o0, dh0, g0 = run()
    monkeypatch.setattr(_core, "_SEQ_PAD", 16)
    o1, dh1, g1 = run()
    assert _rel(o1, o0) < 2e-2 and _rel(dh1, dh0) < 3e-2
    for a, b in zip(g1, g0):
        assert _rel(a, b) < 3e-2


@pytest.mark.gpu
def test_padded_path_returns_a_contiguous_tensor():
    """the padded path used to return out[:, :seqlen], a non-contiguous slice of the padded result -- downstream code that
    .view()s the block's output (the reference mixer returns a contiguous (B, L, D) tensor) failed for ragged lengths only."""
    from mamba_ssm.modules.mamba_simple import Mamba
    torch.manual_seed(0)
    m = Mamba(64, d_state=16, expand=1, bimamba_type="v2").cuda()
    x = torch.randn(3, 197, 64, device="cuda", dtype=torch.bfloat16)
    assert m._seq_padding(x) == 11
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = m(x)
    assert y.shape == (3, 197, 64) and y.is_contiguous()
    y.view(3 * 197, 64)   # what a caller may do with the reference's output
