"""scan_fwd_sg_kernel (round 4): few rows, long sequences in ONE pass -- a row's 16 states over the four waves of a workgroup,
the per-element work shared through LDS -- against the row-per-wave kernel it replaces at those shapes (unsplit: vms_hip.debug.fwd_segments = 1)
and against the f64 oracle.  Reference: the chunk loop of mamba/csrc/selective_scan/selective_scan_fwd_kernel.cuh:131-132, 236-254."""
import numpy as np
import pytest
import torch

from test_hip_parity import _dbg, DEV, TOL, check, tol_for

pytestmark = pytest.mark.gpu


def _problem(b, d, L, itype, groups=1, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g)
    u = r(b, d, L).to(itype).to(DEV)
    delta = (0.5 * torch.rand(b, d, L, generator=g)).to(itype).to(DEV)
    A = (-0.5 * torch.rand(d, 16, generator=g) - 0.02).to(DEV)
    B = r(b, groups, 16, L).to(itype).to(DEV)
    C = r(b, groups, 16, L).to(itype).to(DEV)
    D = r(d).to(DEV)
    z = r(b, d, L).to(itype).to(DEV)
    bias = (0.5 * torch.rand(d, generator=g)).to(DEV)
    dout = r(b, d, L).to(itype).to(DEV)
    return u, delta, A, B, C, D, z, bias, dout


@pytest.mark.parametrize("layout1", [False, True])
@pytest.mark.parametrize("has_z", [True, False])
@pytest.mark.parametrize("reverse", [False, True])
@pytest.mark.parametrize("shape,groups,itype", [((1, 768, 4096), 1, torch.bfloat16), ((1, 768, 5136), 1, torch.bfloat16),
                                                ((2, 320, 4112), 2, torch.float16), ((1, 1024, 8192), 4, torch.bfloat16)])
def test_state_group_forward_equals_row_per_wave(monkeypatch, shape, groups, itype, reverse, has_z, layout1):
    import selective_scan_cuda as ssc
    import vms_hip
    if layout1:
        monkeypatch.setenv("VMS_X_LAYOUT", "1")
    b, d, L = shape
    u, delta, A, B, C, D, z, bias, dout = _problem(b, d, L, itype, groups, seed=L)
    zz = z if has_z else None
    sg = ssc.fwd(u, delta, A, B, C, D, zz, bias, True, reverse)
    assert vms_hip.last_kernel() == "scan_fwd_sg", vms_hip.last_kernel()
    monkeypatch.setattr(_dbg(), "fwd_segments", int("1"))
    ref = ssc.fwd(u, delta, A, B, C, D, zz, bias, True, reverse)
    assert vms_hip.last_kernel().startswith("scan_fwd_pair"), vms_hip.last_kernel()
    monkeypatch.setattr(_dbg(), "fwd_segments", 0)
    assert vms_hip.x_layout_of(sg[1], 16) == vms_hip.x_layout_of(ref[1], 16) == (1 if layout1 else 3)
    # same arithmetic per state; y is summed over the states in another order (4 partial sums of 4): an ulp of the 16-bit output
    ulp = 2.0 ** -8 if itype == torch.bfloat16 else 2.0 ** -11
    for name, a, r in zip(("out", "x", "out_z"), sg, ref):
        if name == "x":
            assert torch.equal(a, r), "reference-shaped checkpoints"
            full = lambda t: t.as_strided((b, d, t.shape[2], t.stride(2)), t.stride())
            fa, fr = full(a), full(r)
            n_el = (18 if layout1 else 258) * 16
            # every finer checkpoint the row-per-wave kernel wrote is written here, identically (positions past the end: untouched)
            valid = torch.zeros(fa.shape[2], n_el, dtype=torch.bool, device=DEV)
            for c in range(fa.shape[2]):
                if layout1:
                    for s_ in range(16):
                        valid[c, 32 + s_ * 16:32 + (s_ + 1) * 16] = c * 2048 + 128 * (s_ + 1) <= L
                else:
                    for i in range(256):
                        ok = c * 2048 + 8 * (i + 1) <= L
                        for n4 in range(4):
                            valid[c, 32 + (n4 * 256 + i) * 4:32 + (n4 * 256 + i) * 4 + 4] = ok
            assert torch.equal(fa[..., :n_el][:, :, valid], fr[..., :n_el][:, :, valid]), "finer checkpoints"
        else:
            assert (a.float() - r.float()).abs().max().item() <= 2 * ulp * r.float().abs().max().item(), name
            assert (a != r).float().mean().item() < 0.05, name
    # and the backward runs from them
    if has_z:
        ga = ssc.bwd(u, delta, A, B, C, D, z, bias, dout, sg[1], sg[0], None, True, False, reverse)
        gb = ssc.bwd(u, delta, A, B, C, D, z, bias, dout, ref[1], ref[0], None, True, False, reverse)
        for name, a, r in zip(("du", "ddelta", "dA", "dB", "dC"), ga, gb):
            check(a, r.float().cpu().numpy(), 1e-2 * (5 if name == "dA" else 2), name)


def test_state_group_forward_accumulates_and_mixes_directions(monkeypatch):
    """out_z_into (the second direction of a bidirectional block adds its gated output) and reverse_from (per-batch-entry
    direction, the stacked DBM form) through the state-group kernel == the row-per-wave kernels"""
    import selective_scan_cuda as ssc
    import vms_hip
    b, d, L = 2, 384, 4096
    u, delta, A, B, C, D, z, bias, _ = _problem(b, d, L, torch.bfloat16, seed=3)
    base = torch.randn(b, d, L, device=DEV).to(torch.bfloat16)
    res = {}
    for tag in ("sg", "rows"):
        if tag == "rows":
            monkeypatch.setattr(_dbg(), "fwd_segments", int("1"))
        acc = base.clone()
        o = ssc.fwd(u, delta, A, B, C, D, z, bias, True, True, out_z_into=acc)
        k1 = vms_hip.last_kernel()
        m = ssc.fwd(u, delta, A, B, C, D, z, bias, True, reverse_from=1)
        k2 = vms_hip.last_kernel()
        res[tag] = (o[2], m[0], m[2], m[1], k1, k2)
    assert res["sg"][4] == "scan_fwd_sg" and res["sg"][5] == "scan_fwd_sg+mixed", res["sg"][4:]
    assert res["rows"][4].startswith("scan_fwd_pair") and res["rows"][5].startswith("scan_fwd_pair")
    for i, name in enumerate(("out_z accumulated", "out mixed", "out_z mixed")):
        a, r = res["sg"][i].float(), res["rows"][i].float()
        assert (a - r).abs().max().item() <= 2 ** -7 * r.abs().max().item(), name
    assert torch.equal(res["sg"][3], res["rows"][3])


def test_state_group_forward_vs_oracle(oracle):
    import selective_scan_cuda as ssc
    import vms_hip
    b, d, L = 1, 768, 4096
    u, delta, A, B, C, D, z, bias, _ = _problem(b, d, L, torch.bfloat16, seed=11)
    out, x, out_z = ssc.fwd(u, delta, A, B, C, D, z, bias, True)
    assert vms_hip.last_kernel() == "scan_fwd_sg"
    rows = [0, 1, 383, 767]
    f = lambda t: np.ascontiguousarray(t.detach().float().cpu().numpy())
    o = oracle.scan_fwd(f(u[:, rows]), f(delta[:, rows]), f(A[rows]), f(B), f(C), f(D[rows]), f(z[:, rows]), f(bias[rows]), True, prec="f64")
    check(out[:, rows], o["out"], TOL[torch.bfloat16], "out rows vs oracle")
    check(out_z[:, rows], o["out_z"], TOL[torch.bfloat16], "out_z rows vs oracle")
    check(x[:, rows, -1, 1::2], o["last_state"], 1e-3, "last_state vs oracle")


def test_state_group_forward_with_the_reference_shaped_x():
    """A C caller's dense x (pitch 2 * dstate, no finer checkpoints: vms_hip.h x_has_sub == 0) through the state-group kernel"""
    import vms_hip
    b, d, L = 1, 768, 4096
    u, delta, A, B, C, D, z, bias, _ = _problem(b, d, L, torch.bfloat16, seed=21)
    res = {}
    for tag, seg in (("sg", None), ("rows", "1")):
        out, out_z = torch.empty_like(delta), torch.empty_like(z)
        x = torch.full((b, d, 2, 32), float("nan"), device=DEV)
        vms_hip.debug.fwd_segments = int(seg) if seg else 0
        try:
            vms_hip.scan_fwd(u, delta, A, B, C, D, z, bias, out, out_z, x, True)
        finally:
            vms_hip.debug.fwd_segments = 0
        res[tag] = (out, out_z, x, vms_hip.last_kernel())
    assert res["sg"][3] == "scan_fwd_sg" and res["rows"][3].startswith("scan_fwd_pair"), (res["sg"][3], res["rows"][3])
    assert torch.equal(res["sg"][2], res["rows"][2]) and not torch.isnan(res["sg"][2]).any()
    for i in (0, 1):
        a, r = res["sg"][i].float(), res["rows"][i].float()
        assert (a - r).abs().max().item() <= 2 ** -7 * r.abs().max().item()
