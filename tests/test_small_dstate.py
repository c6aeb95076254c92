"""GPU parity tests of the dstate 4 / 8 instantiations of the fast scan kernels (round 6; VERDICT r5 "missing" #1, "next" #2).

The suite's CLIP ViViM (avion/models/model_clip.py:945-947: ssm_cfg = dict(d_state=4), embed_dim 192 -> d_inner 384, dt_rank 12) ran on
the generic scan kernels and the unfused backward tail.  Since round 6 whole-vector rows with dstate 4 or 8 run on the LDS forward
kernel (`scan_fwd_pair_lds_n`) and the four-rows-per-wave backward (`scan_bwd_pair4_n`, `scan_bwd_pair4_dual_n`), and the fused
tail takes k = dt_rank + 2 dstate <= 32.  Checked here: against the f64 oracle on the inputs the kernels saw, against the generic
kernels, both checkpoint layouts, both directions, per-batch-entry directions (the DBM node), the two-direction call, the kernel
names, and the fused tail / head against their unfused compositions at k = 20.  The module-level d_state = 4 fixtures are in
tests/test_hip_parity.py::test_block_vs_golden."""
import numpy as np
import pytest
import torch

from test_hip_parity import DEV, TOL, _dbg, check, rel_err, run_scan

pytestmark = pytest.mark.gpu


def _problem(batch, dim, N, L, groups=1, has_z=True, seed=0):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g)
    d = dict(u=r(batch, dim, L), delta=0.5 * torch.rand(batch, dim, L, generator=g), A=-0.5 * torch.rand(dim, N, generator=g),
             B=r(batch, groups, N, L), C=r(batch, groups, N, L), D=r(dim), delta_bias=0.5 * torch.rand(dim, generator=g),
             g=r(batch, dim, L), softplus=1)
    if has_z:
        d["z"] = r(batch, dim, L)
    return {k: (v.numpy() if torch.is_tensor(v) else v) for k, v in d.items()}


# (batch, dim, L, groups): one chunk, a 16-element tail chunk (the four-states-at-a-time tail form), two 2048-blocks of checkpoints,
# two B / C groups of 32 rows
SHAPES = [(2, 64, 528, 1), (1, 96, 1040, 1), (1, 32, 4368, 1), (2, 64, 272, 2)]


@pytest.mark.parametrize("layout", [3, 1])
@pytest.mark.parametrize("itype", [torch.bfloat16, torch.float32, torch.float16])
@pytest.mark.parametrize("N", [4, 8])
@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("has_z", [True, False])
def test_small_dstate_vs_oracle(oracle, monkeypatch, shape, N, itype, layout, has_z):
    import vms_hip
    monkeypatch.setenv("VMS_X_LAYOUT", str(layout))
    batch, dim, L, groups = shape
    g = _problem(batch, dim, N, L, groups, has_z, seed=N + L)
    got, want = run_scan(g, itype, oracle)
    # (vms_last_kernel() is thread-local: this thread saw the forward; the backward's kernel is asserted in the tests below)
    assert vms_hip.last_kernel() == "scan_fwd_pair_lds_n", vms_hip.last_kernel()
    tol = TOL[itype]
    for k in ("out", "last_state", "du", "ddelta", "dB", "dC", "dz"):
        if want.get(k) is not None:
            check(got[k], want[k], tol * (2 if k != "out" else 1), k)
    for k in ("dA", "dD", "ddelta_bias"):
        check(got[k], want[k], tol * 5, k)


@pytest.mark.parametrize("reverse", [False, True])
@pytest.mark.parametrize("N", [4, 8])
def test_small_dstate_kernels_and_generic(monkeypatch, N, reverse):
    """the fast kernels' names, and their results against the generic kernels' (same inputs, fp32 sums in another order)"""
    import selective_scan_cuda as ssc
    import vms_hip
    b, d, L = 2, 64, 1040
    torch.manual_seed(N)
    dt = torch.bfloat16
    u = torch.randn(b, d, L, device=DEV).to(dt)
    delta = (0.5 * torch.rand(b, d, L, device=DEV)).to(dt)
    A = -0.5 * torch.rand(d, N, device=DEV)
    B = torch.randn(b, 1, N, L, device=DEV).to(dt)
    C = torch.randn(b, 1, N, L, device=DEV).to(dt)
    D = torch.randn(d, device=DEV)
    z = torch.randn(b, d, L, device=DEV).to(dt)
    bias = 0.5 * torch.rand(d, device=DEV)
    dout = torch.randn(b, d, L, device=DEV).to(dt)
    res = {}
    for impl in ("pair", "generic"):
        monkeypatch.setattr(_dbg(), "scan_impl", impl)
        out, x, out_z = ssc.fwd(u, delta, A, B, C, D, z, bias, True, reverse=reverse)
        kf = vms_hip.last_kernel()
        grads = ssc.bwd(u, delta, A, B, C, D, z, bias, dout, x, out, None, True, False, reverse=reverse, keep_fp32=True)
        kb = vms_hip.last_kernel()
        res[impl] = (out, out_z, x[:, :, -1, 1::2], grads, kf, kb)
    assert res["pair"][4] == "scan_fwd_pair_lds_n" and res["pair"][5] == "scan_bwd_pair4_n", res["pair"][4:]
    assert res["generic"][4] == "scan_fwd_generic" and res["generic"][5] == "scan_bwd_generic"
    check(res["pair"][0], res["generic"][0], 2 ** -7, "out")
    check(res["pair"][1], res["generic"][1], 2 ** -7, "out_z")
    check(res["pair"][2], res["generic"][2], 1e-4, "last_state")
    for k, name in enumerate(["du", "ddelta", "dA", "dB", "dC", "dD", "ddelta_bias", "dz"]):
        check(res["pair"][3][k], res["generic"][3][k], 2 ** -6 if name in ("du", "ddelta", "dz") else 2e-3, name)


@pytest.mark.parametrize("N", [4, 8])
def test_small_dstate_mixed_directions(monkeypatch, N):
    """reverse_from (the DBM block as one node): == the two single-direction problems"""
    import selective_scan_cuda as ssc
    import vms_hip
    b, d, L = 4, 64, 528
    torch.manual_seed(10 + N)
    dt = torch.bfloat16
    u = torch.randn(b, d, L, device=DEV).to(dt)
    delta = (0.5 * torch.rand(b, d, L, device=DEV)).to(dt)
    A = -0.5 * torch.rand(d, N, device=DEV)
    B = torch.randn(b, 1, N, L, device=DEV).to(dt)
    C = torch.randn(b, 1, N, L, device=DEV).to(dt)
    D = torch.randn(d, device=DEV)
    z = torch.randn(b, d, L, device=DEV).to(dt)
    bias = 0.5 * torch.rand(d, device=DEV)
    dout = torch.randn(b, d, L, device=DEV).to(dt)
    out, x, out_z = ssc.fwd(u, delta, A, B, C, D, z, bias, True, reverse_from=2)
    assert vms_hip.last_kernel() == "scan_fwd_pair_lds_n+mixed", vms_hip.last_kernel()
    g = ssc.bwd(u, delta, A, B, C, D, z, bias, dout, x, out, None, True, False, keep_fp32=True, reverse_from=2)
    assert vms_hip.last_kernel() == "scan_bwd_pair4_n+mixed", vms_hip.last_kernel()
    parts, outs = [], []
    for s, rev in ((slice(0, 2), False), (slice(2, 4), True)):
        o, xx, oz = ssc.fwd(u[s], delta[s], A, B[s], C[s], D, z[s], bias, True, reverse=rev)
        outs.append(oz)
        parts.append(ssc.bwd(u[s], delta[s], A, B[s], C[s], D, z[s], bias, dout[s], xx, o, None, True, False, reverse=rev, keep_fp32=True))
    assert rel_err(out_z, torch.cat(outs)) == 0.0
    for k, name in enumerate(["du", "ddelta", "dA", "dB", "dC", "dD", "ddelta_bias", "dz"]):
        ref = torch.cat([parts[0][k], parts[1][k]]) if name in ("du", "ddelta", "dB", "dC", "dz") else parts[0][k] + parts[1][k]
        check(g[k], ref, 0.0 if name in ("du", "ddelta", "dz") else 2e-5, name)   # (fp32 atomics: order-dependent in the last bits)


@pytest.mark.parametrize("layout1", [False, True])
@pytest.mark.parametrize("N", [4, 8])
def test_small_dstate_dual_backward(oracle, monkeypatch, N, layout1):
    """both directions of a bidirectional block in one grid (vms_selective_scan_bwd_dual) at dstate 4 / 8: against the f64 oracle"""
    import selective_scan_cuda as ssc
    import vms_hip
    if layout1:
        monkeypatch.setenv("VMS_X_LAYOUT", "1")
    b, d, L = 8, 512, 272
    dt = torch.bfloat16
    gen = torch.Generator().manual_seed(20 + N)

    def side():
        r = lambda *s: torch.randn(*s, generator=gen)
        return [r(b, d, L).to(dt).to(DEV), (0.5 * torch.rand(b, d, L, generator=gen)).to(dt).to(DEV), (-0.5 * torch.rand(d, N, generator=gen)).to(DEV),
                r(b, 1, N, L).to(dt).to(DEV), r(b, 1, N, L).to(dt).to(DEV), r(d).to(DEV), (0.5 * torch.rand(d, generator=gen)).to(DEV)]
    a, bb = side(), side()
    z = torch.randn(b, d, L, generator=gen).to(dt).to(DEV)
    dout = torch.randn(b, d, L, generator=gen).to(dt).to(DEV)
    fw = [ssc.fwd(*t[:6], z, t[6], True, reverse=(i == 1)) for i, t in enumerate((a, bb))]
    dz = torch.full_like(z, float("nan"))
    da, db = ssc.bwd_dual((*a[:6], a[6], fw[0][1], fw[0][0]), (*bb[:6], bb[6], fw[1][1], fw[1][0]), z, dout, dz, True, keep_fp32=True)
    assert vms_hip.last_kernel() == "scan_bwd_pair4_dual_n", vms_hip.last_kernel()
    f = lambda t: t.detach().float().cpu().numpy()
    names = ["du", "ddelta", "dA", "dB", "dC", "dD", "ddelta_bias"]
    dz_want = 0
    for got, inp, rev in ((da, a, False), (db, bb, True)):
        u, delta, A, B, C, D, bias = [f(t) for t in inp]
        fl = (lambda t: np.ascontiguousarray(t[..., ::-1])) if rev else (lambda t: t)
        o = oracle.scan_bwd(fl(u), fl(delta), A, fl(B), fl(C), D, fl(f(z)), bias, fl(f(dout)), True, prec="f64")
        for k, name in enumerate(names):
            ref = o[name][..., ::-1] if rev and name in ("du", "ddelta", "dB", "dC") else o[name]
            check(got[k], ref, TOL[dt] * (5 if name in ("dA", "dD", "ddelta_bias") else 2), f"{'b' if rev else 'a'}.{name}")
        dz_want = dz_want + (o["dz"][..., ::-1] if rev else o["dz"])
    check(da[7], dz_want, TOL[dt] * 2, "dz (both directions)")


def test_fused_tail_and_head_at_k20(monkeypatch):
    """the CLIP ViViM mixer's own sizes (d_inner 384, dt_rank 12, d_state 4: x_dbl has 20 rows): the block step lands on the fast scan
    kernels and on the fused head / tail (vms_proj_conv_bwd took k >= 33 only until ABI v11), and equals the step with the tail unfused"""
    import vms_hip
    from mamba_ssm.modules.mamba_simple import Mamba
    torch.manual_seed(0)
    m = Mamba(192, d_state=4, d_conv=4, expand=2, bimamba_type="v2", if_devide_out=True).to(DEV)
    x = torch.randn(4, 784, 192, device=DEV, dtype=torch.bfloat16, requires_grad=True)
    g = torch.randn(4, 784, 192, device=DEV, dtype=torch.bfloat16)

    def step():
        m.zero_grad(set_to_none=True)
        x.grad = None
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = m(x)
        y.backward(g)
        return y.detach().clone(), x.grad.clone(), {k: p.grad.clone() for k, p in m.named_parameters()}

    vms_hip.start_timing(reserve=64)
    y1, dx1, g1 = step()
    torch.cuda.synchronize()
    vms_hip._timing = None
    recs = vms_hip.ext().timing_stop_detail() if vms_hip.ext() is not None else []
    if recs:
        kernels = {k for _, _, k in recs}
        entries = [e for e, _, _ in recs]
        assert "scan_fwd_pair_lds_n" in kernels and ("scan_bwd_pair4_dual_n" in kernels or "scan_bwd_pair4_n" in kernels), kernels
        assert entries.count("vms_proj_conv_bwd") == 2 and "vms_conv_xproj_dual" in entries, entries
    monkeypatch.setattr(_dbg(), "no_fused_tail", True)
    y0, dx0, g0 = step()
    assert rel_err(y1, y0) == 0.0
    check(dx1, dx0, 2e-2, "dx: fused vs unfused tail")
    for k in g0:
        check(g1[k], g0[k], 2e-2, f"{k}: fused vs unfused tail")
