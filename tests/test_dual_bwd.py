"""vms_selective_scan_bwd_dual (ABI v9): the backward scans of both directions of a bidirectional block in one call -- one
grid when the pair qualifies (4-wave or 8-wave workgroups), two launches otherwise -- against the two single-direction calls
it replaces (mamba/mamba_ssm/ops/selective_scan_interface.py:541-561: the reference runs selective_scan_cuda.bwd twice) and
against the f64 oracle; then the same through the block's one-node form."""
import numpy as np
import pytest
import torch

from test_hip_parity import _dbg, DEV, TOL, check, rel_err, tol_for

pytestmark = pytest.mark.gpu


def _dir_inputs(b, d, L, itype, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g)
    u = r(b, d, L).to(itype).to(DEV)
    delta = (0.5 * torch.rand(b, d, L, generator=g)).to(itype).to(DEV)
    A = (-0.5 * torch.rand(d, 16, generator=g) - 0.05).to(DEV)
    B = r(b, 1, 16, L).to(itype).to(DEV)
    C = r(b, 1, 16, L).to(itype).to(DEV)
    D = r(d).to(DEV)
    bias = (0.5 * torch.rand(d, generator=g)).to(DEV)
    return u, delta, A, B, C, D, bias


def _run_pair(b, d, L, itype, layout1, monkeypatch, seed=0):
    """-> (dual results, two-call results, kernel name, inputs) on identical inputs"""
    import selective_scan_cuda as ssc
    import vms_hip
    if layout1:
        monkeypatch.setenv("VMS_X_LAYOUT", "1")
    a, bb = _dir_inputs(b, d, L, itype, seed), _dir_inputs(b, d, L, itype, seed + 1)
    g = torch.Generator(device="cpu").manual_seed(seed + 2)
    z = torch.randn(b, d, L, generator=g).to(itype).to(DEV)
    dout = torch.randn(b, d, L, generator=g).to(itype).to(DEV)
    fw = []
    for i, (u, delta, A, B, C, D, bias) in enumerate((a, bb)):
        out, x, _ = ssc.fwd(u, delta, A, B, C, D, z, bias, True, reverse=(i == 1))
        fw.append((out, x))
    # the two single-direction calls, the second adding its dz to the first's (what BiMambaInnerFnNoOutProj did before v9)
    dz_ref = torch.empty_like(z)
    ra = ssc.bwd(*a[:6], z, a[6], dout, fw[0][1], fw[0][0], dz_ref, True, False, reverse=False, keep_fp32=True)
    rb = ssc.bwd(*bb[:6], z, bb[6], dout, fw[1][1], fw[1][0], dz_ref, True, False, reverse=True, keep_fp32=True, accumulate_dz=True)
    single_kernel = vms_hip.lib().vms_last_kernel().decode()
    dz = torch.full_like(z, float("nan"))
    da, db = ssc.bwd_dual((*a[:6], a[6], fw[0][1], fw[0][0]), (*bb[:6], bb[6], fw[1][1], fw[1][0]), z, dout, dz, True, keep_fp32=True)
    kernel = vms_hip.lib().vms_last_kernel().decode()
    torch.cuda.synchronize()
    return (da, db), (ra, rb), kernel, single_kernel, (a, bb, z, dout)


NAMES = ["du", "ddelta", "dA", "dB", "dC", "dD", "ddelta_bias"]


def _ulp(itype):
    return 2 ** -8 if itype == torch.bfloat16 else 2 ** -11


@pytest.mark.parametrize("layout1", [False, True])
@pytest.mark.parametrize("itype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("shape,want", [((5, 1024, 1040), "scan_bwd_pair4_dual_w4"),      # 320 8-wave workgroups: not whole rounds -> 4-wave
                                        ((4, 1024, 528), "scan_bwd_pair4_dual_w8"),       # 256: whole rounds of 8-wave workgroups
                                        ((8, 1024, 528), "scan_bwd_pair4_dual_w8"),       # the benchmark's grid (2 x 256 workgroups of 8 waves)
                                        ((8, 768, 400), "scan_bwd_pair4_dual_w4"),        # the suite's grid: 2 x 192
                                        ((1, 64, 2048), None)])                             # too small to fill the chip: two launches
def test_dual_equals_two_single_calls(monkeypatch, shape, want, itype, layout1):
    (da, db), (ra, rb), kernel, single_kernel, _ = _run_pair(*shape, itype, layout1, monkeypatch)
    if want is None:
        assert "dual" not in kernel, kernel
    else:
        assert kernel == want, kernel
    for got, ref, tag in ((da, ra, "a"), (db, rb, "b")):
        for k, name in enumerate(NAMES):
            # same kernel body, same order of operations within a row; the fp32 atomics (dB, dC over 32 / 16 rows, dA / dD /
            # dbias over the batch) are order-dependent in the last bits and the dual grid sums 16-row partials
            tol = 0.0 if name in ("du", "ddelta") else 2e-5
            e = rel_err(got[k], ref[k])
            assert e <= tol, f"{tag}.{name}: {e:.3e}"
    # dz: one rounding of dout (out_a + out_b) dsilu(z) against the two-call form's two roundings: within an ulp of the dtype
    check(da[7], ra[7], 2 * _ulp(itype), "dz dual vs accumulated")
    assert len(db) == 7 and not torch.isnan(da[7].float()).any()


# the oracle comparison at BOTH workgroup widths of the dual grid and with >= 2 chunks of 2048 (the headline's
# instantiation, 8-wave workgroups, used to meet the oracle only through the two single calls), for both checkpoint layouts
@pytest.mark.parametrize("layout1", [False, True])
@pytest.mark.parametrize("itype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("shape,want", [((8, 768, 272), "scan_bwd_pair4_dual_w4"),     # the suite's grid, short rows for the oracle
                                        ((8, 1024, 528), "scan_bwd_pair4_dual_w8"),    # the benchmark's grid: 8-wave workgroups in both bodies
                                        ((8, 512, 4368), "scan_bwd_pair4_dual_w8"),    # 3 chunks of 2048 (the last one partial), a 16-element tail chunk
                                        ((8, 1024, 2304), "scan_bwd_pair4_dual_w8")])  # 2 chunks of 2048, 8-wave workgroups
def test_dual_vs_oracle(oracle, monkeypatch, itype, shape, want, layout1):
    b, d, L = shape
    (da, db), _, kernel, _, (a, bb, z, dout) = _run_pair(b, d, L, itype, layout1, monkeypatch, seed=5)
    assert kernel == want, kernel
    f = lambda t: t.detach().float().cpu().numpy()
    dz_want = 0
    for got, inp, rev in ((da, a, False), (db, bb, True)):
        u, delta, A, B, C, D, bias = [f(t) for t in inp]
        fl = (lambda t: np.ascontiguousarray(t[..., ::-1])) if rev else (lambda t: t)
        o = oracle.scan_bwd(fl(u), fl(delta), A, fl(B), fl(C), D, fl(f(z)), bias, fl(f(dout)), True, prec="f64")
        for k, name in enumerate(NAMES):
            ref = o[name]
            if rev and name in ("du", "ddelta", "dB", "dC"):
                ref = ref[..., ::-1]
            check(got[k], ref, tol_for(name, itype, "oracle"), f"{'b' if rev else 'a'}.{name} vs oracle")
        dz_want = dz_want + (o["dz"][..., ::-1] if rev else o["dz"])
    check(da[7], dz_want, tol_for("dz", itype, "oracle"), "dz (both directions) vs oracle")


def test_dual_softplus_extremes(oracle, monkeypatch):
    """delta + bias far below 0 (softplus underflows towards 0), around 0 and above the reference's threshold of 20
    (softplus(t) = t, derivative 1)."""
    import selective_scan_cuda as ssc
    import vms_hip
    b, d, L = 8, 768, 272
    a, bb = list(_dir_inputs(b, d, L, torch.bfloat16, 50)), list(_dir_inputs(b, d, L, torch.bfloat16, 51))
    g = torch.Generator(device="cpu").manual_seed(52)
    for t in (a, bb):
        raw = torch.empty(b, d, L).uniform_(-30.0, 4.0, generator=g)
        raw[:, : d // 8] = torch.empty(b, d // 8, L).uniform_(-110.0, -80.0, generator=g)     # softplus underflows to 0 in fp32
        raw[:, d // 8: d // 4] = torch.empty(b, d // 8, L).uniform_(15.0, 30.0, generator=g)   # around the threshold
        t[1] = raw.to(torch.bfloat16).to(DEV)
        t[6] = torch.zeros(d, device=DEV)
        t[2] = (-0.02 * torch.rand(d, 16, generator=g) - 0.001).to(DEV)   # keep exp(delta A) away from 0 for delta ~ 30
    z = torch.randn(b, d, L, generator=g).to(torch.bfloat16).to(DEV)
    dout = torch.randn(b, d, L, generator=g).to(torch.bfloat16).to(DEV)
    fw = [ssc.fwd(*t[:6], z, t[6], True, reverse=(i == 1)) for i, t in enumerate((a, bb))]
    da, db = ssc.bwd_dual((*a[:6], a[6], fw[0][1], fw[0][0]), (*bb[:6], bb[6], fw[1][1], fw[1][0]), z, dout, torch.empty_like(z), True, keep_fp32=True)
    assert vms_hip.last_kernel() == "scan_bwd_pair4_dual_w4"
    f = lambda t: t.detach().float().cpu().numpy()
    for got, inp, rev in ((da, a, False), (db, bb, True)):
        u, delta, A, B, C, D, bias = [f(t) for t in inp]
        fl = (lambda t: np.ascontiguousarray(t[..., ::-1])) if rev else (lambda t: t)
        o = oracle.scan_bwd(fl(u), fl(delta), A, fl(B), fl(C), D, fl(f(z)), bias, fl(f(dout)), True, prec="f64")
        for k, name in enumerate(NAMES):
            ref = o[name][..., ::-1] if rev and name in ("du", "ddelta", "dB", "dC") else o[name]
            assert np.isfinite(f(got[k])).all(), name
            check(got[k], ref, tol_for(name, torch.bfloat16, "oracle"), f"{'b' if rev else 'a'}.{name} vs oracle (softplus extremes)")


def test_dual_rejects_mismatched_dz():
    import selective_scan_cuda as ssc
    import vms_hip
    if vms_hip.ext() is not None:
        pytest.skip("argument check of the C entry point: reached through the ctypes binding (VMS_DEBUG=no_torch_ext=1)")
    a, bb = _dir_inputs(1, 32, 64, torch.bfloat16, 0), _dir_inputs(1, 32, 64, torch.bfloat16, 1)
    z = torch.randn(1, 32, 64, device=DEV, dtype=torch.bfloat16)
    out, x, _ = ssc.fwd(*a[:6], z, a[6], True)
    ka, _ = ssc._bwd_prepare(*a[:6], z, a[6], z, x, out, None, True, False, False, None, True, False, 0, 0)
    kb, _ = ssc._bwd_prepare(*bb[:6], z, bb[6], z, x, out, None, True, False, True, None, True, False, 0, 0)   # b with its OWN dz
    with pytest.raises(RuntimeError, match="delivered in a->dz"):
        vms_hip.scan_bwd_dual(ka, kb)


@pytest.mark.parametrize("shape", [(8, 400, 768), (2, 272, 256)])   # (B, L, d_model): the dual grid / the two-launch fallback
def test_block_backward_dual_vs_two_launches(monkeypatch, shape):
    """The ViM block's one-node backward with the two scans as one call == with one launch per direction."""
    from mamba_ssm.modules.mamba_simple import Mamba
    from mamba_ssm.ops import selective_scan_interface as ssi
    b, L, dm = shape
    torch.manual_seed(0)
    block = Mamba(dm, d_state=16, expand=1, bimamba_type="v2").to(DEV)
    x = torch.randn(b, L, dm, device=DEV, dtype=torch.bfloat16, requires_grad=True)
    g = torch.randn(b, L, dm, device=DEV, dtype=torch.bfloat16)

    def step():
        block.zero_grad(set_to_none=True)
        x.grad = None
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = block(x)
        y.backward(g)
        return x.grad.clone(), {k: p.grad.clone() for k, p in block.named_parameters()}
    dx1, g1 = step()
    monkeypatch.setattr(ssi, "_DUAL_BWD", False)
    dx0, g0 = step()
    check(dx1, dx0, 2e-2, "block dx: dual vs two launches")
    for k in g0:
        check(g1[k], g0[k], 2e-2, f"block {k}: dual vs two launches")


def test_dual_with_groups(monkeypatch):
    """two B / C groups: a 16-row workgroup of the 4-wave grid never straddles a group ((dim / n_groups) % 32 == 0)"""
    import selective_scan_cuda as ssc
    import vms_hip
    b, d, L, G = 8, 768, 272, 2
    a, bb = list(_dir_inputs(b, d, L, torch.bfloat16, 30)), list(_dir_inputs(b, d, L, torch.bfloat16, 31))
    for t in (a, bb):
        t[3] = torch.randn(b, G, 16, L, device=DEV).to(torch.bfloat16)
        t[4] = torch.randn(b, G, 16, L, device=DEV).to(torch.bfloat16)
    z = torch.randn(b, d, L, device=DEV).to(torch.bfloat16)
    dout = torch.randn(b, d, L, device=DEV).to(torch.bfloat16)
    fw = [ssc.fwd(*t[:6], z, t[6], True, reverse=(i == 1)) for i, t in enumerate((a, bb))]
    dz_ref = torch.empty_like(z)
    ra = ssc.bwd(*a[:6], z, a[6], dout, fw[0][1], fw[0][0], dz_ref, True, False, keep_fp32=True)
    rb = ssc.bwd(*bb[:6], z, bb[6], dout, fw[1][1], fw[1][0], dz_ref, True, False, reverse=True, keep_fp32=True, accumulate_dz=True)
    da, db = ssc.bwd_dual((*a[:6], a[6], fw[0][1], fw[0][0]), (*bb[:6], bb[6], fw[1][1], fw[1][0]), z, dout, torch.empty_like(z), True, keep_fp32=True)
    assert vms_hip.last_kernel() == "scan_bwd_pair4_dual_w4"
    for got, ref in ((da, ra), (db, rb)):
        for k, name in enumerate(NAMES):
            tol = 0.0 if name in ("du", "ddelta") else 2e-5
            assert rel_err(got[k], ref[k]) <= tol, name
    check(da[7], ra[7], 2 ** -7, "dz")


def test_split_backward_mixed_directions_with_carry_sub_ranges(monkeypatch):
    """reverse_from + a sequence split whose carry pass runs over sub-ranges (long rows, few of them): == the two single-direction
    problems, each split on its own"""
    import selective_scan_cuda as ssc
    import vms_hip
    b, d, L = 2, 64, 32768
    u, delta, A, B, C, D, bias = _dir_inputs(b, d, L, torch.bfloat16, 40)
    z = torch.randn(b, d, L, device=DEV).to(torch.bfloat16)
    dout = torch.randn(b, d, L, device=DEV).to(torch.bfloat16)
    out, x, _ = ssc.fwd(u, delta, A, B, C, D, z, bias, True, reverse_from=1)
    g = ssc.bwd(u, delta, A, B, C, D, z, bias, dout, x, out, None, True, False, keep_fp32=True, reverse_from=1)
    assert vms_hip.last_kernel() == "scan_bwd_pair4+mixed+split", vms_hip.last_kernel()
    parts = []
    for i, rev in ((0, False), (1, True)):
        s = slice(i, i + 1)
        o, xx, _ = ssc.fwd(u[s], delta[s], A, B[s], C[s], D, z[s], bias, True, reverse=rev)
        parts.append(ssc.bwd(u[s], delta[s], A, B[s], C[s], D, z[s], bias, dout[s], xx, o, None, True, False, reverse=rev, keep_fp32=True))
    for k, name in enumerate(NAMES + ["dz"]):
        ref = torch.cat([parts[0][k], parts[1][k]]) if name in ("du", "ddelta", "dB", "dC", "dz") else parts[0][k] + parts[1][k]
        check(g[k], ref, 1e-2 if name in ("du", "ddelta", "dz") else 1e-3, f"mixed split {name}")
