"""The inner node's small projections on the matrix cores (csrc/inner_proj.hip, vms_hip.h vms_proj_apply / vms_proj_wgrad)
against fp32 / fp64 matrix products of the same 16-bit inputs (the reference computes them with torch GEMMs,
mamba_ssm/ops/selective_scan_interface.py:182, 275-279).  Tolerances: the product is accumulated in fp32 and rounded once
to the 16-bit output (proj_apply) or kept in fp32 (proj_wgrad), so the bar is the output format's rounding."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _vms():
    import vms_hip
    return vms_hip


# (batch, rows, k, seqlen): the block shapes of BASELINE configs (dt_proj: k = dt_rank; x_proj^T: k = dt_rank + 2 d_state),
# ragged tiles (seqlen % 64, rows % 128, k % 16 != 0), one-tile and sub-tile problems
APPLY_SHAPES = [(2, 256, 64, 1024), (8, 1024, 96, 8192), (2, 768, 48, 3136), (2, 768, 80, 3136), (2, 512, 32, 2304),
                (1, 384, 24, 3152), (1, 384, 56, 3152), (3, 200, 17, 72), (2, 96, 96, 8), (1, 130, 5, 200)]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("shape", APPLY_SHAPES)
@pytest.mark.parametrize("accumulate", [False, True])
def test_proj_apply_vs_matmul(shape, dtype, accumulate):
    vms = _vms()
    b, rows, k, L = shape
    torch.manual_seed(rows + k)
    w = (torch.randn(rows, k, device=DEV) * 0.2).to(dtype)
    x = torch.randn(b, k, L, device=DEV).to(dtype)
    out = torch.randn(b, rows, L, device=DEV).to(dtype)
    want = w.double() @ x.double()
    if accumulate:
        want = want + out.double()
    vms.proj_apply(w, x, out, accumulate)
    assert vms.lib().vms_last_kernel().decode().startswith("proj_apply")
    err = (out.double() - want).abs().max().item()
    eps = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    assert err <= eps * want.abs().max().item() * 1.01 + 1e-6, f"max abs err {err:.3e}"


def test_proj_apply_strided_views():
    """in = the first rows of a wider (batch, R + 2N, L) tensor, w = the transposed view of x_proj's weight, out = one half of a
    (batch, 2 d, L) buffer: exactly the views the inner node passes."""
    vms = _vms()
    torch.manual_seed(0)
    b, d, R, N, L = 2, 256, 48, 16, 640
    x_dbl = torch.randn(b, R + 2 * N, L, device=DEV).bfloat16()
    w_dt = (torch.randn(d, R, device=DEV) * 0.2).bfloat16()
    wide = torch.zeros(b, 2 * d, L, device=DEV, dtype=torch.bfloat16)
    out = wide[:, d:, :]
    vms.proj_apply(w_dt, x_dbl[:, :R, :], out, False)
    want = w_dt.double() @ x_dbl[:, :R, :].double()
    assert (out.double() - want).abs().max().item() <= 2.0 ** -8 * want.abs().max().item() * 1.01
    assert wide[:, :d, :].abs().max().item() == 0.0
    w_x = (torch.randn(R + 2 * N, d, device=DEV) * 0.2).bfloat16()     # x_proj.weight (R + 2N, d): w = its transpose, a view
    g = torch.randn(b, d, L, device=DEV).bfloat16()
    want = g.double() + w_x.t().double() @ x_dbl.double()
    vms.proj_apply(w_x.t(), x_dbl, g, True)
    assert (g.double() - want).abs().max().item() <= 2.0 ** -8 * want.abs().max().item() * 1.01


WGRAD_SHAPES = [(2, 64, 256, 1024), (8, 96, 1024, 8192), (8, 64, 1024, 8192), (2, 48, 768, 3136), (2, 80, 768, 3136),
                (2, 64, 512, 2304), (1, 24, 384, 3152), (1, 56, 384, 3152), (3, 17, 200, 72), (2, 128, 130, 136), (1, 5, 96, 8)]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("shape", WGRAD_SHAPES)
def test_proj_wgrad_vs_matmul(shape, dtype):
    vms = _vms()
    b, m, n, L = shape
    torch.manual_seed(m + n)
    p = torch.randn(b, m, L, device=DEV).to(dtype)
    q = torch.randn(b, n, L, device=DEV).to(dtype)
    dw = torch.zeros(m, n, device=DEV)
    vms.proj_wgrad(p, q, dw)
    assert vms.lib().vms_last_kernel().decode() == "proj_wgrad"
    want = torch.einsum("bml,bnl->mn", p.double(), q.double())
    scale = (p.double().abs().unsqueeze(2) * q.double().abs().unsqueeze(1)).sum(dim=(0, 3)).max().item() if b * m * n * L < 2e8 else want.abs().max().item() * 30
    err = (dw.double() - want).abs().max().item()
    assert err <= 1e-5 * scale + 1e-6, f"max abs err {err:.3e} (sum of |products| {scale:.3e})"
    vms.proj_wgrad(p, q, dw)      # the buffer is added to: a second call doubles it
    assert (dw.double() - 2 * want).abs().max().item() <= 2e-5 * scale + 1e-6


@pytest.mark.parametrize("shape", [(2, 48, 768, 3136), (8, 64, 1024, 1024), (1, 24, 384, 3152), (3, 17, 200, 72), (2, 128, 130, 136)])
def test_proj_wgrad_transposed(shape):
    """dw stored (n, m) -- ddt_proj.weight in the parameter's own (d_inner, dt_rank) layout -- == the (m, n) call, transposed"""
    vms = _vms()
    b, m, n, L = shape
    torch.manual_seed(m * n)
    p = torch.randn(b, m, L, device=DEV).bfloat16()
    q = torch.randn(b, n, L, device=DEV).bfloat16()
    dw, dwt = torch.zeros(m, n, device=DEV), torch.zeros(n, m, device=DEV)
    vms.proj_wgrad(p, q, dw)
    vms.proj_wgrad(p, q, dwt, transposed=True)
    want = torch.einsum("bml,bnl->mn", p.double(), q.double())
    scale = want.abs().max().item()
    assert (dwt.t().double() - want).abs().max().item() <= 1e-4 * scale + 1e-5
    assert (dwt.t() - dw).abs().max().item() <= 1e-4 * scale + 1e-5     # (the atomics' order differs)


def test_proj_wgrad_strided_rows():
    vms = _vms()
    torch.manual_seed(1)
    b, R, N, d, L = 2, 48, 16, 256, 640
    x_dbl = torch.randn(b, R + 2 * N, L, device=DEV).bfloat16()
    xz = torch.randn(b, 2 * d, L, device=DEV).bfloat16()
    dw = torch.zeros(R, d, device=DEV)
    vms.proj_wgrad(x_dbl[:, :R, :], xz[:, d:, :], dw)
    want = torch.einsum("bml,bnl->mn", x_dbl[:, :R, :].double(), xz[:, d:, :].double())
    assert (dw.double() - want).abs().max().item() <= 1e-4 * want.abs().max().item()


def test_proj_checks():
    vms = _vms()
    w = torch.randn(64, 16, device=DEV).bfloat16()
    x = torch.randn(1, 16, 64, device=DEV).bfloat16()
    with pytest.raises(RuntimeError):   # fp32 is left to the library GEMM
        vms.proj_apply(w.float(), x.float(), torch.empty(1, 64, 64, device=DEV), False)
    with pytest.raises(RuntimeError):   # seqlen % 8
        vms.proj_apply(w, x[:, :, :60].contiguous(), torch.empty(1, 64, 60, device=DEV, dtype=torch.bfloat16), False)
    with pytest.raises(RuntimeError):   # k > 96
        vms.proj_apply(torch.randn(64, 112, device=DEV).bfloat16(), torch.randn(1, 112, 64, device=DEV).bfloat16(),
                       torch.empty(1, 64, 64, device=DEV, dtype=torch.bfloat16), False)


# (batch, m, k, seqlen): x_proj (m = dt_rank + 2 d_state, k = d_inner) and dt_proj^T (m = dt_rank) of the BASELINE configs; ragged tiles
# (seqlen % 64, k % 64, m % 16 != 0), sub-tile problems
KRED_SHAPES = [(8, 96, 1024, 8192), (8, 64, 1024, 8192), (2, 80, 768, 3136), (2, 48, 768, 3136), (2, 64, 512, 2304), (1, 56, 384, 3152),
               (1, 24, 384, 3152), (1, 80, 768, 8192), (3, 17, 200, 72), (2, 96, 40, 8), (1, 5, 136, 200), (2, 33, 64, 264)]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("shape", KRED_SHAPES)
@pytest.mark.parametrize("tile", [0, 64, 128, 256])
def test_proj_kred_vs_matmul(shape, dtype, tile):
    """out[b, m, l] = sum_k w[m, k] in[b, k, l] with w stored (m, k) (x_proj.weight): every tile width, fp32 accumulation, one rounding."""
    vms = _vms()
    b, m, k, L = shape
    torch.manual_seed(m + k)
    w = (torch.randn(m, k, device=DEV) * k ** -0.5).to(dtype)
    x = torch.randn(b, k, L, device=DEV).to(dtype)
    out = torch.full((b, m, L), float("nan"), device=DEV, dtype=dtype)
    vms.proj_kred(w, x, out, tile=tile)
    assert vms.lib().vms_last_kernel().decode() == "proj_kred"
    want = w.double() @ x.double()
    err = (out.double() - want).abs().max().item()
    eps = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    assert err <= eps * want.abs().max().item() * 1.01 + 1e-6, f"max abs err {err:.3e}"


@pytest.mark.parametrize("shape", [(8, 64, 1024, 8192), (2, 48, 768, 3136), (2, 32, 512, 2304), (1, 24, 384, 3152), (3, 8, 200, 72), (2, 96, 136, 264)])
@pytest.mark.parametrize("tile", [0, 64, 256])
def test_proj_kred_transposed_weight(shape, tile):
    """The weight stored (k, m) -- dt_proj.weight (d_inner, dt_rank) used as its transpose -- and the views the backward passes: in =
    ddelta in the scan's channel-slowest layout, out = the first rows of dx_dbl."""
    vms = _vms()
    b, m, k, L = shape
    torch.manual_seed(m * k)
    w_dt = (torch.randn(k, m, device=DEV) * k ** -0.5).bfloat16()          # (d_inner, dt_rank)
    ddelta = torch.randn(k, b, L, device=DEV).bfloat16().permute(1, 0, 2)   # strides (L, b L, 1)
    dx_dbl = torch.zeros(b, m + 32, L, device=DEV, dtype=torch.bfloat16)
    vms.proj_kred(w_dt.t(), ddelta, dx_dbl[:, :m, :], tile=tile)
    assert vms.lib().vms_last_kernel().decode() == "proj_kred_t"
    want = w_dt.t().double() @ ddelta.double()
    assert (dx_dbl[:, :m, :].double() - want).abs().max().item() <= 2.0 ** -8 * want.abs().max().item() * 1.01 + 1e-6
    assert dx_dbl[:, m:, :].abs().max().item() == 0.0


@pytest.mark.parametrize("shape", [(8, 96, 1024, 8192), (2, 80, 768, 3136), (1, 56, 384, 136)])
def test_proj_kred_dual(shape):
    """Two problems of one shape in one launch == the two single launches, bit for bit."""
    vms = _vms()
    b, m, k, L = shape
    torch.manual_seed(7)
    ws = [(torch.randn(m, k, device=DEV) * k ** -0.5).bfloat16() for _ in range(2)]
    xs = [torch.randn(b, k, L, device=DEV).bfloat16() for _ in range(2)]
    single = [torch.empty(b, m, L, device=DEV, dtype=torch.bfloat16) for _ in range(2)]
    for w, x, o in zip(ws, xs, single):
        vms.proj_kred(w, x, o)
    dual = [torch.empty(b, m, L, device=DEV, dtype=torch.bfloat16) for _ in range(2)]
    vms.proj_kred(ws[0], xs[0], dual[0], ws[1], xs[1], dual[1])
    assert vms.lib().vms_last_kernel().decode() == "proj_kred+dual"
    for o, d2 in zip(single, dual):
        assert torch.equal(o, d2)
    ext = vms.ext()
    if ext is not None:
        xa, xb = ext.x_proj_dual(ws[0], xs[0], ws[1], xs[1], True)
        assert torch.equal(xa, single[0]) and torch.equal(xb, single[1])
        la, lb = ext.x_proj_dual(ws[0], xs[0], ws[1], xs[1], False)      # the library's GEMMs: same values to the output format's rounding
        assert (la.float() - xa.float()).abs().max().item() <= 2.0 ** -7 * xa.float().abs().max().item()


@pytest.mark.parametrize("shape", [(8, 64, 1024, 1024, 16), (2, 48, 768, 3136, 16), (1, 24, 384, 3152, 16), (3, 8, 200, 72, 4), (2, 32, 512, 264, 64)])
@pytest.mark.parametrize("tile", [0, 64, 256])
def test_proj_kred_cast_rows(shape, tile):
    """The backward call's side job: the scan's fp32 dB / dC sums rounded into the rows of dx_dbl behind d_dt (both bindings)."""
    vms = _vms()
    b, R, d, L, N = shape
    torch.manual_seed(R + d)
    w_dt = (torch.randn(d, R, device=DEV) * d ** -0.5).bfloat16()
    ddelta = torch.randn(d, b, L, device=DEV).bfloat16().permute(1, 0, 2)
    dbc = torch.randn(2, b, N, L, device=DEV)                               # dB then dC, as the scan's backward leaves them
    want_dt = (w_dt.t().double() @ ddelta.double())
    for how in ("ctypes", "ext"):
        dx_dbl = torch.full((b, R + 2 * N + 8, L), 7.0, device=DEV, dtype=torch.bfloat16)
        if how == "ctypes":
            vms.proj_kred(w_dt.t(), ddelta, dx_dbl[:, :R, :], tile=tile, cast_src=dbc)
        else:
            if vms.ext() is None:
                continue
            vms.ext().proj_kred(w_dt.t(), ddelta, dx_dbl[:, :R, :], tile=tile, cast_src=dbc)
        assert (dx_dbl[:, :R, :].double() - want_dt).abs().max().item() <= 2.0 ** -8 * want_dt.abs().max().item() * 1.01 + 1e-6
        assert torch.equal(dx_dbl[:, R:R + N, :], dbc[0].bfloat16()) and torch.equal(dx_dbl[:, R + N:R + 2 * N, :], dbc[1].bfloat16())
        assert (dx_dbl[:, R + 2 * N:, :] == 7.0).all()


# (batch, dim, seqlen, m, width)
CXP_SHAPES = [(8, 1024, 8192, 96, 4), (2, 768, 3136, 80, 4), (1, 384, 3152, 56, 4), (2, 200, 72, 17, 3), (3, 64, 8, 33, 2), (1, 136, 264, 96, 4)]


@pytest.mark.parametrize("dtype,wdtype,has_bias", [(torch.bfloat16, torch.float32, True), (torch.bfloat16, torch.bfloat16, True),
                                                    (torch.float16, torch.float32, False)])
@pytest.mark.parametrize("shape", CXP_SHAPES)
@pytest.mark.parametrize("tile", [0, 64, 128])
def test_conv_xproj_dual_vs_separate(shape, dtype, wdtype, has_bias, tile):
    """vms_conv_xproj_dual == vms_causal_conv1d_fwd_dual followed by vms_proj_kred (both directions), bit for bit: the same conv
    arithmetic tap for tap, the same order of matrix-core steps per output."""
    vms = _vms()
    b, d, L, m, width = shape
    torch.manual_seed(d + L)
    xz = torch.randn(b, 2 * d, L, device=DEV).to(dtype)
    x = xz[:, :d, :]                                        # the view the block passes (batch stride 2 d L)
    cw, cwb = [(torch.randn(d, width, device=DEV) * 0.4).to(wdtype) for _ in range(2)]
    cb, cbb = [(torch.randn(d, device=DEV) * 0.2).to(wdtype) if has_bias else None for _ in range(2)]
    wx, wxb = [(torch.randn(m, d, device=DEV) * d ** -0.5).to(dtype) for _ in range(2)]
    ref_o, ref_ob = torch.empty(b, d, L, device=DEV, dtype=dtype), torch.empty(b, d, L, device=DEV, dtype=dtype)
    vms.conv_fwd_dual(x, cw, cb, ref_o, cwb, cbb, ref_ob, True)
    ref_x, ref_xb = torch.empty(b, m, L, device=DEV, dtype=dtype), torch.empty(b, m, L, device=DEV, dtype=dtype)
    vms.proj_kred(wx, ref_o, ref_x, wxb, ref_ob, ref_xb)
    o, ob = torch.full_like(ref_o, float("nan")), torch.full_like(ref_ob, float("nan"))
    xd, xdb = torch.full_like(ref_x, float("nan")), torch.full_like(ref_xb, float("nan"))
    assert vms.conv_xproj_dual_eligible(x, cw, cb, cwb, cbb, wx, wxb)
    vms.conv_xproj_dual(x, cw, cb, o, cwb, cbb, ob, wx, wxb, xd, xdb, tile=tile)
    assert vms.lib().vms_last_kernel().decode().startswith("conv_xproj_dual")
    assert torch.equal(o, ref_o) and torch.equal(ob, ref_ob), "conv1d outputs differ from vms_causal_conv1d_fwd_dual"
    assert torch.equal(xd, ref_x) and torch.equal(xdb, ref_xb), "x_dbl differs from vms_proj_kred on the same conv1d outputs"
    ext = vms.ext()
    if ext is not None and tile == 0:
        r = ext.conv_xproj_dual(x, cw, cb, cwb, cbb, wx, wxb)
        assert len(r) == 4 and all(torch.equal(a, w) for a, w in zip(r, (ref_o, ref_ob, ref_x, ref_xb)))
        # what the kernel declines comes back empty (fp32 activations, m > 96)
        assert len(ext.conv_xproj_dual(x.float(), cw.float(), None, cwb.float(), None, wx.float(), wxb.float())) == 0


def test_bidirectional_node_fused_head_vs_separate(monkeypatch):
    """The bidirectional node with the fused head (conv1d + x_proj of both directions in one launch) == the node on the separate
    launches: outputs and all gradients identical (same values feed the same kernels)."""
    import vms_hip
    import mamba_ssm.ops.selective_scan_interface as ssi
    if vms_hip.ext() is None:
        pytest.skip("compiled binding not built")
    b, L, d, N = 2, 640, 256, 16
    R = d // 16
    torch.manual_seed(11)
    mk = lambda *s, sc=1.0: (torch.randn(*s, device=DEV) * sc).requires_grad_()
    xz0 = torch.randn(b, 2 * d, L, device=DEV).bfloat16()
    sets = [(mk(d, 1, 4, sc=0.3), mk(d, sc=0.1), mk(R + 2 * N, d, sc=d ** -0.5), mk(d, R, sc=R ** -0.5),
             (-torch.rand(d, N, device=DEV) - 0.2).requires_grad_(), mk(d), mk(d, sc=0.3)) for _ in range(2)]
    g = torch.randn(b, d, L, device=DEV)

    def run(fused):
        monkeypatch.setattr(ssi, "_CONV_XPROJ", fused)
        xz = xz0.clone().requires_grad_()
        for st in sets:
            for t in st:
                t.grad = None
        with torch.autocast("cuda", dtype=torch.bfloat16):
            out = ssi.bimamba_inner_fn_no_out_proj(xz, sets[0], sets[1], checkpoint_lvl=0)
        out.backward(g.to(out.dtype))
        return [out.detach(), xz.grad] + [t.grad.clone() for st in sets for t in st]

    got, want = run(True), run(False)
    for i, (a, w) in enumerate(zip(got, want)):
        err = (a.float() - w.float()).abs().max().item() / max(w.float().abs().max().item(), 1e-6)
        assert err <= 1e-3, f"tensor {i}: rel err {err:.3e}"     # identical forward values; atomics' order in the batch sums


@pytest.mark.parametrize("shape", [(8, 2048, 1024), (14, 1536, 768), (16, 768, 768), (2, 96, 8), (1, 64, 64), (3, 5, 7)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("odt", [torch.float32, torch.bfloat16])
def test_sum_slices(shape, dtype, odt):
    """vms_sum_slices (the sum over the K slices of the large weight-gradient GEMMs) vs a float64 sum; shapes it declines go to torch"""
    vms = _vms()
    torch.manual_seed(shape[0])
    t = torch.randn(*shape, device=DEV).to(dtype)
    got = vms.sum_slices(t, odt)
    assert got.dtype == odt and tuple(got.shape) == shape[1:]
    want = t.double().sum(0)
    tol = (2.0 ** -8 if odt == torch.bfloat16 else 2e-6) * max(want.abs().max().item(), 1.0) * (1 if odt == torch.bfloat16 else shape[0])
    assert (got.double() - want).abs().max().item() <= tol
    if (shape[1] * shape[2]) % 8 == 0 and shape[0] > 0:
        assert vms.lib().vms_last_kernel().decode() == "sum_slices"


def test_proj_kred_and_fused_head_random_shapes():
    """Random small problems (every tile tail, channel-block tail and row-count class): vms_proj_kred vs float64 products, the fused head
    vs the two launches it replaces, bit for bit."""
    import random
    vms = _vms()
    rnd = random.Random(20250930)
    for it in range(24):
        b, d, L = rnd.randint(1, 3), 8 * rnd.randint(1, 40), 8 * rnd.randint(1, 60)
        m, width = rnd.randint(1, 96), rnd.randint(2, 4)
        torch.manual_seed(it)
        x = torch.randn(b, d, L, device=DEV).bfloat16()
        w = (torch.randn(m, d, device=DEV) * d ** -0.5).bfloat16()
        out = torch.full((b, m, L), float("nan"), device=DEV, dtype=torch.bfloat16)
        tile = rnd.choice([0, 64, 128, 256])
        vms.proj_kred(w, x, out, tile=tile)
        want = w.double() @ x.double()
        assert (out.double() - want).abs().max().item() <= 2.0 ** -8 * want.abs().max().item() * 1.01 + 1e-6, (b, d, L, m, tile)
        if m % 8 == 0:      # the transposed weight form needs whole 16-byte pieces along m
            wt = w.t().contiguous()                                        # (d, m): read as its transpose
            out2 = torch.full_like(out, float("nan"))
            vms.proj_kred(wt.t(), x, out2, tile=tile)
            assert torch.equal(out2, out), (b, d, L, m, tile, "transposed weight")
        cw, cwb = torch.randn(d, width, device=DEV) * 0.4, torch.randn(d, width, device=DEV) * 0.4
        cb, cbb = (torch.randn(d, device=DEV) * 0.2, torch.randn(d, device=DEV) * 0.2) if it % 2 else (None, None)
        wb = (torch.randn(m, d, device=DEV) * d ** -0.5).bfloat16()
        ro, rob = torch.empty_like(x), torch.empty_like(x)
        vms.conv_fwd_dual(x, cw, cb, ro, cwb, cbb, rob, True)
        rx, rxb = torch.empty_like(out), torch.empty_like(out)
        vms.proj_kred(w, ro, rx, wb, rob, rxb)
        o, ob, xd, xdb = torch.empty_like(x), torch.empty_like(x), torch.empty_like(out), torch.empty_like(out)
        vms.conv_xproj_dual(x, cw, cb, o, cwb, cbb, ob, w, wb, xd, xdb, tile=rnd.choice([0, 64, 128]))
        assert torch.equal(o, ro) and torch.equal(ob, rob) and torch.equal(xd, rx) and torch.equal(xdb, rxb), (b, d, L, m, width)


def test_proj_kred_checks():
    vms = _vms()
    w = torch.randn(48, 128, device=DEV).bfloat16()
    x = torch.randn(1, 128, 64, device=DEV).bfloat16()
    assert vms.proj_kred_eligible(w, x, torch.empty(1, 48, 64, device=DEV, dtype=torch.bfloat16))
    with pytest.raises(RuntimeError):   # fp32 is left to the library GEMM
        vms.proj_kred(w.float(), x.float(), torch.empty(1, 48, 64, device=DEV))
    with pytest.raises(RuntimeError):   # m > 96
        vms.proj_kred(torch.randn(112, 128, device=DEV).bfloat16(), x, torch.empty(1, 112, 64, device=DEV, dtype=torch.bfloat16))
    with pytest.raises(RuntimeError):   # seqlen % 8
        vms.proj_kred(w, x[:, :, :60].contiguous(), torch.empty(1, 48, 60, device=DEV, dtype=torch.bfloat16))
    with pytest.raises(RuntimeError):   # a weight view with no unit stride
        vms.proj_kred(torch.randn(48, 256, device=DEV).bfloat16()[:, ::2], x, torch.empty(1, 48, 64, device=DEV, dtype=torch.bfloat16))
    P = vms.ProjKredParams()            # the library's own checks (status code + message)
    assert vms.lib().vms_proj_kred(None, None) != 0
    P.batch, P.m, P.k, P.seqlen, P.dtype = 1, 200, 128, 64, vms.dtype_code(x)
    P.w, P.inp, P.out = w.data_ptr(), x.data_ptr(), x.data_ptr()
    import ctypes
    assert vms.lib().vms_proj_kred(ctypes.byref(P), None) != 0 and b"m <= 96" in vms.lib().vms_last_error()


@pytest.mark.parametrize("shape", [(2, 640, 256, 16), (2, 1024, 96, 8), (1, 2304, 512, 16), (2, 200, 768, 16)])
@pytest.mark.parametrize("variant", ["fused_tail", "mfma_proj", "both"])
@pytest.mark.parametrize("reverse", [False, True])
def test_inner_node_variants_vs_library(shape, variant, reverse, monkeypatch):
    """The fused inner node (conv -> x_proj -> dt_proj -> scan, and its backward) with its small GEMMs on the hand-written
    kernels -- the default one-pass backward tail (vms_proj_conv_bwd), the opt-in one-for-one MFMA projections
    (vms_hip.debug.mfma_proj), and both -- vs the same node on library GEMMs + vms_causal_conv1d_bwd (debug.no_fused_tail):
    outputs and every gradient within the bf16 bar."""
    import vms_hip
    from mamba_ssm.ops.selective_scan_interface import mamba_inner_fn_no_out_proj
    if vms_hip.ext() is None:
        pytest.skip("compiled binding not built")
    b, L, d, N = shape
    R = (d + 15) // 16
    torch.manual_seed(L)
    mk = lambda *s, sc=1.0: (torch.randn(*s, device=DEV) * sc).requires_grad_()
    xz0 = torch.randn(b, 2 * d, L, device=DEV).bfloat16()   # what in_proj hands the node under autocast
    conv_w, conv_b = mk(d, 1, 4, sc=0.3), mk(d, sc=0.1)
    x_proj_w, dt_proj_w = mk(R + 2 * N, d, sc=d ** -0.5), mk(d, R, sc=R ** -0.5)
    A = (-torch.rand(d, N, device=DEV) - 0.2).requires_grad_()
    D, bias = mk(d), mk(d, sc=0.3)
    g = torch.randn(b, d, L, device=DEV)
    params = (conv_w, conv_b, x_proj_w, dt_proj_w, A, D, bias)

    def run(env):
        import mamba_ssm.ops.selective_scan_interface as ssi
        import vms_hip
        monkeypatch.setattr(ssi, "_PROJ_KRED", "NO_KRED" not in env)   # the library-GEMM node also runs x_proj / dt_proj^T on the library
        monkeypatch.setattr(vms_hip.debug, "mfma_proj", True if "VMS_MFMA_PROJ" in env else None)
        monkeypatch.setattr(vms_hip.debug, "no_fused_tail", "VMS_NO_FUSED_TAIL" in env)
        xz = xz0.clone().requires_grad_()
        for t in params:
            t.grad = None
        with torch.autocast("cuda", dtype=torch.bfloat16):
            out = mamba_inner_fn_no_out_proj(xz, conv_w, conv_b, x_proj_w, dt_proj_w, A, None, None, D, delta_bias=bias,
                                             delta_softplus=True, reverse=reverse)
        out.backward(g.to(out.dtype))
        return [out.detach().float(), xz.grad.float()] + [t.grad.float().clone() for t in params]

    # (which kernels ran is not visible here: vms_last_kernel() is thread-local and autograd runs the backward on its own thread;
    # the C-ABI level tests above assert the kernel names)
    env = {"fused_tail": (), "mfma_proj": ("VMS_MFMA_PROJ", "VMS_NO_FUSED_TAIL"), "both": ("VMS_MFMA_PROJ",)}[variant]
    got = run(env)
    want = run(("VMS_NO_FUSED_TAIL", "NO_KRED"))
    names = ["out", "dxz", "dconv_w", "dconv_b", "dx_proj_w", "ddt_proj_w", "dA", "dD", "dbias"]
    for n, a, w in zip(names, got, want):
        err = (a - w).abs().max().item() / max(w.abs().max().item(), 1e-6)
        assert err <= 2e-2, f"{n}: rel err {err:.3e} between the {variant} node and the library-GEMM node"


# (batch, dim, k, seqlen, width)
CONV_BWD_SHAPES = [(2, 256, 96, 1024, 4), (2, 128, 64, 512, 4), (1, 200, 80, 328, 3), (3, 96, 56, 72, 2), (2, 130, 48, 8, 4),
                   (1, 768, 80, 3136, 4), (2, 64, 36, 1152, 4),
                   # ragged lengths (seqlen % 8 != 0: odd row strides, a partly valid last piece per row): 8 x 196 + 1, short rows
                   (2, 192, 80, 1569, 4), (2, 100, 96, 77, 3), (3, 64, 48, 17, 4), (1, 96, 64, 7, 4), (2, 72, 40, 131, 2)]


def _conv_tail_reference(x, du, dx_dbl, w_x, conv_w, conv_b, reverse_rows):
    """fp64 statement of selective_scan_interface.py:278-283 on the values the kernel sees; reverse_rows: bool per batch entry."""
    xd = x.double()
    flip = lambda t, b: t.flip(-1) if reverse_rows[b] else t
    outs = []
    dW = torch.zeros(w_x.shape, dtype=torch.float64, device=x.device)
    dcw = torch.zeros(conv_w.shape, dtype=torch.float64, device=x.device)
    dcb = torch.zeros(conv_w.shape[0], dtype=torch.float64, device=x.device)
    W = conv_w.shape[1]
    for b in range(x.shape[0]):
        xb = flip(xd[b], b).clone().requires_grad_()
        cw = conv_w.double().clone().requires_grad_()
        cb = (conv_b.double() if conv_b is not None else torch.zeros(conv_w.shape[0], dtype=torch.float64, device=x.device)).clone().requires_grad_()
        pre = torch.nn.functional.conv1d(torch.nn.functional.pad(xb, (W - 1, 0)).unsqueeze(0), cw.unsqueeze(1), cb, groups=xb.shape[0])[0]
        conv_out = pre * torch.sigmoid(pre)
        # the kernel multiplies the ROUNDED conv1d_out (what the forward stored) into dW_x
        co_r = conv_out.detach().to(x.dtype).double()
        g = flip(du[b].double(), b) + w_x.double().t() @ flip(dx_dbl[b].double(), b)
        dW += flip(dx_dbl[b].double(), b) @ co_r.t()
        conv_out.backward(g)
        outs.append(flip(xb.grad, b))
        dcw += cw.grad
        dcb += cb.grad
    return torch.stack(outs), dcw, dcb, dW


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("shape", CONV_BWD_SHAPES)
@pytest.mark.parametrize("mode", ["fwd", "rev", "mixed", "acc"])
def test_proj_conv_bwd_vs_reference(shape, dtype, mode):
    vms = _vms()
    b, d, k, L, W = shape
    if mode == "mixed" and b < 2:
        pytest.skip("reverse_from needs two batch entries")
    torch.manual_seed(d + k + L)
    xz = torch.randn(b, 2 * d, L, device=DEV).to(dtype)
    x = xz[:, :d, :]                                              # the conv's input is a channel half of xz
    du = torch.randn(b, d, L, device=DEV).to(dtype)
    dx_dbl = (torch.randn(b, k, L, device=DEV) * 0.5).to(dtype)
    w_x = (torch.randn(k, d, device=DEV) * d ** -0.5).to(dtype)
    conv_w, conv_b = torch.randn(d, W, device=DEV) * 0.4, torch.randn(d, device=DEV) * 0.2
    dxz = torch.randn(b, 2 * d, L, device=DEV).to(dtype)
    dx = dxz[:, :d, :]
    old = dx.clone()
    dcw, dcb, dW = torch.zeros(d, W, device=DEV), torch.zeros(d, device=DEV), torch.zeros(k, d, device=DEV)
    rev_rows = [mode == "rev" or (mode == "mixed" and i >= 1) for i in range(b)]
    vms.proj_conv_bwd(x, du, dx_dbl, w_x, conv_w, conv_b, dx, dcw, dcb, dW, reverse=mode == "rev",
                      reverse_from=1 if mode == "mixed" else 0, dx_accumulate=mode == "acc")
    assert vms.lib().vms_last_kernel().decode() == ("proj_conv_bwd" if L % 8 == 0 else "proj_conv_bwd_ragged")
    r_dx, r_dcw, r_dcb, r_dW = _conv_tail_reference(x, du, dx_dbl, w_x, conv_w, conv_b, rev_rows)
    if mode == "acc":
        r_dx = r_dx + old.double()
    eps = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    rel = lambda a, r: (a.double() - r).abs().max().item() / max(r.abs().max().item(), 1e-9)
    assert rel(dx, r_dx) <= eps * 1.5, f"dx rel err {rel(dx, r_dx):.3e}"
    assert (dxz[:, d:, :] != dxz[:, d:, :]).sum().item() == 0    # the z half is untouched (no NaN scribbles)
    assert rel(dcw, r_dcw) <= 5e-4, f"dconv_w rel err {rel(dcw, r_dcw):.3e}"
    assert rel(dcb, r_dcb) <= 5e-4, f"dconv_b rel err {rel(dcb, r_dcb):.3e}"
    # conv1d_out enters dW_x rounded to the 16-bit dtype; the kernel's sigmoid (v_exp / v_rcp) and torch's differ in the last
    # fp32 bit, which flips the rounding of a few elements: ~1e-4 of the sum
    assert rel(dW, r_dW) <= 1e-3, f"dW_x rel err {rel(dW, r_dW):.3e}"


def test_proj_conv_bwd_random_shapes():
    """40 seeded random problems (batch 1-3, dim 8-300, k 33-96, seqlen up to 1,000 -- half of them not a multiple of 8 --, width 2-4, every direction
    mode, +-bias, +-dx_accumulate, strided channel-half views) against the fp64 statement: tile edges, channel tails, the carry
    across tiles and across workgroup ranges (tiles_per_wg forced small)."""
    vms = _vms()
    import random
    rng = random.Random(1234)
    for case in range(40):
        b, d = rng.randint(1, 3), rng.randint(8, 300)
        k, L, W = rng.randint(33, 96), (8 * rng.randint(1, 125) if rng.random() < 0.5 else rng.randint(1, 1000)), rng.randint(2, 4)
        mode = rng.choice(["fwd", "rev", "mixed"]) if b > 1 else rng.choice(["fwd", "rev"])
        acc, has_bias, dtype = rng.random() < 0.5, rng.random() < 0.7, rng.choice([torch.bfloat16, torch.float16])
        tpw = rng.choice([0, 1, 2, 3])
        torch.manual_seed(case)
        xz = torch.randn(b, 2 * d, L, device=DEV).to(dtype)
        x, du = xz[:, :d, :], torch.randn(b, d, L, device=DEV).to(dtype)
        dx_dbl = (torch.randn(b, k, L, device=DEV) * 0.5).to(dtype)
        w_x = (torch.randn(k, d, device=DEV) * d ** -0.5).to(dtype)
        conv_w = torch.randn(d, W, device=DEV) * 0.4
        conv_b = torch.randn(d, device=DEV) * 0.2 if has_bias else None
        dxz = torch.randn(b, 2 * d, L, device=DEV).to(dtype)
        dx = dxz[:, :d, :]
        old, z_before = dx.clone(), dxz[:, d:, :].clone()
        dcw, dW = torch.zeros(d, W, device=DEV), torch.zeros(k, d, device=DEV)
        dcb = torch.zeros(d, device=DEV) if has_bias else None
        rf = rng.randint(1, b - 1) if mode == "mixed" else 0
        rev_rows = [mode == "rev" or (mode == "mixed" and i >= rf) for i in range(b)]
        vms.proj_conv_bwd(x, du, dx_dbl, w_x, conv_w, conv_b, dx, dcw, dcb, dW, reverse=mode == "rev", reverse_from=rf,
                          dx_accumulate=acc, tiles_per_wg=tpw)
        r_dx, r_dcw, r_dcb, r_dW = _conv_tail_reference(x, du, dx_dbl, w_x, conv_w, conv_b, rev_rows)
        if acc:
            r_dx = r_dx + old.double()
        eps = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
        rel = lambda a, r: (a.double() - r).abs().max().item() / max(r.abs().max().item(), 1e-9)
        tag = f"case {case}: b={b} d={d} k={k} L={L} W={W} {mode} rf={rf} acc={acc} bias={has_bias} {dtype} tpw={tpw}"
        assert rel(dx, r_dx) <= eps * 1.5, f"{tag}: dx rel err {rel(dx, r_dx):.3e}"
        assert torch.equal(dxz[:, d:, :], z_before), f"{tag}: the z half of dxz was touched"
        assert rel(dcw, r_dcw) <= 1e-3, f"{tag}: dconv_w rel err {rel(dcw, r_dcw):.3e}"
        if has_bias:
            assert rel(dcb, r_dcb) <= 1e-3, f"{tag}: dconv_b rel err {rel(dcb, r_dcb):.3e}"
        assert rel(dW, r_dW) <= 2e-3, f"{tag}: dW_x rel err {rel(dW, r_dW):.3e}"
