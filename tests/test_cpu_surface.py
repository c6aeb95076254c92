"""CPU-only checks of the host side: the C-ABI library loads and exports what include/vms_hip.h
declares, the Python surface mirrors the reference's names / state-dict keys, the pure-PyTorch
*_ref functions shipped in the product agree with the golden vectors, and the fused autograd
nodes' host logic (layout juggling, gradient plumbing) is right when the two extension modules are
replaced by checker-backed fakes (test-only; the product itself has no CPU path)."""
import ctypes
import importlib
import os
import re
import sys
import types

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import ROOT, load_golden

import vms_hip
import causal_conv1d
import causal_conv1d_cuda
import selective_scan_cuda
import mamba_ssm
from causal_conv1d.causal_conv1d_interface import causal_conv1d_ref, causal_conv1d_update_ref
from mamba_ssm.ops import selective_scan_interface as ssi


def T(a, dtype=torch.float32, grad=False):
    t = torch.tensor(np.asarray(a), dtype=dtype)
    return t.requires_grad_() if grad else t


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "vms_hip.h")).read()
    declared = set(re.findall(r"\b(vms_[a-z0-9_]+)\s*\(", header))
    assert set(vms_hip.EXPORTS) == declared, (sorted(declared), sorted(vms_hip.EXPORTS))
    L = ctypes.CDLL(vms_hip.LIB_PATH)
    for name in declared:
        assert hasattr(L, name), name
    assert vms_hip.lib().vms_abi_version() == vms_hip.ABI_VERSION == 11


def test_no_cpu_fallback():
    u = torch.randn(1, 2, 8)
    A = -torch.rand(2, 4)
    B = torch.randn(1, 4, 8)
    with pytest.raises(RuntimeError):
        ssi.selective_scan_fn(u, u.abs(), A, B, B)
    with pytest.raises(RuntimeError):
        causal_conv1d.causal_conv1d_fn(u, torch.randn(2, 4))
    with pytest.raises(NotImplementedError):
        causal_conv1d.causal_conv1d_fn(u, torch.randn(2, 4), activation="gelu")
    with pytest.raises(RuntimeError):  # width check comes before any launch (causal_conv1d.cpp:157)
        causal_conv1d_cuda.causal_conv1d_fwd(u, torch.randn(2, 5), None, False)


def test_product_does_not_touch_checker_or_reference():
    pkg = os.path.join(ROOT, "video-mamba-suite_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cuh", ".h")):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, re.M), f"{f} imports the oracle"
                assert "vms_oracle" not in src, f
                assert "/root/reference" not in src, f


def test_public_names():
    for n in ("selective_scan_fn", "selective_scan_ref", "mamba_inner_fn", "mamba_inner_fn_no_out_proj",
              "bimamba_inner_fn", "mamba_inner_ref", "bimamba_inner_ref", "SelectiveScanFn", "MambaInnerFn",
              "MambaInnerFnNoOutProj", "BiMambaInnerFn"):
        assert hasattr(ssi, n), n
    from mamba_ssm.modules.mamba_simple import Mamba, Block  # noqa: F401
    from mamba_ssm.modules.mamba_new import Mamba as M2  # noqa: F401
    from mamba_ssm.modules.mamba_simple_scan_norm import Mamba as M3  # noqa: F401
    from mamba_ssm.ops.triton.layernorm import RMSNorm, layer_norm_fn, rms_norm_fn  # noqa: F401
    from mamba_ssm.utils.generation import GenerationMixin, InferenceParams  # noqa: F401
    from mamba_ssm.utils.hf import load_config_hf, load_state_dict_hf  # noqa: F401
    assert mamba_ssm.Mamba is Mamba
    for n in ("fwd", "bwd"):
        assert callable(getattr(selective_scan_cuda, n))
    for n in ("causal_conv1d_fwd", "causal_conv1d_bwd", "causal_conv1d_update"):
        assert callable(getattr(causal_conv1d_cuda, n))


@pytest.mark.parametrize("name", ["scan_L128_g1", "scan_L372_g2", "scan_constBC", "scan_constB", "scan_constC",
                                  "scan_L1134_plain"])
def test_selective_scan_ref_matches_golden(name):
    g = load_golden(name)
    args = {k: T(g[k], grad=True) for k in ("u", "delta", "A", "B", "C")}
    opt = {k: (T(g[k], grad=True) if k in g else None) for k in ("D", "z", "delta_bias")}
    out, last = ssi.selective_scan_ref(args["u"], args["delta"], args["A"], args["B"], args["C"], opt["D"],
                                       z=opt["z"], delta_bias=opt["delta_bias"],
                                       delta_softplus=bool(g["softplus"]), return_last_state=True)
    torch.testing.assert_close(out, T(g["out"]), rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(last, T(g["last_state"]), rtol=1e-4, atol=1e-4)
    out.backward(T(g["g"]))
    for k, gk in (("u", "du"), ("delta", "ddelta"), ("A", "dA"), ("B", "dB"), ("C", "dC")):
        torch.testing.assert_close(args[k].grad, T(g[gk]), rtol=2e-3, atol=2e-3)


@pytest.mark.parametrize("name", ["conv_L151_W4_b1_s1", "conv_L8_W2_b0_s0", "conv_L372_W3_b1_s0"])
def test_conv_ref_matches_golden(name):
    g = load_golden(name)
    out = causal_conv1d_ref(T(g["x"]), T(g["weight"]), T(g["bias"]) if "bias" in g else None,
                            "silu" if g["silu"] else None)
    torch.testing.assert_close(out, T(g["out"]), rtol=1e-5, atol=1e-5)


def test_conv_update_ref_matches_golden():
    g = load_golden("convupd_W4")
    cs = T(g["conv_state_in"])
    out = causal_conv1d_update_ref(T(g["x"]), cs, T(g["weight"]), T(g["bias"]), "silu")
    assert torch.equal(cs, T(g["conv_state_out"]))
    torch.testing.assert_close(out, T(g["out"]), rtol=1e-5, atol=1e-5)


# ---- host-logic tests with checker-backed fake extensions ----------------------------------------
@pytest.fixture
def fake_extensions(monkeypatch, oracle):
    """Replace the two extension modules seen by the interface code by CPU fakes that follow the
    extension ABI (same argument lists / returns) and compute with the C oracle."""
    from fake_ext import make_fakes, make_norm_fake
    fs, fc = make_fakes(oracle)
    import mamba_ssm.ops.triton.layernorm as lnm
    monkeypatch.setattr(lnm, "layer_norm_cuda", make_norm_fake(oracle))
    monkeypatch.setattr(ssi, "selective_scan_cuda", fs)
    monkeypatch.setattr(ssi, "causal_conv1d_cuda", fc)
    import causal_conv1d.causal_conv1d_interface as cci
    monkeypatch.setattr(cci, "causal_conv1d_cuda", fc)
    return fs, fc


INNER_KEYS = ("xz", "conv1d_weight", "conv1d_bias", "x_proj_weight", "delta_proj_weight", "out_proj_weight",
              "A", "A_b", "D", "delta_bias")


@pytest.mark.parametrize("kind", ["no_out_proj", "out_proj", "bi"])
def test_fused_inner_host_logic(fake_extensions, kind):
    g = load_golden("inner_" + kind)
    t = {k: T(g[k], grad=True) for k in INNER_KEYS}
    if kind == "no_out_proj":
        out = ssi.mamba_inner_fn_no_out_proj(t["xz"], t["conv1d_weight"], t["conv1d_bias"], t["x_proj_weight"],
                                             t["delta_proj_weight"], t["A"], None, None, t["D"],
                                             delta_bias=t["delta_bias"], delta_softplus=True)
    elif kind == "out_proj":
        out = ssi.mamba_inner_fn(t["xz"], t["conv1d_weight"], t["conv1d_bias"], t["x_proj_weight"],
                                 t["delta_proj_weight"], t["out_proj_weight"], None, t["A"], None, None, t["D"],
                                 delta_bias=t["delta_bias"], delta_softplus=True)
    else:
        out = ssi.bimamba_inner_fn(t["xz"], t["conv1d_weight"], t["conv1d_bias"], t["x_proj_weight"],
                                   t["delta_proj_weight"], t["out_proj_weight"], None, t["A"], t["A_b"], None, None,
                                   t["D"], delta_bias=t["delta_bias"], delta_softplus=True)
    ref = T(g["out"])
    scale = ref.abs().max().item()
    assert (out - ref).abs().max().item() <= 2e-4 * scale
    out.backward(T(g["g"]))
    for k in INNER_KEYS:
        gk = "d" + k
        if gk in g:
            ref_g = T(g[gk])
            err = (t[k].grad - ref_g).abs().max().item()
            assert err <= 5e-4 * max(1.0, ref_g.abs().max().item()), (k, err)


@pytest.mark.parametrize("name", ["inner768_out_proj_B0C0_f32", "inner768_out_proj_B0C1_f32", "inner768_out_proj_B1C0_f32_pbias",
                                  "inner768_no_out_proj_B1C1_f32_pbias", "inner768_bi_B1C1_f32_pbias", "inner768_bi_B0C0_f32",
                                  "inner768_out_proj_B1C1_c64_pbias"])
def test_inner768_host_logic(fake_extensions, name):
    """The nodes' branches that no suite model takes -- constant B / C, the projection biases (SSI:164, 324, 358-359), the
    bidirectional function with either -- at the reference's own test problem (test_selective_scan.py:152-199), over the
    oracle-backed extension stand-ins: the host side of tests/test_hip_parity.py::test_inner768_vs_reference_fixtures."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from recipes import INNER768_CASES, checksum, inner768_inputs, sample
    fn, var_B, var_C, is_complex, pbias = INNER768_CASES[name]
    g = load_golden(name)
    t = inner768_inputs(var_B, var_C, is_complex, pbias)
    for k, v in t.items():
        if v is not None:
            np.testing.assert_allclose(np.array(checksum(v)), g["in_sum." + k], rtol=1e-10, err_msg=f"input {k} differs from the fixture's")
    leaves = {k: v.clone().requires_grad_() for k, v in t.items() if v is not None and not k.startswith("g_")}
    a = leaves.get
    if fn == "out_proj":
        out = ssi.mamba_inner_fn(a("xz"), a("conv1d_weight"), a("conv1d_bias"), a("x_proj_weight"), a("delta_proj_weight"),
                                 a("out_proj_weight"), None, a("A"), a("B"), a("C"), a("D"), delta_bias=a("delta_bias"),
                                 B_proj_bias=a("B_proj_bias"), C_proj_bias=a("C_proj_bias"), delta_softplus=True)
        gout = t["g_out_proj"]
    elif fn == "bi":
        out = ssi.bimamba_inner_fn(a("xz"), a("conv1d_weight"), a("conv1d_bias"), a("x_proj_weight"), a("delta_proj_weight"),
                                   a("out_proj_weight"), None, a("A"), a("A_b"), a("B"), a("C"), a("D"), delta_bias=a("delta_bias"),
                                   B_proj_bias=a("B_proj_bias"), C_proj_bias=a("C_proj_bias"), delta_softplus=True)
        gout = t["g_out_proj"]
    else:
        out = ssi.mamba_inner_fn_no_out_proj(a("xz"), a("conv1d_weight"), a("conv1d_bias"), a("x_proj_weight"), a("delta_proj_weight"),
                                             a("A"), a("B"), a("C"), a("D"), delta_bias=a("delta_bias"),
                                             B_proj_bias=a("B_proj_bias"), C_proj_bias=a("C_proj_bias"), delta_softplus=True)
        gout = t["g_no_out_proj"]

    def cmp(got, key, tol):
        sm, stride = sample(got.detach(), 16384 if key == "out" else 8192)
        assert stride == int(g[key + ".stride"]), key
        ref = torch.from_numpy(g[key])
        assert (sm - ref).abs().max().item() <= tol * max(1.0, ref.abs().max().item()), key

    cmp(out, "out", 1e-3)
    out.backward(gout)
    for k, leaf in leaves.items():
        if "d" + k in g:
            assert leaf.grad is not None, f"no gradient for {k}"
            cmp(leaf.grad, "d" + k, 2e-3)
        else:
            assert leaf.grad is None or float(leaf.grad.abs().max()) == 0.0, k


@pytest.mark.parametrize("name,which,kw", [
    ("block_vim", "mamba_simple", dict(bimamba_type="v2")),
    ("block_vim_div", "mamba_simple", dict(bimamba_type="v2", if_devide_out=True)),
    ("block_vim_norm", "mamba_simple_scan_norm", dict(bimamba_type="v2", if_devide_out=True)),
    ("block_dbm", "mamba_new", dict(expand=1)),
    # round 6: d_state = 4 (the suite's CLIP ViViM, avion/models/model_clip.py:945-947) and expand = 2 at d_state 16
    ("block_vim_n4_div", "mamba_simple", dict(bimamba_type="v2", if_devide_out=True)),
    ("block_vim_n4", "mamba_simple", dict(bimamba_type="v2")),
    ("block_dbm_n4", "mamba_new", dict(expand=1)),
    ("block_vim_e2_n16", "mamba_simple", dict(bimamba_type="v2")),
])
@pytest.mark.parametrize("fast", [True, False])
def test_block_host_logic(fake_extensions, name, which, kw, fast):
    if which == "mamba_new" and not fast:
        pytest.skip("DBM has no slow path (reference mamba_new.py:216)")
    g = load_golden(name)
    Mamba = importlib.import_module("mamba_ssm.modules." + which).Mamba
    sd = {k[3:]: T(v) for k, v in g.items() if k.startswith("sd.")}
    m = Mamba(g["x"].shape[-1], d_state=g["sd.A_log"].shape[1], d_conv=4, use_fast_path=fast, **({"expand": 2} | kw))
    assert sorted(m.state_dict().keys()) == sorted(sd.keys())
    m.load_state_dict(sd)
    x = T(g["x"], grad=True)
    y = m(x)
    ref = T(g["y"])
    assert (y - ref).abs().max().item() <= 3e-4 * max(1.0, ref.abs().max().item())
    y.backward(T(g["g"]))
    ref_dx = T(g["dx"])
    assert (x.grad - ref_dx).abs().max().item() <= 1e-3 * max(1.0, ref_dx.abs().max().item())
    for k, p in m.named_parameters():
        rg = T(g["grad." + k])
        err = (p.grad - rg).abs().max().item()
        assert err <= 2e-3 * max(1.0, rg.abs().max().item()), (k, err, rg.abs().max().item())


# ---- fused add + norm: host logic of LayerNormFn over a checker-backed fake extension ----------------------
@pytest.mark.parametrize("name", ["norm_ln_N64_r1b1p1", "norm_rms_N192_r0b0p0", "norm_rms_N1000_r1b0p0",
                                  "norm_ln_N1024_r1b1p1"])
def test_norm_fn_host_logic(fake_extensions, name):
    import mamba_ssm.ops.triton.layernorm as lnm
    g = load_golden(name)
    x, w = T(g["x"], grad=True), T(g["weight"], grad=True)
    b = T(g["bias"], grad=True) if "bias" in g else None
    res = T(g["residual"], grad=True) if "residual" in g else None
    prenorm = bool(g["prenorm"])
    fn = lnm.rms_norm_fn if g["is_rms"] else lnm.layer_norm_fn
    out = fn(x, w, b, residual=res, eps=float(g["eps"]), prenorm=prenorm)
    y, pre = out if prenorm else (out, None)
    torch.testing.assert_close(y, T(g["y"]), rtol=2e-4, atol=2e-5)
    loss = (y * T(g["g"])).sum()
    if prenorm:
        torch.testing.assert_close(pre, T(g["pre"]), rtol=2e-4, atol=2e-5)
        loss = loss + (pre * T(g["gpre"])).sum()
    loss.backward()
    torch.testing.assert_close(x.grad, T(g["dx"]), rtol=1e-3, atol=1e-4)
    torch.testing.assert_close(w.grad, T(g["dweight"]), rtol=1e-3, atol=1e-3)
    if b is not None:
        torch.testing.assert_close(b.grad, T(g["dbias"]), rtol=1e-3, atol=1e-3)
    if res is not None:
        torch.testing.assert_close(res.grad, T(g["dresidual"]), rtol=1e-3, atol=1e-4)


def test_projection_nodes_match_plain_autograd():
    """in_proj_fn / out_proj_fn (mamba_ssm/ops/projections.py): same values and gradients as the reference's
    rearrange(W @ rearrange(x)) (mamba_simple.py:144-149) and out_proj(y^T) left to autograd."""
    import torch
    from mamba_ssm.ops.projections import _k_splits, in_proj_fn, out_proj_fn
    torch.manual_seed(0)
    B, L, dm, C = 3, 40, 16, 24
    x = torch.randn(B, L, dm, requires_grad=True)
    w = torch.randn(C, dm, requires_grad=True)
    b = torch.randn(C, requires_grad=True)
    xz = in_proj_fn(x, w, b)
    ref = (w @ x.reshape(B * L, dm).t()).view(C, B, L).permute(1, 0, 2) + b[:, None]
    assert xz.shape == (B, C, L) and xz.stride(2) == 1
    torch.testing.assert_close(xz, ref)
    g = torch.randn(B, C, L)
    got = torch.autograd.grad(xz, (x, w, b), g)
    want = torch.autograd.grad(ref, (x, w, b), g)
    for a, e in zip(got, want):
        torch.testing.assert_close(a, e, rtol=1e-5, atol=1e-5)
    # a gradient that does not have xz's memory layout takes the copy path
    got2 = torch.autograd.grad(in_proj_fn(x, w, None), (x, w), g.contiguous())
    want2 = torch.autograd.grad((w @ x.reshape(B * L, dm).t()).view(C, B, L).permute(1, 0, 2), (x, w), g)
    for a, e in zip(got2, want2):
        torch.testing.assert_close(a, e, rtol=1e-5, atol=1e-5)

    y = torch.randn(B, C, L, requires_grad=True)
    wo = torch.randn(dm, C, requires_grad=True)
    bo = torch.randn(dm, requires_grad=True)
    out = out_proj_fn(y, wo, bo)
    ref = torch.nn.functional.linear(y.transpose(1, 2), wo, bo)
    torch.testing.assert_close(out, ref)
    go = torch.randn(B, L, dm)
    got = torch.autograd.grad(out, (y, wo, bo), go)
    want = torch.autograd.grad(ref, (y, wo, bo), go)
    for a, e in zip(got, want):
        torch.testing.assert_close(a, e, rtol=1e-5, atol=1e-5)
    assert got[0].stride(2) == 1   # (B, C, L) with a unit seqlen stride: what the scan backward reads
    # y in the scan's channel-slowest layout (strides (L, B L, 1)): the weight gradient takes K slices of the flattened rows
    import mamba_ssm.ops.projections as proj
    y_cs = torch.randn(C, B, L).permute(1, 0, 2).requires_grad_()
    old = proj._k_splits
    proj._k_splits = lambda k, r, c: 4                      # 120 rows in 4 slices of 30 (the rule wants >= 1024 rows per slice)
    try:
        got3 = torch.autograd.grad(out_proj_fn(y_cs, wo, bo), (y_cs, wo, bo), go)
    finally:
        proj._k_splits = old
    want3 = torch.autograd.grad(torch.nn.functional.linear(y_cs.transpose(1, 2), wo, bo), (y_cs, wo, bo), go)
    for a, e in zip(got3, want3):
        torch.testing.assert_close(a, e, rtol=1e-5, atol=1e-5)
    # slices x 256-wide output tiles ~ one workgroup per CU, >= 1024 rows per slice, the count divides K
    assert _k_splits(8 * 8192) == 8 and _k_splits(197) == 1 and _k_splits(65536, 1536, 768) == 16
    assert _k_splits(8 * 3136, 1536, 768) == 14 and _k_splits(65536, 1024, 1024) == 16 and _k_splits(4608, 2048, 512) == 1


@pytest.mark.parametrize("name", ["stack_ln", "stack_rms_fp32res"])
@pytest.mark.parametrize("fused_add_norm", [False, True])
def test_block_stack_host_logic(fake_extensions, name, fused_add_norm):
    """Block.forward (reference mamba_simple.py:381-437) wired three deep + the closing add / norm_f, against the
    fixture the reference's own Block produced: residual dtype, prenorm return order, eps, both add+norm forms."""
    from conftest import build_stack
    g = load_golden(name)
    layers, norm_f, run = build_stack(g, fused_add_norm=fused_add_norm)
    x = T(g["x"], grad=True)
    y = run(x)
    ref = T(g["y"])
    assert y.dtype == ref.dtype
    assert (y - ref).abs().max().item() <= 5e-4 * max(1.0, ref.abs().max().item())
    y.backward(T(g["g"]))
    ref_dx = T(g["dx"])
    assert (x.grad - ref_dx).abs().max().item() <= 2e-3 * max(1.0, ref_dx.abs().max().item())
    for prefix, mod in (("layers.", layers), ("norm_f.", norm_f)):
        for k, p in mod.named_parameters():
            rg = T(g["grad." + prefix + k])
            err = (p.grad - rg).abs().max().item()
            assert err <= 3e-3 * max(1.0, rg.abs().max().item()), (prefix + k, err, rg.abs().max().item())


def test_complex_A_has_no_cpu_fallback():
    """Complex A runs on its own HIP kernels (csrc/selective_scan_complex.hip): like the real case, CPU tensors raise."""
    torch.manual_seed(0)
    b, d, n, L = 2, 4, 8, 24
    u, delta = torch.randn(b, d, L), torch.rand(b, d, L)
    A = torch.complex(-torch.rand(d, n), torch.randn(d, n))
    B, C = torch.randn(b, n, 2 * L), torch.randn(b, n, 2 * L)   # interleaved (re, im) along L (SSI:100-104)
    with pytest.raises(RuntimeError):
        ssi.selective_scan_fn(u, delta, A, B, C, delta_softplus=True)
    with pytest.raises(RuntimeError):
        selective_scan_cuda.fwd(u, delta, A, B.unsqueeze(1), C.unsqueeze(1), None, None, None, True)


@pytest.mark.parametrize("var_B,var_C", [(True, True), (False, True), (True, False), (False, False)])
def test_complex_A_host_logic(fake_extensions, var_B, var_C):
    """selective_scan_fn / mamba_inner_fn with a complex A over the extension stand-ins (the oracle's complex scan):
    values and every gradient against autograd through the PyTorch statement of the op."""
    torch.manual_seed(1)
    b, d, n, L = 2, 4, 4, 40
    mk = lambda *s, **k: torch.randn(*s, **k).requires_grad_()
    u, delta = mk(b, d, L), torch.rand(b, d, L).requires_grad_()
    A = (-0.5 * torch.rand(d, n, dtype=torch.complex64)).requires_grad_()
    B = mk(b, n, 2 * L) if var_B else mk(d, n, dtype=torch.complex64)
    C = mk(b, 1, n, 2 * L) if var_C else mk(d, n, dtype=torch.complex64)
    D, z, bias = mk(d), mk(b, d, L), torch.rand(d).requires_grad_()
    ins = (u, delta, A, B, C, D, z, bias)
    out, last = ssi.selective_scan_fn(u, delta, A, B, C, D, z, bias, delta_softplus=True, return_last_state=True)
    g = torch.randn_like(out)
    grads = torch.autograd.grad(out, ins, g)
    ref, last_ref = ssi.selective_scan_ref(u, delta, A, B, C, D, z, bias, delta_softplus=True, return_last_state=True)
    grads_ref = torch.autograd.grad(ref, ins, g)
    assert last.dtype == torch.complex64 and (last - last_ref).abs().max().item() < 1e-4
    assert (out - ref).abs().max().item() < 1e-4
    for a, r_, name in zip(grads, grads_ref, "u delta A B C D z bias".split()):
        assert a.dtype == r_.dtype and a.shape == r_.shape, name
        assert (a - r_).abs().max().item() < 2e-4 * max(1.0, r_.abs().max().item()), name


def test_complex_A_inner_fn_host_logic(fake_extensions):
    """mamba_inner_fn with a complex A = the composition of the ops (the reference tests it: test_selective_scan.py:152-250)."""
    torch.manual_seed(2)
    b, d, n, R, L, W = 2, 8, 4, 3, 24, 3
    xz = torch.randn(b, 2 * d, L, requires_grad=True)
    conv_w, conv_b = torch.randn(d, 1, W, requires_grad=True), torch.randn(d, requires_grad=True)
    x_proj_w = torch.randn(R + 4 * n, d, requires_grad=True)     # d_state = 2 n for a complex A (SSI:168)
    dt_w = torch.randn(d, R, requires_grad=True)
    out_w = torch.randn(5, d, requires_grad=True)
    A = (-0.5 * torch.rand(d, n, dtype=torch.complex64)).requires_grad_()
    D, bias = torch.randn(d, requires_grad=True), torch.rand(d, requires_grad=True)
    ins = (xz, conv_w, conv_b, x_proj_w, dt_w, out_w, A, D, bias)
    out = ssi.mamba_inner_fn(xz, conv_w, conv_b, x_proj_w, dt_w, out_w, None, A, None, None, D, bias)
    g = torch.randn_like(out)
    grads = torch.autograd.grad(out, ins, g)
    # the same composition with the scan stated in PyTorch
    x, zz = xz.chunk(2, dim=1)
    x = ssi.causal_conv1d_fn(x, conv_w.squeeze(1), conv_b, "silu")
    x_dbl = F.linear(x.transpose(1, 2).reshape(b * L, -1), x_proj_w)
    dl = (dt_w @ x_dbl[:, :R].t()).view(d, b, L).permute(1, 0, 2)
    Bm = x_dbl[:, R:R + 2 * n].view(b, L, n, 2).permute(0, 2, 1, 3).reshape(b, n, 2 * L)
    Cm = x_dbl[:, -2 * n:].view(b, L, n, 2).permute(0, 2, 1, 3).reshape(b, n, 2 * L)
    y = ssi.selective_scan_ref(x, dl, A, Bm, Cm, D, zz, bias, delta_softplus=True)
    ref = F.linear(y.transpose(1, 2), out_w)
    grads_ref = torch.autograd.grad(ref, ins, g)
    assert (out - ref).abs().max().item() < 1e-3 * max(1.0, ref.abs().max().item())
    for a, r_ in zip(grads, grads_ref):
        assert (a - r_).abs().max().item() < 1e-3 * max(1.0, r_.abs().max().item())


def test_compiled_binding_speaks_the_packages_abi():
    """A stale _vms_torch.so (older ABI) used to be dropped silently: every call then ran on ctypes (round 4 found it so)."""
    header = open(os.path.join(ROOT, "include", "vms_hip.h")).read()
    assert int(re.search(r"#define VMS_ABI_VERSION (\d+)", header).group(1)) == vms_hip.ABI_VERSION
    so = os.path.join(ROOT, "video-mamba-suite_amd", "_vms_torch.so")
    if not os.path.exists(so) or vms_hip.debug.no_torch_ext or "VMS_HIP_LIB" in os.environ:
        pytest.skip("compiled binding not built / disabled")
    ext = vms_hip.ext()
    assert ext is not None and ext.abi_version() == vms_hip.ABI_VERSION
    for fn in ("scan_fwd", "scan_bwd", "scan_bwd_dual", "inner_fwd", "inner_bwd", "inner_bwd_dual", "conv_fwd_dual"):
        assert hasattr(ext, fn), fn


def test_checkpoint_layout_policy(monkeypatch):
    """vms_hip.x_mode_for_shape: which checkpoints a forward leaves (the 8-element layout -- 8 B D L bytes per scan --
    must not be unconditional).  Environment > per-thread context (the modules' scan_checkpoints=) > process default; "auto" on a
    CPU tensor cannot look at device memory and lets the library choose."""
    monkeypatch.delenv("VMS_X_LAYOUT", raising=False)
    shp = (8, 1024, 8192, 16, "cpu")
    assert vms_hip.current_x_layout_policy() == "auto"
    assert vms_hip.x_mode_for_shape(*shp, for_backward=False) == 1          # inference: the small layout, whatever the policy
    assert vms_hip.x_mode_for_shape(*shp) == -1
    with vms_hip.x_layout_policy("coarse"):
        assert vms_hip.x_mode_for_shape(*shp) == 1
        with vms_hip.x_layout_policy(None):                                  # None = no opinion: the enclosing one stays
            assert vms_hip.x_mode_for_shape(*shp) == 1
        monkeypatch.setenv("VMS_X_LAYOUT", "3")
        assert vms_hip.x_mode_for_shape(*shp) == -1                          # the environment wins
        monkeypatch.delenv("VMS_X_LAYOUT")
    assert vms_hip.current_x_layout_policy() == "auto"
    vms_hip.set_x_layout_policy("coarse")
    try:
        assert vms_hip.x_mode_for_shape(*shp) == 1
        with vms_hip.x_layout_policy("fine"):
            assert vms_hip.x_mode_for_shape(*shp) == -1
    finally:
        vms_hip.set_x_layout_policy("auto")
    monkeypatch.setenv("VMS_X_LAYOUT", "1")
    assert vms_hip.x_mode_for_shape(*shp) == 1
    # "auto" on a device: by the allocator's numbers -- and the small layout when the allocator keeps none (a pluggable allocator:
    # torch.cuda.memory_allocated raises; tools/efence_run.py met it)
    monkeypatch.delenv("VMS_X_LAYOUT")
    monkeypatch.setitem(vms_hip._total_mem, 0, 256 << 30)
    monkeypatch.setattr(vms_hip, "_allocated_bytes", lambda idx: 1 << 30)
    assert vms_hip.x_mode_for_shape(8, 1024, 8192, 16, "cuda:0") == -1
    monkeypatch.setattr(vms_hip, "_allocated_bytes", lambda idx: 200 << 30)
    assert vms_hip.x_mode_for_shape(8, 1024, 8192, 16, "cuda:0") == 1
    monkeypatch.setattr(vms_hip, "_allocated_bytes", lambda idx: None)
    assert vms_hip.x_mode_for_shape(8, 1024, 8192, 16, "cuda:0") == 1
    from mamba_ssm.modules.mamba_simple import Mamba
    from mamba_ssm.modules.mamba_new import Mamba as DBM
    assert Mamba(32, expand=1, bimamba_type="v2", scan_checkpoints="coarse").scan_checkpoints == "coarse"
    assert DBM(32, expand=1).scan_checkpoints is None


def test_auto_checkpoint_policy_is_decided_once_per_module_and_shape(monkeypatch):
    """under the memory-aware "auto" policy a module asks the allocator ONCE per input shape (at its first training forward of
    that shape) and keeps the answer -- the kernel choice must not follow unrelated allocations from step to step; an explicit
    scan_checkpoints=, a process policy other than "auto", inference and CPU tensors never consult the cache."""
    import types
    import torch
    import mamba_ssm.modules._core as core
    from mamba_ssm.modules.mamba_simple import Mamba
    monkeypatch.delenv("VMS_X_LAYOUT", raising=False)
    answers = iter([1, -1, -1])
    calls = []

    def fake_mode(batch, dim, seqlen, dstate, device, for_backward=True):
        calls.append((batch, dim, seqlen, dstate))
        return next(answers)
    monkeypatch.setattr(core._vms, "x_mode_for_shape", fake_mode)
    m = Mamba(32, expand=2, d_state=16, bimamba_type="v2")
    gpu = lambda b, l: types.SimpleNamespace(is_cuda=True, device=types.SimpleNamespace(index=0), shape=(b, l, 32))
    assert m._checkpoint_policy(gpu(4, 64)) == "coarse" and calls == [(4, 64, 64, 16)]       # (batch, d_inner, seqlen, d_state)
    assert m._checkpoint_policy(gpu(4, 64)) == "coarse" and len(calls) == 1                   # kept: no second query
    assert m._checkpoint_policy(gpu(4, 128)) == "fine" and len(calls) == 2                    # another shape: its own decision
    with torch.no_grad():
        assert m._checkpoint_policy(gpu(4, 256)) is None and len(calls) == 2                  # no backward: the node picks coarse itself
    with core._vms.x_layout_policy("fine"):
        assert m._checkpoint_policy(gpu(4, 64)) is None and len(calls) == 2                   # an enclosing explicit policy stays in charge
    assert m._checkpoint_policy(torch.zeros(4, 64, 32)) is None                               # CPU tensor
    m.reset_checkpoint_policy()
    assert m._checkpoint_policy(gpu(4, 64)) == "fine" and len(calls) == 3
    assert Mamba(32, expand=2, bimamba_type="v2", scan_checkpoints="coarse")._checkpoint_policy(gpu(4, 64)) == "coarse" and len(calls) == 3


def test_in_proj_weight_gradient_is_ready_before_the_input_gradient():
    """in_proj's weight gradient is the last parameter gradient of a block's backward: produced by its own autograd node AHEAD of
    the input-gradient GEMM, the reducer's all-reduce of the last DDP bucket overlaps that GEMM instead of trailing the step"""
    import torch
    from mamba_ssm.ops.projections import in_proj_fn
    torch.manual_seed(0)
    x = torch.randn(2, 12, 8, requires_grad=True)
    w = torch.randn(10, 8, requires_grad=True)
    b = torch.randn(10, requires_grad=True)
    order = []
    w.register_hook(lambda g: order.append("weight"))
    b.register_hook(lambda g: order.append("bias"))
    x.register_hook(lambda g: order.append("input"))
    in_proj_fn(x, w, b).square().sum().backward()
    assert order.index("weight") < order.index("input") and order.index("bias") < order.index("input"), order
    ref = (w @ x.detach().reshape(24, 8).t()).view(10, 2, 12).permute(1, 0, 2) + b[:, None]
    xr = x.detach().clone().requires_grad_()
    wr, br = w.detach().clone().requires_grad_(), b.detach().clone().requires_grad_()
    ((wr @ xr.reshape(24, 8).t()).view(10, 2, 12).permute(1, 0, 2) + br[:, None]).square().sum().backward()
    for a, e in ((x.grad, xr.grad), (w.grad, wr.grad), (b.grad, br.grad)):
        torch.testing.assert_close(a, e, rtol=1e-5, atol=1e-5)
    # an input that needs no gradient: one node, as before
    x2 = torch.randn(2, 12, 8)
    in_proj_fn(x2, w, b).sum().backward()


def test_isa_hazard_scanner_finds_trans_then_asm_fma(tmp_path):
    """tools/isa_hazards.py: a v_pk_fma_f32 directly behind the v_exp_f32 that writes one of its sources is reported, one with an
    instruction in between (or reading other registers) is not"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("isa_hazards", os.path.join(ROOT, "tools", "isa_hazards.py"))
    hz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(hz)
    bad = tmp_path / "bad.s"
    bad.write_text("\tv_exp_f32_e32 v18, v18\n\tv_exp_f32_e32 v19, v19\n\tv_pk_fma_f32 v[18:19], v[18:19], v[4:5], v[62:63]\n")
    ok = tmp_path / "ok.s"
    ok.write_text("\tv_exp_f32_e32 v19, v19\n\ts_nop 0\n\tv_pk_fma_f32 v[18:19], v[18:19], v[4:5], v[62:63]\n"
                  "\tv_rcp_f32_e32 v7, v7\n\tv_pk_fma_f32 v[18:19], v[18:19], v[4:5], v[62:63]\n"
                  "\tv_exp_f32_e32 v18, v18\n\tv_pk_fma_f32 v[18:19], v[20:21], v[4:5], v[62:63]\n")
    assert len(hz.scan(str(bad))) == 1
    assert hz.scan(str(ok)) == []


def test_no_trans_forwarding_hazard_in_the_pair_kernels():
    """the gfx950 ISA of the two sources with inline-asm VALU statements holds no transcendental -> asm-fma pair without a wait
    state (the compiler does not guard asm statements; tools/isa_hazards.py)"""
    import shutil
    import subprocess
    if shutil.which("hipcc") is None:
        pytest.skip("hipcc not on PATH")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_hazards.py")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


def test_padded_len_rule():
    """modules/_core.py _padded_len: multiples of 16, extended to whole 256-token GEMM tiles when <= 2 % more positions buy it"""
    from mamba_ssm.modules import _core
    for batch, seqlen, want in ((8, 3137, 3168), (8, 1569, 1600), (16, 1569, 1584), (2, 1569, 1584), (1, 3137, 3152), (2, 197, 208),
                                (64, 197, 208), (8, 35, 48), (32, 785, 800), (8, 3136, 3136), (3, 17, 32)):
        got = _core._padded_len(batch, seqlen, 16)
        assert got == want, (batch, seqlen, got, want)
        assert got % 16 == 0 and got >= seqlen and got - seqlen <= 15 + seqlen // 50
    m = _core.MambaCore(32, bimamba_type="v2")
    assert m._seq_padding(torch.zeros(2, 197, 32)) == 0      # CPU tensors are never padded


def test_x_pitch_short_rows_keep_the_reference_shape():
    """vms_scan_x_pitch (no GPU needed): problems the lane-per-row kernels take (selective_scan_short.hip: seqlen <= 16, thousands of
    rows) get the reference's x and nothing behind it -- the 8-element checkpoint layout would cost 16.5 KB per row whatever its
    length (20 GB per scan at TimeMamba's (1568, 8, 768)); long rows keep their checkpoints"""
    import vms_hip
    L = vms_hip.lib()

    def pitch(batch, dim, seqlen, mode=-1, **kw):
        P = vms_hip.ScanFwdParams()
        P.batch, P.dim, P.seqlen, P.dstate, P.n_groups, P.n_chunks = batch, dim, seqlen, 16, 1, (seqlen + 2047) // 2048
        P.dtype, P.is_variable_B, P.is_variable_C, P.delta_softplus = vms_hip.VMS_BF16, 1, 1, 1
        for k, v in kw.items():
            setattr(P, k, v)
        return L.vms_scan_x_pitch(ctypes.byref(P), mode)
    assert pitch(1568, 768, 8) == 32 and pitch(1568, 768, 16) == 32 and pitch(1568, 768, 8, mode=1) == 32
    assert pitch(1568, 768, 8, reverse_from=784) == 32                 # both halves of a mixed-direction batch
    big = (18 * 16, 258 * 16)                                          # room for the long-row kernels' checkpoints
    # rows of 17 .. 64 elements (round 6): chained 16-element segments, the states between them behind the reference-shaped slots
    assert pitch(1568, 768, 32) == 32 + 16 and pitch(1568, 768, 64) == 32 + 3 * 16 and pitch(1568, 768, 17, mode=1) == 32 + 16
    assert pitch(1568, 768, 32, mode=0) == 32                          # (no backward intended: the reference's own pitch)
    assert pitch(1568, 768, 80) in big                                 # longer than the lane-per-row kernels serve
    assert pitch(4, 768, 16) in big                                    # too few rows
    assert pitch(1568, 768, 8, is_variable_B=0) in big                 # constant B
    assert pitch(1568, 768, 8, impl=vms_hip.IMPL_GENERIC) in big       # the caller forced another kernel generation
    assert pitch(8, 768, 3136) in big and pitch(8, 768, 3136, mode=1) == 18 * 16
    assert pitch(1568, 768, 8, mode=0) == 32 and pitch(8, 768, 3136, mode=0) == 32
