"""Pins the CPU oracle (oracle/vms_oracle.c) against the golden vectors produced by the
reference's own pure-PyTorch path (tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest

from conftest import golden_names, load_golden


def close(a, b, rtol, atol, what=""):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    err = np.abs(a - b) - (atol + rtol * np.abs(b))
    assert a.shape == b.shape, (what, a.shape, b.shape)
    assert err.max() <= 0, f"{what}: max abs diff {np.abs(a - b).max():.3e} (rtol {rtol}, atol {atol})"


def tol_for(g):
    it = str(g["itype"])
    if "bfloat16" in it:
        return 2e-2, 2e-2      # golden outputs/grads were rounded to bf16 by the reference
    if "float16" in it:
        return 3e-3, 3e-3
    return 2e-4, 2e-5          # fp32: restatement vs torch differ only by summation order


@pytest.mark.parametrize("name", golden_names("scan_"))
@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_scan_oracle_matches_reference(oracle, name, prec):
    g = load_golden(name)
    rtol, atol = tol_for(g)
    sp = bool(g["softplus"])
    r = oracle.scan_fwd(g["u"], g["delta"], g["A"], g["B"], g["C"], g.get("D"), g.get("z"),
                        g.get("delta_bias"), sp, prec=prec)
    out = r["out_z"] if "z" in g else r["out"]
    close(out, g["out"], rtol, atol * 10, "out")
    close(r["last_state"], g["last_state"], rtol, atol * 10, "last_state")
    # checkpoints: last chunk's odd slots are the last state (SSI:40)
    close(r["x"][:, :, -1, 1::2], g["last_state"], rtol, atol * 10, "x[-1,1::2]")
    b = oracle.scan_bwd(g["u"], g["delta"], g["A"], g["B"], g["C"], g.get("D"), g.get("z"),
                        g.get("delta_bias"), g["g"], sp, prec=prec)
    L = g["u"].shape[-1]
    scale = max(1.0, L / 256)  # sums over L of O(1) terms: absolute noise grows with L
    for k in ("du", "ddelta", "dB", "dC", "dz"):
        if k in g:
            close(b[k], g[k], rtol, atol * 50, k)
    for k in ("dA", "dD", "ddelta_bias"):
        if k in g:
            close(b[k], g[k], rtol * 5, atol * 50 * scale, k)


def cclose(a, b, rtol, atol, what=""):
    """complex or real arrays: real and imaginary parts separately"""
    a, b = np.asarray(a), np.asarray(b)
    if np.iscomplexobj(a) or np.iscomplexobj(b):
        close(a.real, b.real, rtol, atol, what + ".re")
        close(a.imag, b.imag, rtol, atol, what + ".im")
    else:
        close(a, b, rtol, atol, what)


@pytest.mark.parametrize("name", golden_names("cscan_"))
@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_complex_scan_oracle_matches_reference(oracle, name, prec):
    """Complex A (SSI:111-116, 144-145): forward, last state and every gradient against torch.autograd through the
    reference's selective_scan_ref."""
    g = load_golden(name)
    rtol, atol = tol_for(g)
    sp = bool(g["softplus"])
    r = oracle.cscan_fwd(g["u"], g["delta"], g["A"], g["B"], g["C"], g.get("D"), g.get("z"), g.get("delta_bias"), sp, prec=prec)
    out = r["out_z"] if "z" in g else r["out"]
    close(out, g["out"], rtol, atol * 10, "out")
    cclose(r["last_state"], g["last_state"], rtol, atol * 10, "last_state")
    cclose(r["x"][:, :, -1, 1::2], g["last_state"], rtol, atol * 10, "x[-1,1::2]")
    b = oracle.cscan_bwd(g["u"], g["delta"], g["A"], g["B"], g["C"], g.get("D"), g.get("z"), g.get("delta_bias"), g["g"], sp,
                         prec=prec)
    L = g["u"].shape[-1]
    scale = max(1.0, L / 256)
    var_B, var_C = not np.iscomplexobj(g["B"]), not np.iscomplexobj(g["C"])
    for k in ("du", "ddelta", "dz") + (("dB",) if var_B else ()) + (("dC",) if var_C else ()):
        if k in g:
            cclose(b[k], g[k], rtol, atol * 50, k)
    # 16-bit fixtures: the reference evaluates silu(z) in the input dtype (SSI:150 on a bf16 z), i.e. every term of the sums
    # over (batch, L) carries a 2^-9 relative perturbation the fp32 restatement does not have; sums that cancel (dA of a
    # weakly damped state) differ by that fraction of their LARGEST terms: the bar is relative to the tensor's largest entry
    half = "float16" in str(g["itype"])
    for k in ("dA", "dD", "ddelta_bias") + (() if var_B else ("dB",)) + (() if var_C else ("dC",)):
        if k in g:
            cclose(b[k], g[k], rtol * 5, max(atol * 50 * scale, 1e-2 * np.abs(g[k]).max() if half else 0.0), k)


def test_scan_oracle_mid_checkpoint(oracle):
    """x[c, 2n] is the state after the first 1024 elements of chunk c; check against a
    second oracle run on the truncated sequence."""
    g = load_golden("scan_L4100_long")
    sp = bool(g["softplus"])
    full = oracle.scan_fwd(g["u"], g["delta"], g["A"], g["B"], g["C"], g.get("D"), g.get("z"),
                           g.get("delta_bias"), sp, prec="f64")
    for c, cut in ((0, 1024), (1, 2048 + 1024), (2, 4100)):
        t = oracle.scan_fwd(g["u"][..., :cut], g["delta"][..., :cut], g["A"], g["B"][..., :cut],
                            g["C"][..., :cut], g.get("D"), g["z"][..., :cut], g.get("delta_bias"), sp,
                            prec="f64")
        close(full["x"][:, :, c, 0::2], t["last_state"], 1e-6, 1e-7, f"mid ckpt chunk {c}")
    for c, cut in ((0, 2048), (1, 4096), (2, 4100)):
        t = oracle.scan_fwd(g["u"][..., :cut], g["delta"][..., :cut], g["A"], g["B"][..., :cut],
                            g["C"][..., :cut], g.get("D"), g["z"][..., :cut], g.get("delta_bias"), sp,
                            prec="f64")
        close(full["x"][:, :, c, 1::2], t["last_state"], 1e-6, 1e-7, f"end ckpt chunk {c}")


@pytest.mark.parametrize("name", golden_names("conv_"))
@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_conv_oracle_matches_reference(oracle, name, prec):
    g = load_golden(name)
    rtol, atol = tol_for(g)
    silu = bool(g["silu"])
    out = oracle.conv_fwd(g["x"], g["weight"], g.get("bias"), silu, prec=prec)
    close(out, g["out"], rtol, atol * 10, "out")
    b = oracle.conv_bwd(g["x"], g["weight"], g.get("bias"), g["g"], silu, prec=prec)
    close(b["dx"], g["dx"], rtol, atol * 10, "dx")
    close(b["dweight"], g["dweight"], rtol * 5, atol * 100, "dweight")
    if "dbias" in g:
        close(b["dbias"], g["dbias"], rtol * 5, atol * 100, "dbias")


@pytest.mark.parametrize("name", golden_names("convupd_"))
def test_conv_update_oracle_matches_reference(oracle, name):
    g = load_golden(name)
    out, cs = oracle.conv_update(g["x"], g["conv_state_in"], g["weight"], g.get("bias"), bool(g["silu"]))
    assert np.array_equal(cs, g["conv_state_out"])  # state roll is exact (test_causal_conv1d.py:113)
    close(out, g["out"], 1e-5, 1e-6, "out")


@pytest.mark.parametrize("name", golden_names("norm_"))
@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_norm_oracle_matches_reference(oracle, name, prec):
    """fused add + LayerNorm / RMSNorm restatement vs layer_norm_ref / rms_norm_ref (layernorm.py:19-48, upcast)
    and their autograd gradients."""
    g = load_golden(name)
    is_rms, N = bool(g["is_rms"]), g["x"].shape[-1]
    x2 = g["x"].reshape(-1, N)
    res2 = g["residual"].reshape(-1, N) if "residual" in g else None
    r = oracle.norm_fwd(x2, g["weight"], g.get("bias"), res2, float(g["eps"]), is_rms, prec=prec)
    rtol, atol = (2e-2, 2e-2) if "bfloat16" in str(g["itype"]) else (2e-4, 2e-5)
    close(r["y"].reshape(g["y"].shape), g["y"], rtol, atol, "y")
    if "pre" in g:
        close(r["res_out"].reshape(g["pre"].shape), g["pre"], rtol, atol, "prenorm sum")
    dres = g["gpre"].reshape(-1, N) if "gpre" in g else None
    b = oracle.norm_bwd(r["res_out"], g["weight"], r["mean"], r["rstd"], g["g"].reshape(-1, N), dres, is_rms,
                        has_bias="bias" in g, prec=prec)
    close(b["ds"].reshape(g["dx"].shape), g["dx"], rtol * 5, atol * 20, "dx")
    if "dresidual" in g:
        close(b["ds"].reshape(g["dx"].shape), g["dresidual"], rtol * 5, atol * 20, "dresidual")
    rows = x2.shape[0]
    close(b["dw"], g["dweight"], rtol * 5, atol * 20 * rows, "dweight")
    if "bias" in g:
        close(b["db"], g["dbias"], rtol * 5, atol * 20 * rows, "dbias")


@pytest.mark.parametrize("name", golden_names("ssu_"))
@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_state_update_oracle_matches_reference(oracle, name, prec):
    """single-token SSM step vs selective_state_update_ref (selective_state_update.py:157-192)."""
    g = load_golden(name)
    out, st = oracle.state_update(g["state_in"], g["x"], g["dt"], g["A"], g["B"], g["C"], g.get("D"), g.get("z"),
                                  g.get("dt_bias"), bool(g["softplus"]), prec=prec)
    close(out, g["out"], 2e-4, 2e-5, "out")
    close(st, g["state_out"], 2e-4, 2e-5, "state")
