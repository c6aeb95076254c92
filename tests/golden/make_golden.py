"""Generate the golden vectors under tests/golden/ from the REFERENCE's own pure-PyTorch path.

Runs only in the build container (needs /root/reference); the reference never travels
to the GPU box -- only the .npz files written here do.  Import recipe:
the CUDA extensions the reference imports at module top are replaced by empty stub
modules, and the two interface files are loaded by file path, so only
  selective_scan_ref / mamba_inner_ref / bimamba_inner_ref      (SSI:86-152, 636-709)
  causal_conv1d_ref / causal_conv1d_update_ref                   (CCI:49-65, 87-104)
  Mamba(use_fast_path=False) of mamba_simple.py / mamba_simple_scan_norm.py
  Mamba.forward of mamba_new.py (DBM) over a composition of the refs
ever execute.  Input distributions follow the reference tests
(mamba/tests/ops/test_selective_scan.py:53-88, causal-conv1d/tests/test_causal_conv1d.py:36-50).

    python tests/golden/make_golden.py        # rewrites tests/golden/*.npz
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch
import torch.nn.functional as F

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def load_reference():
    for stub in ("causal_conv1d_cuda", "selective_scan_cuda"):
        sys.modules[stub] = types.ModuleType(stub)
    cci = _load("causal_conv1d.causal_conv1d_interface",
                f"{REF}/causal-conv1d/causal_conv1d/causal_conv1d_interface.py")
    pkg = types.ModuleType("causal_conv1d")
    pkg.causal_conv1d_fn = cci.causal_conv1d_ref  # fused fn -> pure-PyTorch ref
    pkg.causal_conv1d_update = cci.causal_conv1d_update_ref
    pkg.causal_conv1d_interface = cci
    sys.modules["causal_conv1d"] = pkg
    for name in ("mamba_ssm", "mamba_ssm.ops", "mamba_ssm.ops.triton", "mamba_ssm.modules"):
        m = types.ModuleType(name)
        m.__path__ = []
        sys.modules[name] = m
    ssi = _load("mamba_ssm.ops.selective_scan_interface",
                f"{REF}/mamba/mamba_ssm/ops/selective_scan_interface.py")
    # the *_inner_ref functions call the fused names: bind them to the refs
    ssi.causal_conv1d_fn = cci.causal_conv1d_ref
    ssi.selective_scan_fn = ssi.selective_scan_ref
    return cci, ssi


def load_reference_modules(ssi):
    """mamba_simple / scan_norm / mamba_new with every fused op bound to the refs."""
    # triton layernorm cannot import without triton: give the modules a torch RMSNorm stand-in
    ln = types.ModuleType("mamba_ssm.ops.triton.layernorm")

    class RMSNorm(torch.nn.Module):  # same math as layernorm.py rms_norm_ref (:35-48)
        def __init__(self, hidden_size, eps=1e-5, device=None, dtype=None):
            super().__init__()
            self.eps = eps
            self.weight = torch.nn.Parameter(torch.ones(hidden_size, device=device, dtype=dtype))
            self.register_parameter("bias", None)

        def forward(self, x):
            rstd = 1 / torch.sqrt((x.float().square()).mean(dim=-1, keepdim=True) + self.eps)
            return (x.float() * rstd * self.weight.float()).to(x.dtype)

    ln.RMSNorm, ln.layer_norm_fn, ln.rms_norm_fn = RMSNorm, None, None
    sys.modules["mamba_ssm.ops.triton.layernorm"] = ln
    ssu = types.ModuleType("mamba_ssm.ops.triton.selective_state_update")
    ssu.selective_state_update = None
    sys.modules["mamba_ssm.ops.triton.selective_state_update"] = ssu

    def inner_no_out_proj_ref(xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight,
                              A, B=None, C=None, D=None, delta_bias=None, B_proj_bias=None,
                              C_proj_bias=None, delta_softplus=True):
        # mamba_inner_ref (SSI:636-670) minus the final F.linear
        eye = torch.eye(A.shape[0], dtype=xz.dtype)
        y = ssi.mamba_inner_ref(xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight,
                                eye, None, A, B, C, D, delta_bias, B_proj_bias, C_proj_bias,
                                delta_softplus)
        return y.transpose(1, 2)  # (b l d) -> (b d l)

    ssi.mamba_inner_fn_no_out_proj = inner_no_out_proj_ref
    ssi.mamba_inner_fn = ssi.mamba_inner_ref
    ssi.bimamba_inner_fn = ssi.bimamba_inner_ref
    mods = {}
    for short, fname in (("simple", "mamba_simple.py"), ("norm", "mamba_simple_scan_norm.py"),
                         ("new", "mamba_new.py")):
        mods[short] = _load(f"mamba_ssm.modules._ref_{short}", f"{REF}/mamba/mamba_ssm/modules/{fname}")
    return mods


def npf(t):
    return None if t is None else t.detach().float().numpy()


def save(name, **arrs):
    arrs = {k: v for k, v in arrs.items() if v is not None}
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **arrs)
    print(f"  {name}.npz  {sum(np.asarray(v).nbytes for v in arrs.values()) / 1e6:.2f} MB raw")


# ---------------------------------------------------------------------------------------------
def gen_scan(ssi, name, batch, dim, dstate, L, groups=1, var_B=True, var_C=True, has_D=True,
             has_z=True, has_bias=True, softplus=True, itype=torch.float32, b3=False, seed=0):
    """Input recipe = test_selective_scan.py:53-88."""
    torch.random.manual_seed(seed)
    A = (-0.5 * torch.rand(dim, dstate)).requires_grad_()
    if not var_B:
        Bs = (dim, dstate)
    elif b3:
        Bs = (batch, dstate, L)
    else:
        Bs = (batch, groups, dstate, L)
    B = torch.randn(*Bs, dtype=torch.float32 if not var_B else itype).requires_grad_()
    if not var_C:
        Cs = (dim, dstate)
    elif b3:
        Cs = (batch, dstate, L)
    else:
        Cs = (batch, groups, dstate, L)
    C = torch.randn(*Cs, dtype=torch.float32 if not var_C else itype).requires_grad_()
    D = torch.randn(dim).requires_grad_() if has_D else None
    z = torch.randn(batch, dim, L, dtype=itype).requires_grad_() if has_z else None
    bias = (0.5 * torch.rand(dim)).requires_grad_() if has_bias else None
    u = torch.randn(batch, dim, L, dtype=itype).requires_grad_()
    delta = (0.5 * torch.rand(batch, dim, L, dtype=itype)).requires_grad_()
    out, last = ssi.selective_scan_ref(u, delta, A, B, C, D, z=z, delta_bias=bias,
                                       delta_softplus=softplus, return_last_state=True)
    g = torch.randn_like(out)
    out.backward(g)
    save(name, u=npf(u), delta=npf(delta), A=npf(A), B=npf(B), C=npf(C), D=npf(D), z=npf(z),
         delta_bias=npf(bias), softplus=np.array(int(softplus)), itype=np.array(str(itype)),
         out=npf(out), last_state=npf(last), g=npf(g),
         du=npf(u.grad), ddelta=npf(delta.grad), dA=npf(A.grad), dB=npf(B.grad), dC=npf(C.grad),
         dD=npf(D.grad) if has_D else None, dz=npf(z.grad) if has_z else None,
         ddelta_bias=npf(bias.grad) if has_bias else None)


def gen_cscan(ssi, name, batch, dim, dstate, L, groups=1, var_B=True, var_C=True, has_D=True, has_z=True, has_bias=True,
              softplus=True, itype=torch.float32, b3=False, seed=0):
    """Complex A: the recipe of test_selective_scan.py:53-88 with wtype = torch.complex64 (variable B / C are real
    (.., 2L) tensors of interleaved (re, im) pairs; constant ones are complex (dim, dstate))."""
    torch.random.manual_seed(seed)
    A = (-0.5 * torch.rand(dim, dstate, dtype=torch.complex64)).requires_grad_()

    def bc(var):
        if not var:
            return torch.randn(dim, dstate, dtype=torch.complex64).requires_grad_()
        shape = (batch, dstate, 2 * L) if b3 else (batch, groups, dstate, 2 * L)
        return torch.randn(*shape, dtype=itype).requires_grad_()

    B, C = bc(var_B), bc(var_C)
    D = torch.randn(dim).requires_grad_() if has_D else None
    z = torch.randn(batch, dim, L, dtype=itype).requires_grad_() if has_z else None
    bias = (0.5 * torch.rand(dim)).requires_grad_() if has_bias else None
    u = torch.randn(batch, dim, L, dtype=itype).requires_grad_()
    delta = (0.5 * torch.rand(batch, dim, L, dtype=itype)).requires_grad_()
    out, last = ssi.selective_scan_ref(u, delta, A, B, C, D, z=z, delta_bias=bias, delta_softplus=softplus,
                                       return_last_state=True)
    g = torch.randn_like(out)
    out.backward(g)

    def npc(t):   # complex tensors keep their dtype in the fixture
        return None if t is None else (t.detach().numpy() if t.is_complex() else t.detach().float().numpy())

    save(name, u=npf(u), delta=npf(delta), A=npc(A), B=npc(B), C=npc(C), D=npf(D), z=npf(z), delta_bias=npf(bias),
         softplus=np.array(int(softplus)), itype=np.array(str(itype)), out=npf(out), last_state=npc(last), g=npf(g),
         du=npf(u.grad), ddelta=npf(delta.grad), dA=npc(A.grad), dB=npc(B.grad), dC=npc(C.grad),
         dD=npf(D.grad) if has_D else None, dz=npf(z.grad) if has_z else None,
         ddelta_bias=npf(bias.grad) if has_bias else None)


def gen_cscan_all(ssi):
    gen_cscan(ssi, "cscan_L128_g1", 2, 4, 8, 128)
    gen_cscan(ssi, "cscan_L372_g2", 2, 4, 8, 372, groups=2)
    gen_cscan(ssi, "cscan_L1134_plain", 2, 4, 8, 1134, has_D=False, has_z=False, has_bias=False, softplus=False)
    gen_cscan(ssi, "cscan_L2048_b3", 1, 4, 8, 2048, b3=True)
    gen_cscan(ssi, "cscan_L2600_long", 1, 2, 4, 2600)
    gen_cscan(ssi, "cscan_constBC", 2, 4, 8, 200, var_B=False, var_C=False)
    gen_cscan(ssi, "cscan_constB", 2, 4, 8, 200, var_B=False)
    gen_cscan(ssi, "cscan_constC", 2, 4, 8, 200, var_C=False)
    gen_cscan(ssi, "cscan_bf16_L300", 2, 4, 16, 300, itype=torch.bfloat16)
    gen_cscan(ssi, "cscan_f16_L130", 2, 4, 8, 130, itype=torch.float16)


def gen_conv(cci, name, batch, dim, L, W, has_bias, silu, itype=torch.float32, seed=0):
    """Input recipe = test_causal_conv1d.py:36-50 (dim reduced)."""
    torch.random.manual_seed(seed)
    x = torch.randn(batch, dim, L, dtype=itype).requires_grad_()
    w = torch.randn(dim, W).requires_grad_()
    b = torch.randn(dim).requires_grad_() if has_bias else None
    out = cci.causal_conv1d_ref(x, w, b, activation="silu" if silu else None)
    g = torch.randn_like(out)
    out.backward(g)
    save(name, x=npf(x), weight=npf(w), bias=npf(b), silu=np.array(int(silu)), itype=np.array(str(itype)),
         out=npf(out), g=npf(g), dx=npf(x.grad), dweight=npf(w.grad),
         dbias=npf(b.grad) if has_bias else None)


def gen_conv_update(cci, name, batch, dim, W, has_bias, silu, seed=0):
    """test_causal_conv1d.py:90-110."""
    torch.random.manual_seed(seed)
    x = torch.randn(batch, dim)
    cs = torch.randn(batch, dim, W)
    w = torch.randn(dim, W)
    b = torch.randn(dim) if has_bias else None
    cs_in = cs.clone()
    out = cci.causal_conv1d_update_ref(x, cs, w, b, activation="silu" if silu else None)
    save(name, x=npf(x), conv_state_in=npf(cs_in), weight=npf(w), bias=npf(b), silu=np.array(int(silu)),
         out=npf(out), conv_state_out=npf(cs))


def gen_inner(ssi, name, kind, batch=2, dim=32, dstate=8, dt_rank=6, L=128, W=3, seed=0):
    """test_mamba_inner_fn recipe (test_selective_scan.py:166-200), dims reduced.
    kind in {"out_proj", "no_out_proj", "bi"}."""
    torch.random.manual_seed(seed)
    xz = torch.randn(batch, 2 * dim, L).requires_grad_()
    conv_w = torch.randn(dim, 1, W).requires_grad_()
    conv_b = torch.randn(dim).requires_grad_()
    x_proj_w = torch.randn(dt_rank + 2 * dstate, dim).requires_grad_()
    dt_proj_w = torch.randn(dim, dt_rank).requires_grad_()
    out_proj_w = torch.randn(dim // 2, dim).requires_grad_()
    A = (-0.5 * torch.rand(dim, dstate)).requires_grad_()
    A_b = (-0.5 * torch.rand(dim, dstate)).requires_grad_()
    D = torch.randn(dim).requires_grad_()
    dt_bias = (0.5 * torch.rand(dim)).requires_grad_()
    if kind == "out_proj":
        out = ssi.mamba_inner_ref(xz, conv_w, conv_b, x_proj_w, dt_proj_w, out_proj_w, None, A, None, None,
                                  D, delta_bias=dt_bias, delta_softplus=True)
    elif kind == "bi":
        out = ssi.bimamba_inner_ref(xz, conv_w, conv_b, x_proj_w, dt_proj_w, out_proj_w, None, A, A_b,
                                    None, None, D, delta_bias=dt_bias, delta_softplus=True)
    else:
        out = ssi.mamba_inner_fn_no_out_proj(xz, conv_w, conv_b, x_proj_w, dt_proj_w, A, None, None, D,
                                             delta_bias=dt_bias, delta_softplus=True)
    g = torch.randn_like(out)
    out.backward(g)
    save(name, xz=npf(xz), conv1d_weight=npf(conv_w), conv1d_bias=npf(conv_b), x_proj_weight=npf(x_proj_w),
         delta_proj_weight=npf(dt_proj_w), out_proj_weight=npf(out_proj_w), A=npf(A), A_b=npf(A_b), D=npf(D),
         delta_bias=npf(dt_bias), out=npf(out), g=npf(g),
         dxz=npf(xz.grad), dconv1d_weight=npf(conv_w.grad), dconv1d_bias=npf(conv_b.grad),
         dx_proj_weight=npf(x_proj_w.grad), ddelta_proj_weight=npf(dt_proj_w.grad),
         dout_proj_weight=npf(out_proj_w.grad) if out_proj_w.grad is not None else None,
         dA=npf(A.grad), dA_b=npf(A_b.grad) if A_b.grad is not None else None,
         dD=npf(D.grad), ddelta_bias=npf(dt_bias.grad))


def gen_inner768(ssi, name):
    """The reference's own inner-function problem (test_selective_scan.py:152-199: dim 768, dstate 8, dt_rank 48, L 128, W 3) for
    constant / variable B and C, real / complex A, with and without the projection biases (recipes.py).  The inputs are NOT
    stored (rebuilt from the seed by the tests; their checksums are): the fixture holds every k-th element of the output and of
    EVERY gradient the function returns, plus each tensor's sum and sum of squares."""
    from recipes import INNER768_CASES, checksum, inner768_inputs, sample
    fn, var_B, var_C, is_complex, pbias = INNER768_CASES[name]
    t = inner768_inputs(var_B, var_C, is_complex, pbias)
    leaves = {k: v.clone().requires_grad_() for k, v in t.items() if v is not None and not k.startswith("g_")}
    a = lambda k: leaves.get(k)
    if fn == "out_proj":
        out = ssi.mamba_inner_ref(a("xz"), a("conv1d_weight"), a("conv1d_bias"), a("x_proj_weight"), a("delta_proj_weight"),
                                  a("out_proj_weight"), None, a("A"), a("B"), a("C"), a("D"), delta_bias=a("delta_bias"),
                                  B_proj_bias=a("B_proj_bias"), C_proj_bias=a("C_proj_bias"), delta_softplus=True)
        g = t["g_out_proj"]
    elif fn == "bi":
        out = ssi.bimamba_inner_ref(a("xz"), a("conv1d_weight"), a("conv1d_bias"), a("x_proj_weight"), a("delta_proj_weight"),
                                    a("out_proj_weight"), None, a("A"), a("A_b"), a("B"), a("C"), a("D"), delta_bias=a("delta_bias"),
                                    B_proj_bias=a("B_proj_bias"), C_proj_bias=a("C_proj_bias"), delta_softplus=True)
        g = t["g_out_proj"]
    else:
        out = ssi.mamba_inner_fn_no_out_proj(a("xz"), a("conv1d_weight"), a("conv1d_bias"), a("x_proj_weight"), a("delta_proj_weight"),
                                             a("A"), a("B"), a("C"), a("D"), delta_bias=a("delta_bias"),
                                             B_proj_bias=a("B_proj_bias"), C_proj_bias=a("C_proj_bias"), delta_softplus=True)
        g = t["g_no_out_proj"]
    out.backward(g)
    arrs = {}
    for k, v in t.items():
        if v is not None:
            arrs["in_sum." + k] = np.array(checksum(v))
    o, stride = sample(out.detach(), 16384)
    arrs["out"], arrs["out.stride"], arrs["out.sums"] = o.numpy(), np.array(stride), np.array(checksum(out.detach()))
    for k, v in leaves.items():
        if v.grad is None:   # (A_b outside the bidirectional function, out_proj_weight outside the functions with an out_proj)
            continue
        sm, stride = sample(v.grad)
        arrs["d" + k], arrs["d" + k + ".stride"], arrs["d" + k + ".sums"] = sm.numpy(), np.array(stride), np.array(checksum(v.grad))
    save(name, **arrs)


def gen_block(mods, name, which, d_model=32, L=33, batch=2, expand=2, d_state=8, seed=0, **kw):
    """Reference nn.Module forward+backward on CPU; state_dict + input + output + grads."""
    torch.random.manual_seed(seed)
    cls = mods[which].Mamba
    extra = dict(bimamba_type="v2") if which in ("simple", "norm") else {}
    extra.update(kw)
    use_fast = which == "new"  # DBM has no slow path (mamba_new.py:216); its fast path is bound to refs
    m = cls(d_model, d_state=d_state, d_conv=4, expand=expand, use_fast_path=use_fast, **extra)
    # move parameters off their deterministic init so that every weight matters
    with torch.no_grad():
        for k, p in m.named_parameters():
            if k.endswith("A_log") or k.endswith("A_b_log"):
                p.add_(0.3 * torch.randn_like(p))
            elif k in ("D", "D_b") or k.endswith("norm.weight"):
                p.add_(0.5 * torch.randn_like(p))
    x = torch.randn(batch, L, d_model).requires_grad_()
    y = m(x)
    g = torch.randn_like(y)
    y.backward(g)
    arrs = {"sd." + k: npf(v) for k, v in m.state_dict().items()}
    arrs.update({"grad." + k: npf(p.grad) for k, p in m.named_parameters()})
    save(name, x=npf(x), y=npf(y), g=npf(g), dx=npf(x.grad), **arrs)


def gen_blocks6(mods):
    """module-level fixtures of round 6: d_inner a multiple of 32 and lengths the mixers pad to whole vectors, so that
    the GPU test runs the dstate-4 instantiations of the fast kernels; the fixture itself is the reference's slow path on CPU"""
    gen_block(mods, "block_vim_n4_div", "simple", d_model=64, L=81, batch=2, expand=2, d_state=4, seed=21, if_devide_out=True)
    gen_block(mods, "block_vim_n4", "simple", d_model=32, L=1041, batch=1, expand=2, d_state=4, seed=22)
    gen_block(mods, "block_dbm_n4", "new", d_model=64, L=96, batch=2, expand=1, d_state=4, seed=23)
    gen_block(mods, "block_vim_e2_n16", "simple", d_model=64, L=160, batch=2, expand=2, d_state=16, seed=24)


def gen_stack(mods, name, norm="ln", residual_in_fp32=False, n_layers=3, d_model=32, L=37, batch=2, seed=0):
    """A stack of the reference's own Block (mamba_simple.py:381-437, fused_add_norm=False: no Triton) around its ViM
    mixer (use_fast_path=False), closed the way the suite's backbones close it (final add + norm_f).  norm = "ln":
    nn.LayerNorm; "rms": the RMSNorm stand-in of load_reference_modules (the math of rms_norm_ref)."""
    from functools import partial
    torch.random.manual_seed(seed)
    ref = mods["simple"]
    RMS = sys.modules["mamba_ssm.ops.triton.layernorm"].RMSNorm
    norm_cls = partial(torch.nn.LayerNorm, eps=1e-5) if norm == "ln" else partial(RMS, eps=1e-5)
    mixer_cls = partial(ref.Mamba, d_state=8, d_conv=4, expand=2, bimamba_type="v2", use_fast_path=False)
    layers = torch.nn.ModuleList([ref.Block(d_model, mixer_cls, norm_cls=norm_cls, fused_add_norm=False,
                                            residual_in_fp32=residual_in_fp32) for _ in range(n_layers)])
    norm_f = norm_cls(d_model)
    with torch.no_grad():
        for k, p_ in list(layers.named_parameters()) + list(norm_f.named_parameters()):
            if k.endswith("A_log") or k.endswith("A_b_log"):
                p_.add_(0.3 * torch.randn_like(p_))
            elif k.endswith(".D") or k.endswith(".D_b") or "norm" in k or k in ("weight", "bias"):
                p_.add_(0.3 * torch.randn_like(p_))
    x = torch.randn(batch, L, d_model).requires_grad_()
    h, res = x, None
    for blk in layers:
        h, res = blk(h, res)
    y = norm_f((h + res).to(norm_f.weight.dtype))
    g = torch.randn_like(y)
    y.backward(g)
    arrs = {"sd.layers." + k: npf(v) for k, v in layers.state_dict().items()}
    arrs.update({"sd.norm_f." + k: npf(v) for k, v in norm_f.state_dict().items()})
    arrs.update({"grad.layers." + k: npf(p_.grad) for k, p_ in layers.named_parameters()})
    arrs.update({"grad.norm_f." + k: npf(p_.grad) for k, p_ in norm_f.named_parameters()})
    save(name, x=npf(x), y=npf(y), g=npf(g), dx=npf(x.grad), norm=np.array(norm),
         residual_in_fp32=np.array(int(residual_in_fp32)), n_layers=np.array(n_layers), **arrs)


def load_reference_norm_refs():
    """layer_norm_ref / rms_norm_ref (mamba/mamba_ssm/ops/triton/layernorm.py:19-48) without importing the
    module (its top imports triton, absent here): the two pure-PyTorch function definitions are taken out of
    the reference file's syntax tree and executed as they stand."""
    import ast
    path = f"{REF}/mamba/mamba_ssm/ops/triton/layernorm.py"
    tree = ast.parse(open(path).read())
    keep = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in ("layer_norm_ref", "rms_norm_ref")]
    assert len(keep) == 2
    ns = {"torch": torch, "F": F}
    exec(compile(ast.Module(body=keep, type_ignores=[]), path, "exec"), ns)
    return ns["layer_norm_ref"], ns["rms_norm_ref"]


def load_reference_state_update_ref():
    """selective_state_update_ref (mamba/mamba_ssm/ops/triton/selective_state_update.py:157-192), taken out of
    the file's syntax tree (the module imports triton at the top)."""
    import ast
    from einops import rearrange
    path = f"{REF}/mamba/mamba_ssm/ops/triton/selective_state_update.py"
    tree = ast.parse(open(path).read())
    keep = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "selective_state_update_ref"]
    assert len(keep) == 1
    ns = {"torch": torch, "F": F, "rearrange": rearrange}
    exec(compile(ast.Module(body=keep, type_ignores=[]), path, "exec"), ns)
    return ns["selective_state_update_ref"]


def gen_state_update(ref, name, batch, dim, dstate, has_D, has_z, has_bias, softplus, seed=0):
    torch.manual_seed(seed)
    state = torch.randn(batch, dim, dstate)
    x, dt = torch.randn(batch, dim), 0.5 * torch.rand(batch, dim)
    A = -0.5 * torch.rand(dim, dstate)
    B, C = torch.randn(batch, dstate), torch.randn(batch, dstate)
    D = torch.randn(dim) if has_D else None
    z = torch.randn(batch, dim) if has_z else None
    bias = 0.5 * torch.rand(dim) if has_bias else None
    st = state.clone()
    out = ref(st, x, dt, A, B, C, D=D, z=z, dt_bias=bias, dt_softplus=softplus)
    save(name, state_in=npf(state), state_out=npf(st), x=npf(x), dt=npf(dt), A=npf(A), B=npf(B), C=npf(C),
         D=npf(D), z=npf(z), dt_bias=npf(bias), out=npf(out), softplus=int(softplus))


def gen_norm(refs, name, shape, is_rms, has_residual, has_bias, prenorm, itype=torch.float32, eps=1e-5, seed=0):
    """Fused add + norm semantics = the ref with upcast=True (what the reference's kernels compute)."""
    layer_norm_ref, rms_norm_ref = refs
    torch.manual_seed(seed)
    N = shape[-1]
    x = torch.randn(*shape).to(itype).float().requires_grad_()
    res = torch.randn(*shape).to(itype).float().requires_grad_() if has_residual else None
    w = (1 + 0.5 * torch.randn(N)).requires_grad_()
    b = (0.5 * torch.randn(N)).requires_grad_() if has_bias else None
    fn = rms_norm_ref if is_rms else layer_norm_ref
    out = fn(x, w, b, residual=res, eps=eps, prenorm=prenorm, upcast=True)
    y, pre = (out if prenorm else (out, None))
    g = torch.randn_like(y)
    gpre = torch.randn_like(y) if prenorm else None
    loss = (y * g).sum() + ((pre * gpre).sum() if prenorm else 0)
    loss.backward()
    arrs = dict(x=npf(x), weight=npf(w), y=npf(y), g=npf(g), dx=npf(x.grad), dweight=npf(w.grad),
                is_rms=int(is_rms), prenorm=int(prenorm), eps=eps, itype=str(itype))
    if has_residual:
        arrs.update(residual=npf(res), dresidual=npf(res.grad))
    if has_bias:
        arrs.update(bias=npf(b), dbias=npf(b.grad))
    if prenorm:
        arrs.update(pre=npf(pre), gpre=npf(gpre))
    save(name, **arrs)


def main():
    torch.set_num_threads(8)
    if os.environ.get("GOLDEN_ONLY") == "inner768":  # add the reference-shape inner-function fixtures without touching the others
        sys.path.insert(0, HERE)
        from recipes import INNER768_CASES
        cci, ssi = load_reference()
        load_reference_modules(ssi)
        for name in INNER768_CASES:
            gen_inner768(ssi, name)
        return
    if os.environ.get("GOLDEN_ONLY") == "ssu":
        ref = load_reference_state_update_ref()
        gen_state_update(ref, "ssu_N16_full", 3, 70, 16, True, True, True, True, seed=1)
        gen_state_update(ref, "ssu_N8_plain", 2, 33, 8, False, False, False, False, seed=2)
        gen_state_update(ref, "ssu_N64_noz", 2, 40, 64, True, False, True, True, seed=3)
        gen_state_update(ref, "ssu_N5_odd", 1, 9, 5, True, True, False, False, seed=4)
        return
    if os.environ.get("GOLDEN_ONLY") == "cscan":  # add the complex-A scan fixtures without touching the others
        cci, ssi = load_reference()
        gen_cscan_all(ssi)
        return
    if os.environ.get("GOLDEN_ONLY") == "blocks6":  # round 6: d_state = 4 (the suite's CLIP ViViM, model_clip.py:945-947) and expand = 2 at d_state 16
        cci, ssi = load_reference()
        mods = load_reference_modules(ssi)
        gen_blocks6(mods)
        return
    if os.environ.get("GOLDEN_ONLY") == "stack":  # add the Block-stack fixtures without touching the others
        cci, ssi = load_reference()
        mods = load_reference_modules(ssi)
        gen_stack(mods, "stack_ln", "ln", False, seed=11)
        gen_stack(mods, "stack_rms_fp32res", "rms", True, seed=12)
        return
    if os.environ.get("GOLDEN_ONLY") == "norm":  # add the norm fixtures without touching the others
        refs = load_reference_norm_refs()
        k = 0
        for is_rms in (False, True):
            for shape in ((3, 17, 64), (2, 9, 192), (4, 1000), (2, 5, 1024), (1, 3, 2304)):
                for has_residual, has_bias, prenorm in ((True, True, True), (False, False, False), (True, False, False)):
                    k += 1
                    gen_norm(refs, f"norm_{'rms' if is_rms else 'ln'}_N{shape[-1]}_r{int(has_residual)}b{int(has_bias)}p{int(prenorm)}",
                             shape, is_rms, has_residual, has_bias, prenorm, seed=k)
        gen_norm(refs, "norm_rms_bf16_N384", (2, 33, 384), True, True, False, True, itype=torch.bfloat16, seed=99)
        gen_norm(refs, "norm_ln_bf16_N768", (2, 7, 768), False, True, True, True, itype=torch.bfloat16, seed=98)
        return
    cci, ssi = load_reference()
    print("scan:")
    gen_scan(ssi, "scan_L128_g1", 2, 4, 8, 128)
    gen_scan(ssi, "scan_L372_g2", 2, 4, 8, 372, groups=2)
    gen_scan(ssi, "scan_L1134_plain", 2, 4, 8, 1134, has_D=False, has_z=False, has_bias=False, softplus=False)
    gen_scan(ssi, "scan_L2048_b3", 1, 4, 8, 2048, b3=True)
    gen_scan(ssi, "scan_L4100_long", 1, 2, 4, 4100)
    gen_scan(ssi, "scan_cfg1", 2, 128, 16, 256)  # BASELINE.json configs[0]
    gen_scan(ssi, "scan_constBC", 2, 4, 8, 200, var_B=False, var_C=False)
    gen_scan(ssi, "scan_constB", 2, 4, 8, 200, var_B=False)
    gen_scan(ssi, "scan_constC", 2, 4, 8, 200, var_C=False)
    gen_scan(ssi, "scan_bf16_L300", 2, 4, 16, 300, itype=torch.bfloat16)
    gen_scan(ssi, "scan_f16_L130", 2, 4, 8, 130, itype=torch.float16)
    print("scan, complex A:")
    gen_cscan_all(ssi)
    print("conv:")
    for L in (8, 151, 372):
        for W in (2, 3, 4):
            for has_bias in (False, True):
                for silu in (False, True):
                    gen_conv(cci, f"conv_L{L}_W{W}_b{int(has_bias)}_s{int(silu)}", 2, 6, L, W, has_bias, silu)
    gen_conv(cci, "conv_L1134_W4_b1_s1", 2, 10, 1134, 4, True, True)
    gen_conv(cci, "conv_bf16_L200_W4_b1_s1", 2, 6, 200, 4, True, True, itype=torch.bfloat16)
    for W in (2, 3, 4):
        gen_conv_update(cci, f"convupd_W{W}", 2, 24, W, True, True)
    gen_conv_update(cci, "convupd_W4_plain", 2, 24, 4, False, False)
    print("norm:")
    refs = load_reference_norm_refs()
    k = 0
    for is_rms in (False, True):
        for shape in ((3, 17, 64), (2, 9, 192), (4, 1000), (2, 5, 1024), (1, 3, 2304)):
            for has_residual, has_bias, prenorm in ((True, True, True), (False, False, False), (True, False, False)):
                k += 1
                gen_norm(refs, f"norm_{'rms' if is_rms else 'ln'}_N{shape[-1]}_r{int(has_residual)}b{int(has_bias)}p{int(prenorm)}",
                         shape, is_rms, has_residual, has_bias, prenorm, seed=k)
    gen_norm(refs, "norm_rms_bf16_N384", (2, 33, 384), True, True, False, True, itype=torch.bfloat16, seed=99)
    gen_norm(refs, "norm_ln_bf16_N768", (2, 7, 768), False, True, True, True, itype=torch.bfloat16, seed=98)
    print("state update:")
    ref = load_reference_state_update_ref()
    gen_state_update(ref, "ssu_N16_full", 3, 70, 16, True, True, True, True, seed=1)
    gen_state_update(ref, "ssu_N8_plain", 2, 33, 8, False, False, False, False, seed=2)
    gen_state_update(ref, "ssu_N64_noz", 2, 40, 64, True, False, True, True, seed=3)
    gen_state_update(ref, "ssu_N5_odd", 1, 9, 5, True, True, False, False, seed=4)
    print("inner:")
    mods = load_reference_modules(ssi)
    gen_inner(ssi, "inner_out_proj", "out_proj")
    gen_inner(ssi, "inner_no_out_proj", "no_out_proj")
    gen_inner(ssi, "inner_bi", "bi")
    sys.path.insert(0, HERE)
    from recipes import INNER768_CASES
    for name in INNER768_CASES:
        gen_inner768(ssi, name)
    print("blocks:")
    gen_block(mods, "block_vim", "simple")
    gen_block(mods, "block_vim_div", "simple", if_devide_out=True, L=257, batch=1)
    gen_block(mods, "block_vim_norm", "norm", if_devide_out=True)
    gen_block(mods, "block_dbm", "new", expand=1)
    gen_blocks6(mods)
    print("block stacks:")
    gen_stack(mods, "stack_ln", "ln", False, seed=11)
    gen_stack(mods, "stack_rms_fp32res", "rms", True, seed=12)


if __name__ == "__main__":
    main()
