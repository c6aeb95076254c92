"""Input recipes shared by the fixture generator (make_golden.py, build container, needs /root/reference) and the tests that
check against the fixtures: the tensors are rebuilt from the seed on both sides (torch's CPU generator), the fixture holds a
checksum of every input next to the expected outputs, so a generator mismatch fails loudly instead of comparing different problems.

inner768: the reference's own inner-function test problem (mamba/tests/ops/test_selective_scan.py:152-199): batch 2, dim 768,
dstate 8, dt_rank 48, seqlen 128, conv width 3, torch.random.manual_seed(0), the tensors drawn in the reference's order; with
is_variable_B / is_variable_C in {False, True} and a real or complex A.  The projection biases (B_proj_bias / C_proj_bias, which the
reference's test leaves None, selective_scan_interface.py:164, 324, 358-359) and the second direction's A_b are drawn AFTER the
reference's tensors so that the common ones keep the reference's values."""
import torch

INNER768 = dict(batch=2, dim=768, dstate=8, dt_rank=48, seqlen=128, width=3)


def inner768_inputs(var_B, var_C, is_complex, proj_bias=False, seed=0):
    c = INNER768
    b, dim, N, R, L = c["batch"], c["dim"], c["dstate"], c["dt_rank"], c["seqlen"]
    wtype = torch.complex64 if is_complex else torch.float32
    torch.random.manual_seed(seed)
    t = {}
    t["xz"] = torch.randn(b, 2 * dim, L)
    t["conv1d_weight"] = torch.randn(dim, 1, c["width"])
    t["conv1d_bias"] = torch.randn(dim)
    t["x_proj_weight"] = torch.randn(R + (int(var_B) + int(var_C)) * N * (2 if is_complex else 1), dim)
    t["delta_proj_weight"] = torch.randn(dim, R)
    t["out_proj_weight"] = torch.randn(dim // 2, dim)
    t["A"] = -0.5 * torch.rand(dim, N, dtype=wtype)
    t["B"] = None if var_B else torch.randn(dim, N, dtype=wtype)
    t["C"] = None if var_C else torch.randn(dim, N, dtype=wtype)
    t["D"] = torch.randn(dim)
    t["delta_bias"] = 0.5 * torch.rand(dim)
    # not part of the reference's test: drawn behind its tensors
    t["A_b"] = -0.5 * torch.rand(dim, N)
    nb = N * (2 if is_complex else 1)
    t["B_proj_bias"] = torch.randn(nb) if proj_bias and var_B else None
    t["C_proj_bias"] = torch.randn(nb) if proj_bias and var_C else None
    t["g_out_proj"] = torch.randn(b, L, dim // 2)   # upstream gradients of the two output layouts
    t["g_no_out_proj"] = torch.randn(b, dim, L)
    return t


def checksum(x):
    """(sum, sum of squares) of a tensor, real and imaginary parts of a complex one as separate values, in float64"""
    x = torch.view_as_real(x) if x.is_complex() else x
    x = x.double()
    return [x.sum().item(), x.square().sum().item()]


def sample(x, n_max=8192):
    """every k-th element of the flattened tensor (k so that at most n_max are kept): what a fixture stores of a large result"""
    x = torch.view_as_real(x) if x.is_complex() else x
    flat = x.reshape(-1)
    k = max(1, -(-flat.numel() // n_max))
    return flat[::k], k


# name -> (function, var_B, var_C, complex A, projection biases)
INNER768_CASES = {}
for _vb in (False, True):
    for _vc in (False, True):
        for _cx in (False, True):   # the reference's grid: mamba_inner_fn, is_variable_B x is_variable_C x wtype
            INNER768_CASES[f"inner768_out_proj_B{int(_vb)}C{int(_vc)}_{'c64' if _cx else 'f32'}"] = ("out_proj", _vb, _vc, _cx, False)
INNER768_CASES.update({
    "inner768_no_out_proj_B0C0_f32": ("no_out_proj", False, False, False, False),
    "inner768_no_out_proj_B0C1_f32": ("no_out_proj", False, True, False, False),
    "inner768_no_out_proj_B1C0_f32": ("no_out_proj", True, False, False, False),
    "inner768_no_out_proj_B1C1_c64": ("no_out_proj", True, True, True, False),
    "inner768_no_out_proj_B0C0_c64": ("no_out_proj", False, False, True, False),
    "inner768_out_proj_B1C1_f32_pbias": ("out_proj", True, True, False, True),
    "inner768_out_proj_B1C1_c64_pbias": ("out_proj", True, True, True, True),
    "inner768_out_proj_B1C0_f32_pbias": ("out_proj", True, False, False, True),
    "inner768_no_out_proj_B1C1_f32_pbias": ("no_out_proj", True, True, False, True),
    "inner768_bi_B1C1_f32": ("bi", True, True, False, False),
    "inner768_bi_B1C1_f32_pbias": ("bi", True, True, False, True),
    "inner768_bi_B0C0_f32": ("bi", False, False, False, False),
})
