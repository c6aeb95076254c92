import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "video-mamba-suite_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name + ".npz")) as f:
        return {k: f[k] for k in f.files}


def golden_names(prefix):
    return sorted(f[:-4] for f in os.listdir(GOLDEN) if f.startswith(prefix) and f.endswith(".npz"))


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as orc
    orc.build()
    return orc


def build_stack(g, device="cpu", fused_add_norm=False, use_fast_path=True):
    """The Block stack of tests/golden/stack_*.npz (reference Block + ViM mixer + final norm_f) out of THIS repo's
    classes, with the fixture's state dict loaded.  -> (layers, norm_f, run) where run(x) -> y."""
    from functools import partial
    import torch
    from mamba_ssm.modules.mamba_simple import Block, Mamba
    from mamba_ssm.ops.triton.layernorm import RMSNorm, layer_norm_fn, rms_norm_fn
    norm, res32, n = str(g["norm"]), bool(g["residual_in_fp32"]), int(g["n_layers"])
    d_model = g["x"].shape[-1]
    norm_cls = partial(torch.nn.LayerNorm, eps=1e-5) if norm == "ln" else partial(RMSNorm, eps=1e-5)
    mixer_cls = partial(Mamba, d_state=8, d_conv=4, expand=2, bimamba_type="v2", use_fast_path=use_fast_path)
    layers = torch.nn.ModuleList([Block(d_model, mixer_cls, norm_cls=norm_cls, fused_add_norm=fused_add_norm,
                                        residual_in_fp32=res32) for _ in range(n)])
    norm_f = norm_cls(d_model)
    layers.load_state_dict({k[len("sd.layers."):]: torch.tensor(v) for k, v in g.items() if k.startswith("sd.layers.")})
    norm_f.load_state_dict({k[len("sd.norm_f."):]: torch.tensor(v) for k, v in g.items() if k.startswith("sd.norm_f.")})
    layers, norm_f = layers.to(device), norm_f.to(device)

    def run(x):
        h, res = x, None
        for blk in layers:
            h, res = blk(h, res)
        if fused_add_norm:  # the backbones' fused closing (e.g. videomamba forward_features): add + norm_f in one op
            fn = rms_norm_fn if isinstance(norm_f, RMSNorm) else layer_norm_fn
            return fn(h, norm_f.weight, norm_f.bias, eps=norm_f.eps, residual=res, prenorm=False, residual_in_fp32=res32)
        return norm_f((h + res).to(norm_f.weight.dtype))
    return layers, norm_f, run
