"""Random-configuration sweep of the three fused inner functions against an fp64 PyTorch statement of the reference's unfused path on the
CPU (selective_scan_ref + causal_conv1d_ref + the projections; selective_scan_interface.py:155-633 of the reference):
    python tools/fuzz_inner.py [n] [seed]
fp32 operands (tight bar) and bf16 autocast (the modules' regime); every returned gradient compared."""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "video-mamba-suite_amd"))
import torch
import torch.nn.functional as F
from mamba_ssm.ops.selective_scan_interface import (mamba_inner_fn, mamba_inner_fn_no_out_proj, bimamba_inner_fn, selective_scan_ref)
from causal_conv1d.causal_conv1d_interface import causal_conv1d_ref

DEV = "cuda"


def ref_inner(kind, xz, cw, cb, xw, dtw, ow, ob, A, A_b, D, dbias, reverse):
    """fp64, CPU: the reference's unfused composition (mamba_inner_ref / bimamba_inner_ref)"""
    batch, _, L = xz.shape
    R, N = dtw.shape[1], A.shape[1]
    if reverse:
        xz = xz.flip(-1)
    x, z = xz.chunk(2, dim=1)
    x = causal_conv1d_ref(x, cw, cb, "silu")
    x_dbl = F.linear(x.transpose(1, 2).reshape(batch * L, -1), xw)
    delta = (dtw @ x_dbl[:, :R].t()).view(-1, batch, L).permute(1, 0, 2)
    B = x_dbl[:, R:R + N].view(batch, L, N).permute(0, 2, 1).contiguous()
    C = x_dbl[:, -N:].view(batch, L, N).permute(0, 2, 1).contiguous()
    y = selective_scan_ref(x, delta, A, B, C, D, z=z, delta_bias=dbias, delta_softplus=True)
    if kind == "bi":
        y_b = selective_scan_ref(x.flip(-1), delta.flip(-1), A_b, B.flip(-1), C.flip(-1), D, z.flip(-1), dbias, delta_softplus=True)
        y = y + y_b.flip(-1)
    if reverse:
        y = y.flip(-1)
    if kind == "no_out":
        return y
    return F.linear(y.transpose(1, 2), ow, ob)


def case(rng):
    kind = rng.choice(["out", "no_out", "no_out", "bi"])
    d = rng.choice([16, 32, 48, 64, 96, 128, 192, 256])
    N = rng.choice([4, 8, 16, 16, 16])
    R = rng.choice([1, 2, 4, 6, 8, 16, 24, 48])
    W = rng.choice([2, 3, 4, 4])
    batch = rng.choice([1, 2, 3])
    L = max(1, rng.choice([rng.randint(1, 64), 8 * rng.randint(1, 80), 16 * rng.randint(1, 40) + rng.choice([0, 1, 3, 9]), rng.randint(64, 700)]))
    d_model = rng.choice([d // 2, d, 40])
    bf16 = rng.random() < 0.5
    reverse = kind == "no_out" and rng.random() < 0.4
    has_cb, has_ob, has_D = rng.random() < 0.8, rng.random() < 0.5, rng.random() < 0.9
    seed = rng.randint(0, 1 << 30)
    g = torch.Generator().manual_seed(seed)
    R_ = lambda *s, sc=1.0: (sc * torch.randn(*s, generator=g)).double()
    P = dict(xz=R_(batch, 2 * d, L), cw=R_(d, W, sc=0.5), cb=R_(d, sc=0.5) if has_cb else None, xw=R_(R + 2 * N, d, sc=d ** -0.5),
             dtw=R_(d, R, sc=R ** -0.5), ow=R_(d_model, d, sc=d ** -0.5), ob=R_(d_model, sc=0.1) if has_ob else None,
             A=-torch.rand(d, N, generator=g).double() - 0.05, A_b=-torch.rand(d, N, generator=g).double() - 0.05,
             D=R_(d) if has_D else None, dbias=(0.5 * torch.rand(d, generator=g)).double())
    if bf16:   # the values the GPU run sees for its 16-bit activation
        P["xz"] = P["xz"].bfloat16().double()
    desc = f"{kind} b{batch} d{d} N{N} R{R} W{W} L{L} dm{d_model} {'bf16' if bf16 else 'fp32'} rev{int(reverse)} cb{int(has_cb)} ob{int(has_ob)} D{int(has_D)} seed{seed}"
    leaves = {k: v.clone().requires_grad_() for k, v in P.items() if v is not None and not (k == "A_b" and kind != "bi")
              and not (k in ("ow", "ob") and kind == "no_out")}
    get = lambda k: leaves.get(k)
    ref = ref_inner(kind, get("xz"), get("cw"), get("cb"), get("xw"), get("dtw"), get("ow"), get("ob"), get("A"), get("A_b"), get("D"), get("dbias"), reverse)
    gout = torch.randn(ref.shape, generator=g).double()
    ref.backward(gout)
    dv = {k: (v.detach().to(DEV).float() if k != "xz" or not bf16 else v.detach().to(DEV).bfloat16()).requires_grad_() for k, v in leaves.items()}
    gd = lambda k: dv.get(k)
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=bf16):
        cw3 = gd("cw").unsqueeze(1)
        if kind == "no_out":
            out = mamba_inner_fn_no_out_proj(gd("xz"), cw3, gd("cb"), gd("xw"), gd("dtw"), gd("A"), None, None, gd("D"), delta_bias=gd("dbias"),
                                             delta_softplus=True, reverse=reverse, checkpoint_lvl=rng.choice([0, 1]))
        elif kind == "out":
            out = mamba_inner_fn(gd("xz"), cw3, gd("cb"), gd("xw"), gd("dtw"), gd("ow"), gd("ob"), gd("A"), None, None, gd("D"), delta_bias=gd("dbias"),
                                 delta_softplus=True)
        else:
            out = bimamba_inner_fn(gd("xz"), cw3, gd("cb"), gd("xw"), gd("dtw"), gd("ow"), gd("ob"), gd("A"), gd("A_b"), None, None, gd("D"),
                                   delta_bias=gd("dbias"), delta_softplus=True)
    out.float().backward(gout.to(DEV).float().to(out.dtype).float())
    tol = 4e-2 if bf16 else 2e-3
    bad = []

    def cmp(name, a, r, fac=1.0):
        a = a.detach().double().cpu()
        if a.shape != r.shape or not torch.isfinite(a).all():
            bad.append(f"{name} shape/finite")
            return
        e = ((a - r).abs().max() / r.abs().max().clamp_min(1e-6)).item()
        if not e <= tol * fac:
            bad.append(f"{name} {e:.2e}")
    cmp("out", out, ref.detach())
    for k, v in leaves.items():
        if dv[k].grad is None:
            bad.append(f"d{k} missing")
        else:
            cmp("d" + k, dv[k].grad, v.grad, 2.5 if k in ("A", "A_b", "D", "dbias", "cw", "cb", "xw", "dtw", "ow", "ob") else 1.5)
    return desc, bad


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    fails = 0
    for i in range(n):
        try:
            desc, bad = case(rng)
        except Exception as e:
            import traceback
            desc, bad = f"case {i}", [f"EXCEPTION {type(e).__name__}: {str(e)[:300]}", traceback.format_exc()[-600:]]
        if bad:
            fails += 1
            print("FAIL", desc, "::", "; ".join(bad), flush=True)
    print(f"{n} inner-function cases, {fails} failed")
    sys.exit(1 if fails else 0)


if __name__ == "__main__":
    main()
