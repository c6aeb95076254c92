"""Random-configuration sweep of the three Mamba modules: the fused fast path under bf16 autocast against the SAME module's fp32 run
on the unfused path (use_fast_path=False: separate conv / projection / scan ops), outputs and every parameter gradient:
    python tools/fuzz_modules.py [n] [seed] [n_stack]
(n_stack: stacks of 1-3 Blocks with the fused add + norm kernels against PyTorch's norm + the unfused mixer in fp32)"""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "video-mamba-suite_amd"))
import torch
from mamba_ssm.modules.mamba_simple import Mamba as ViM
from mamba_ssm.modules.mamba_new import Mamba as DBM
from mamba_ssm.modules.mamba_simple_scan_norm import Mamba as ViMNorm

DEV = "cuda"


def rel(a, r):
    return ((a.double() - r.double()).abs().max() / r.double().abs().max().clamp_min(1e-6)).item()


def case(rng):
    which = rng.choice(["vim", "vim", "dbm", "vim_norm"])   # (bimamba_type "none" is not constructible in the reference either: mamba_simple.py:126)
    d_model = rng.choice([16, 24, 32, 48, 64, 96, 128, 192, 256])
    expand = rng.choice([1, 2, 2])
    d_state = rng.choice([4, 8, 16, 16, 16])
    d_conv = rng.choice([2, 3, 4, 4])
    batch = rng.choice([1, 2, 3, 4])
    L = max(1, rng.choice([rng.randint(1, 20), 8 * rng.randint(1, 60), 16 * rng.randint(1, 30) + 1, rng.randint(20, 500), 197, 785]))
    kw = dict(d_state=d_state, d_conv=d_conv, expand=expand, conv_bias=rng.random() < 0.85, bias=rng.random() < 0.3)
    if rng.random() < 0.3:
        kw["dt_rank"] = rng.choice([1, 3, 8, 17])
    seed = rng.randint(0, 1 << 30)
    torch.manual_seed(seed)
    if which == "dbm":
        mod = DBM(d_model, **kw)
    elif which == "vim_norm":
        mod = ViMNorm(d_model, bimamba_type="v2", if_devide_out=rng.random() < 0.5, **kw)
    else:
        mod = ViM(d_model, bimamba_type="v2", if_devide_out=rng.random() < 0.5, **kw)
    mod = mod.to(DEV)
    desc = f"{which} dm{d_model} e{expand} N{d_state} W{d_conv} b{batch} L{L} {kw} seed{seed}"
    x = torch.randn(batch, L, d_model, device=DEV)
    gout = torch.randn(batch, L, d_model, device=DEV)

    def run(fast, autocast):
        mod.use_fast_path = fast
        mod.zero_grad(set_to_none=True)
        h = x.clone().requires_grad_()
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
            out = mod(h)
        out.float().backward(gout)
        return out.detach().float(), h.grad.float(), {n: p.grad.float().clone() for n, p in mod.named_parameters() if p.grad is not None}
    ref = run(which == "dbm", False)   # the DBM block has no unfused path (mamba_new.py:216): its fp32 fast path is the reference there
    bad = []
    for name, (fast, ac, tol) in dict(fp32_fast=(True, False, 3e-3), bf16_fast=(True, True, 5e-2)).items():
        got = run(fast, ac)
        if not torch.isfinite(got[0]).all():
            bad.append(f"{name} out not finite")
            continue
        for what, a, r in (("out", got[0], ref[0]), ("dx", got[1], ref[1])):
            e = rel(a, r)
            if not e <= tol:
                bad.append(f"{name} {what} {e:.2e}")
        for n in ref[2]:
            if n not in got[2]:
                bad.append(f"{name} d{n} missing")
                continue
            e = rel(got[2][n], ref[2][n])
            if not e <= 2 * tol:
                bad.append(f"{name} d{n} {e:.2e}")
    return desc, bad


def stack_case(rng):
    """a stack of 1-3 Blocks (Add -> Norm -> Mixer, mamba_simple.py:381-437) with the fused add + LayerNorm / RMSNorm kernels under bf16
    autocast against the same stack on PyTorch's norm and the unfused mixer path in fp32"""
    from functools import partial
    from mamba_ssm.modules.mamba_simple import Block
    from mamba_ssm.ops.triton.layernorm import RMSNorm
    d_model = rng.choice([32, 48, 64, 96, 128, 192])
    depth = rng.choice([1, 2, 3])
    rms = rng.random() < 0.5
    res32 = rng.random() < 0.6
    batch = rng.choice([1, 2, 3])
    L = max(1, rng.choice([rng.randint(1, 40), 8 * rng.randint(1, 40), 16 * rng.randint(1, 20) + 1, 197]))
    kw = dict(d_state=rng.choice([8, 16]), d_conv=rng.choice([3, 4]), expand=rng.choice([1, 2]), bimamba_type="v2")
    seed = rng.randint(0, 1 << 30)
    torch.manual_seed(seed)
    norm_cls = partial(RMSNorm, eps=1e-5) if rms else partial(torch.nn.LayerNorm, eps=1e-5)
    blocks = torch.nn.ModuleList([Block(d_model, partial(ViM, **kw), norm_cls=norm_cls, fused_add_norm=True, residual_in_fp32=res32)
                                  for _ in range(depth)]).to(DEV)
    desc = f"stack dm{d_model} depth{depth} {'rms' if rms else 'ln'} res32={int(res32)} b{batch} L{L} {kw} seed{seed}"
    if os.environ.get("FUZZ_VERBOSE"):
        print("RUN", desc, flush=True)
    if desc.split(" seed")[0] in os.environ.get("FUZZ_SKIP", "").split(";"):
        return desc, []
    x = torch.randn(batch, L, d_model, device=DEV)
    gout = torch.randn(batch, L, d_model, device=DEV)

    def run(fused, autocast):
        for blk in blocks:
            blk.fused_add_norm = fused
            blk.mixer.use_fast_path = fused
        blocks.zero_grad(set_to_none=True)
        h, res = x.clone().requires_grad_(), None
        h0 = h
        verbose = bool(os.environ.get("FUZZ_VERBOSE"))
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
            for i, blk in enumerate(blocks):
                h, res = blk(h, res)
                if verbose:
                    torch.cuda.synchronize(); print("   fwd block", i, "fused" if fused else "unfused", flush=True)
            out = h.float() + res.float()
        if verbose:
            hooks = []
            for i, blk in enumerate(blocks):
                def mk(i):
                    def hook(mod, gi, go):
                        torch.cuda.synchronize(); print("   bwd through block", i, flush=True)
                    return hook
                hooks.append(blk.register_full_backward_hook(mk(i)))
        out.backward(gout)
        if verbose:
            torch.cuda.synchronize(); print("   bwd done", flush=True)
            for hk in hooks:
                hk.remove()
        return out.detach(), h0.grad.float(), {n: p.grad.float().clone() for n, p in blocks.named_parameters() if p.grad is not None}
    ref = run(False, False)
    got = run(True, True)
    bad = []
    tol = 6e-2
    if not torch.isfinite(got[0]).all():
        return desc, ["out not finite"]
    for what, a, r in (("out", got[0], ref[0]), ("dx", got[1], ref[1])):
        e = rel(a, r)
        if not e <= tol:
            bad.append(f"{what} {e:.2e}")
    for n in ref[2]:
        if n not in got[2]:
            bad.append(f"d{n} missing")
        elif not rel(got[2][n], ref[2][n]) <= 2 * tol:
            bad.append(f"d{n} {rel(got[2][n], ref[2][n]):.2e}")
    return desc, bad


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    n_stack = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    fails = 0
    for i in range(n + n_stack):
        try:
            desc, bad = case(rng) if i < n else stack_case(rng)
        except Exception as e:
            import traceback
            desc, bad = f"case {i}", [f"EXCEPTION {type(e).__name__}: {str(e)[:300]}", traceback.format_exc()[-700:]]
        if bad:
            fails += 1
            print("FAIL", desc, "::", "; ".join(bad), flush=True)
    print(f"{n} module + {n_stack} block-stack cases, {fails} failed")
    sys.exit(1 if fails else 0)


if __name__ == "__main__":
    main()
