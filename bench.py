"""bench.py -- headline benchmark (BASELINE.json): Mamba-block forward+backward tokens/s at
(B, L, D, d_state) = (8, 8192, 1024, 16) on N GPUs of one node, with the roofline of the dominant
hot-path kernel and the CPU baseline (the oracle port on the host cores) beside it.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
      bench.py --gpus N --steps K --warmup W

A step = one pass of the hot path over one synthetic batch: the ViM ("v2") Mamba block exactly as the
suite instantiates it (timemamba.py:115, blocks.py:910: d_model=1024, expand=1 -> d_inner=1024,
d_conv=4, d_state=16) under autocast(bf16), forward + backward (both scan directions, both conv
directions, the in/x/dt/out projection GEMMs, and -- for N>1 -- DDP's bucketed RCCL all-reduce of
the gradients).  Inputs are resident in HBM before the timed region.  Per-GPU batch is fixed
(weak scaling); value = N * B * L * K / max-over-ranks wall time.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "video-mamba-suite_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch
import torch.distributed as dist

B, L, D_MODEL, D_STATE, EXPAND, D_CONV = 8, 8192, 1024, 16, 1, 4
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
# HBM bytes per launch from the PMC passes of the same kernels at the same size (FETCH_SIZE x2 + WRITE_SIZE,
# separate rocprofv3 --pmc runs, tools/traffic.py); a profile of the committed build, not a live measurement
TRAFFIC_PROFILE = os.path.join(ROOT, "profiles", "r01h_traffic.json")


def profiled_traffic(kernel):
    try:
        with open(TRAFFIC_PROFILE) as f:
            return json.load(f)[kernel]["hbm_bytes"]
    except (OSError, KeyError, ValueError):
        return None


def algorithmic_bytes(batch=B, dim=D_MODEL * EXPAND, seqlen=L, n=D_STATE, s=2, groups=1, w=D_CONV):
    """SURVEY.md 8(d) / BASELINE.md section 2: algorithmic HBM bytes per launch of each kernel."""
    n_c = (seqlen + 2047) // 2048
    bdl = batch * dim * seqlen
    bc = 2 * batch * groups * n * seqlen
    x = batch * dim * n_c * 2 * n * 4
    return {
        "vms_selective_scan_fwd": 5 * bdl * s + bc * s + x + (dim * n + 2 * dim) * 4,
        "vms_selective_scan_bwd": 9 * bdl * s + bc * s + x + bc * 4 + (2 * dim * n + 4 * dim) * 4,
        "vms_causal_conv1d_fwd": 2 * bdl * s + dim * (w + 1) * 4,
        "vms_causal_conv1d_bwd": 3 * bdl * s + 2 * dim * (w + 1) * 4,
    }


def cpu_baseline(seconds_budget=25.0):
    """The oracle port (oracle/vms_oracle.c, f32 arithmetic, OpenMP over rows) timed on this host's
    cores on a bounded sample of the same workload: the scan + conv forward and backward of ONE
    direction for a slice of the batch (the projection GEMMs are not part of the oracle)."""
    import numpy as np
    from oracle import oracle as orc
    orc.build()
    cores = os.cpu_count() or 1
    os.environ.setdefault("OMP_NUM_THREADS", str(cores))
    rng = np.random.default_rng(0)
    d, n = D_MODEL * EXPAND, D_STATE
    sb, sl = 1, 2048  # sample: 1 sequence of 2048 tokens (of 8 x 8192), all 1024 channels
    f = lambda *shape: rng.standard_normal(shape, dtype=np.float32)
    u, z, g = f(sb, d, sl), f(sb, d, sl), f(sb, d, sl)
    delta = 0.5 * rng.random((sb, d, sl), dtype=np.float32)
    A = -0.5 * rng.random((d, n), dtype=np.float32)
    Bm, Cm = f(sb, 1, n, sl), f(sb, 1, n, sl)
    Dv, bias = f(d), 0.5 * rng.random(d, dtype=np.float32)
    w, cb = f(d, D_CONV), f(d)
    t0 = time.perf_counter()
    reps = 0
    while True:
        x = orc.conv_fwd(u, w, cb, True)
        orc.scan_fwd(x, delta, A, Bm, Cm, Dv, z, bias, True)
        orc.scan_bwd(x, delta, A, Bm, Cm, Dv, z, bias, g, True)
        orc.conv_bwd(u, w, cb, g, True)
        reps += 1
        el = time.perf_counter() - t0
        if el > seconds_budget or reps >= 3:
            break
    tok_s = reps * sb * sl / el / 2.0  # the block runs two directions per token
    return {"value": tok_s, "unit": "tokens/s", "cores": cores, "kind": "port",
            "sample": f"oracle C port (f32, OpenMP {cores} threads): conv fwd + scan fwd + scan bwd + conv bwd of one "
                      f"direction on ({sb}, {sl}, {d}, {n}) x{reps}, halved for the block's two directions; "
                      "projection GEMMs not included"}


MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16, MI355X_MICROARCH.md


def projection_mfma(block, hidden, iters=20):
    """The block's two large projections (library GEMMs on the matrix cores, mamba_ssm/ops/projections.py) timed on
    their own AFTER the timed region: forward + backward of in_proj and of out_proj, as TFLOP/s against the dense
    bf16 MFMA peak (north_star: "MFMA utilisation for the projections")."""
    from mamba_ssm.ops.projections import in_proj_fn, out_proj_fn
    out = {}
    with torch.autocast("cuda", dtype=torch.bfloat16):
        cases = {
            "in_proj": (lambda x: in_proj_fn(x, block.in_proj.weight, block.in_proj.bias),
                        hidden.detach().clone().requires_grad_(), block.in_proj.weight),
            "out_proj": (lambda y: out_proj_fn(y, block.out_proj.weight, block.out_proj.bias),
                         torch.randn(hidden.shape[0], block.d_inner, hidden.shape[1], device=hidden.device,
                                     dtype=torch.bfloat16, requires_grad=True), block.out_proj.weight),
        }
        for name, (fn, x, w) in cases.items():
            y = fn(x)
            g = torch.randn_like(y)

            def fb():
                x.grad = None
                w.grad = None
                fn(x).backward(g)
            for _ in range(3):
                fb()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                fb()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / iters
            flops = 3 * 2.0 * x.shape[0] * hidden.shape[1] * w.shape[0] * w.shape[1]   # fwd + dgrad + wgrad
            out[name] = {"fwd_bwd_ms": ms, "tflops": flops / ms / 1e9,
                         "mfma_frac": flops / ms / 1e9 / MFMA_BF16_PEAK_TFLOPS}
    block.zero_grad(set_to_none=True)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-projections", action="store_true", help="skip the separate projection GEMM timing (profiling runs)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1
    assert torch.cuda.is_available(), "bench.py needs a GPU (the HIP path has no CPU fallback)"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=dev)  # RCCL over xGMI

    import vms_hip
    from mamba_ssm.modules.mamba_simple import Mamba

    torch.manual_seed(0 + rank)
    block = Mamba(D_MODEL, d_state=D_STATE, d_conv=D_CONV, expand=EXPAND, bimamba_type="v2").to(dev)
    model = block
    if distributed:
        model = torch.nn.parallel.DistributedDataParallel(block, device_ids=[local_rank], bucket_cap_mb=32,
                                                          gradient_as_bucket_view=True)
    # the block sits inside a network: its input gradient is part of the backward
    hidden = torch.randn(B, L, D_MODEL, device=dev, dtype=torch.bfloat16, requires_grad=True)
    # fixed upstream gradient: the step is exactly the block's forward + backward (no loss kernels)
    gout = torch.randn(B, L, D_MODEL, device=dev, dtype=torch.bfloat16)

    def step():
        model.zero_grad(set_to_none=True)
        hidden.grad = None
        with torch.autocast("cuda", dtype=torch.bfloat16):
            out = model(hidden)
        out.backward(gout)

    # the GPU's clocks need ~20 steps (0.1 s) to settle: with fewer the first timed steps run 1-2 % slow.  The extra
    # untimed steps below are reported as config.clock_ramp_steps; the timed region is exactly --steps steps.
    ramp = max(0, 20 - args.warmup)
    for _ in range(args.warmup + ramp):
        step()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    vms_hip.start_timing()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    kernel_ms = vms_hip.stop_timing()
    if distributed:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        tokens = world * B * L * args.steps
        ab = algorithmic_bytes()
        kern = {}
        for name, ts in kernel_ms.items():
            avg = sum(ts) / len(ts)
            kern[name] = {"calls_per_step": len(ts) / args.steps, "avg_ms": avg,
                          "ms_per_step": sum(ts) / args.steps}
            if name in ab:
                kern[name]["algorithmic_GBs"] = ab[name] / (avg * 1e-3) / 1e9
                kern[name]["hbm_frac"] = kern[name]["algorithmic_GBs"] / HBM_PEAK_GBS
        dom = max((k for k in kern if k in ab), key=lambda k: kern[k]["ms_per_step"])
        roofline = {"kernel": dom, "bound": "hbm", "achieved": kern[dom]["algorithmic_GBs"], "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": kern[dom]["hbm_frac"], "traffic": profiled_traffic(dom),
                    "algorithmic_bytes_per_launch": ab[dom], "avg_launch_ms": kern[dom]["avg_ms"]}
        res = {
            "metric": "Mamba-block fwd+bwd tokens/s at (B,L,D,d_state)=(8,8192,1024,16); % HBM roofline",
            "value": tokens / elapsed, "unit": "tokens/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "configs[1]: single ViM (bimamba v2) Mamba block fwd+bwd, autocast bf16, "
                                   "B=8 per GPU, L=8192, d_model=1024, expand=1 (d_inner=1024), d_state=16, d_conv=4",
                       "step": "fwd+bwd" + (" + DDP RCCL all-reduce" if distributed else ""),
                       "global_batch": world * B, "seq_len": L, "parallelism": f"dp{world}",
                       "input_grad": True, "clock_ramp_steps": ramp,
                       "checkpoint_lvl": int(os.environ.get("VMS_CHECKPOINT_LVL", "0"))},
            "roofline": roofline,
            "kernels": kern,
        }
        if world == 1 and not args.no_projections:
            res["projections"] = projection_mfma(block, hidden)
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline()
        print(json.dumps(res))
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
