"""bench.py -- headline benchmark (BASELINE.json): Mamba-block forward+backward tokens/s at
(B, L, D, d_state) = (8, 8192, 1024, 16) on N GPUs of one node, with the roofline of the dominant
hot-path kernel and the CPU baseline (the reference's selective_scan_ref path on the host cores) beside it.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--config block|stack|dbm|long] [--graph]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
      bench.py --gpus N --steps K --warmup W
--gpus N must be the job's world size: under torch.distributed.run a mismatch is an error; `python bench.py --gpus N` with
N > 1 and no launcher re-launches itself under torch.distributed.run (ensure_world below).

A step = one pass of the hot path over one synthetic batch.  --config (default block = the judged line):
  block  configs[1]: the ViM ("v2") Mamba block exactly as the suite instantiates it (timemamba.py:115,
         blocks.py:910: d_model=1024, expand=1 -> d_inner=1024, d_conv=4, d_state=16), B=8 per GPU, L=8192
  stack  configs[2]: 12 x Block(Add -> RMSNorm -> ViM) at d_model 768, (8, 3136) per GPU, fused add+norm kernels
  dbm    configs[3]: the DBM block (mamba_new.py) at d_model 512, (2, 2304) per GPU
  long   configs[4]: the ViM block at d_model 768, (1, 65536) per GPU (sequence-split scans)
all under autocast(bf16), forward + backward (both scan directions, both conv directions, the in/x/dt/out
projection GEMMs, and -- for N>1 -- DDP's bucketed RCCL all-reduce of the gradients).  Inputs are resident in
HBM before the timed region.  Per-GPU batch is fixed (weak scaling); value = N * B * L * K / max-over-ranks wall
time.
"""
import argparse
import json
import os
import platform
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "video-mamba-suite_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch
import torch.distributed as dist

D_STATE, D_CONV = 16, 4
# name -> (what, per-GPU batch, seqlen, d_model, expand, layers)
WORKLOADS = {
    "block": ("configs[1]: single ViM (bimamba v2) Mamba block", 8, 8192, 1024, 1, 1),
    "stack": ("configs[2]: 12 x Block(Add -> RMSNorm -> ViM), fused add+norm", 8, 3136, 768, 1, 12),
    "dbm": ("configs[3]: DBM bidirectional Mamba block (shared weights, fwd + reversed scan)", 2, 2304, 512, 1, 1),
    "long": ("configs[4]: ViM block, long-video regime (sequence-split scans)", 1, 65536, 768, 1, 1),
    # not a BASELINE config: the suite's other regime -- TimeMamba's scans along time (timemamba.py:135-140): 8 clips x 196 tokens,
    # 8 frames each (the lane-per-row scan kernels, the folded small projections; profiles/r04_short_rows.md)
    "frames": ("extra: ViM block on short sequences with many rows (TimeMamba, 'b (n t) d -> (b n) t d')", 1568, 8, 768, 1, 1),
    # the secondary lines SURVEY.md section 8 wrote down (round 6; attached to the judged line as extra_configs):
    #   section 8 "Decision recorded for the builder": the headline block with the module's default expand=2 (d_inner 2048)
    "block_expand2": ("secondary line: configs[1]'s ViM block with expand=2 (the module default)", 8, 8192, 1024, 2, 1),
    #   8(d) configs[2]: the ViViM-S-like stack, 24 x (d_model 384, expand 2) on 16 frames x 197 tokens (vivim.py:406-423, 555-556)
    "vivim_s": ("configs[2] secondary: ViViM-S-like 24 x Block(Add -> RMSNorm -> ViM), fused add+norm", 8, 3152, 384, 2, 24),
    #   8(d) configs[3]: the temporal-action-localization backbone's 7 DBM mixers (arch (2, 2, 5), backbones.py:282-288, blocks.py:899-942):
    #   2 stem blocks at 2304, 5 branch blocks at 2304 / 1152 / 576 / 288 / 144 (each followed by the stride-2 max pool), LayerNorm + residual
    "dbm_pyramid": ("configs[3] secondary: the TAL backbone's DBM pyramid, 7 x (LayerNorm -> DBM -> residual [-> maxpool/2]) at L = 2304 ... 144", 2, 2304, 512, 1, 7),
}
B, L, D_MODEL, EXPAND = WORKLOADS["block"][1:5]
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
HBM_ACHIEVABLE_GBS = 6290.0  # the same guide: what a float4 copy kernel reaches (79 %): the ceiling a streaming kernel can be held to
# DDP bucket cap (MB).  A bucket closes once it HOLDS >= the cap, in the order the gradients become ready: out_proj's weight
# gradient (4 MB at d_model 1024) closes the first bucket by itself and its all-reduce runs under the inner node's backward; the
# inner node's small parameters and in_proj's weight gradient (8 MB) follow in the last.  (32 MB, the earlier setting, put the
# block's ~14 MB into ONE bucket: the all-reduce started after the last gradient -- VERDICT r3.)  VMS_DDP_BUCKET_MB overrides.
DDP_BUCKET_MB = 4
# HBM bytes per call from the PMC passes of the same kernels at the config's size (FETCH_SIZE x2 + WRITE_SIZE, separate
# rocprofv3 --pmc runs, main and carry kernels added up per entry point: tools/measure_cfg.sh, tools/pmc_table.py); a profile of
# the committed build, not a live measurement
TRAFFIC_PROFILE = os.path.join(ROOT, "profiles", "r06_{config}_traffic.json")


def profiled_traffic(kernel, config="block", x_layout=3):
    """x_layout 1 = the 128-element checkpoint layout (the `block_coarse_checkpoints` extra line): its scans move ~1 GB less per
    launch than the default layout's, so it has its own profile (profiles/r06_block_coarse_traffic.json); no profile -> None."""
    path = TRAFFIC_PROFILE.format(config=config + ("_coarse" if x_layout == 1 else ""))
    if not os.path.exists(path) and x_layout != 1:   # a config this round did not re-profile: the previous round's profile of it
        path = path.replace("r06_", "r05_")
    try:
        with open(path) as f:
            t = json.load(f)
        if kernel in t:
            return t[kernel]["hbm_bytes"], os.path.relpath(path, ROOT)
        if kernel == "vms_selective_scan_bwd_dual":   # a pair that ran as two launches (small grids): two single calls
            return 2 * t["vms_selective_scan_bwd"]["hbm_bytes"], os.path.relpath(path, ROOT) + " (2 x the single call)"
    except (OSError, KeyError, ValueError):
        pass
    return None, None


def algorithmic_bytes(batch=B, dim=D_MODEL * EXPAND, seqlen=L, n=D_STATE, s=2, groups=1, w=D_CONV, bwd_out_z=False):
    """SURVEY.md 8(d) / BASELINE.md section 2: algorithmic HBM bytes per launch of each kernel.
    bwd_out_z: the backward scan also rewrites the gated output (SURVEY's 9 B D L s: u, delta, dout, z, out read; du,
    ddelta, dz, out_z written).  The blocks' fused nodes have no reader for it (no fused out_proj) and do not ask for
    it since round 3 -- 8 B D L s; the roofline fraction is quoted on the bytes the launch really owes."""
    n_c = (seqlen + 2047) // 2048
    bdl = batch * dim * seqlen
    bc = 2 * batch * groups * n * seqlen
    x = batch * dim * n_c * 2 * n * 4
    bwd_small = bc * s + x + bc * 4 + (2 * dim * n + 4 * dim) * 4
    return {
        "vms_selective_scan_fwd": 5 * bdl * s + bc * s + x + (dim * n + 2 * dim) * 4,
        "vms_selective_scan_bwd": (9 if bwd_out_z else 8) * bdl * s + bwd_small,
        # both directions of a bidirectional block in one call (vms_hip.h ABI v9): per direction u, delta, dout, z read and du,
        # ddelta written (6 B D L s each), both pre-gate outputs read and dz written ONCE (3 B D L s) -- 15 where two single
        # calls owe 16 (17 with the second call's read of the first's dz); SURVEY 8d's figure also rewrites out_z per direction
        "vms_selective_scan_bwd_dual": (17 if bwd_out_z else 15) * bdl * s + 2 * bwd_small,
        "vms_causal_conv1d_fwd": 2 * bdl * s + dim * (w + 1) * 4,
        "vms_causal_conv1d_bwd": 3 * bdl * s + 2 * dim * (w + 1) * 4,
        # both directions' conv1d of a bidirectional block in one pass: x read once, two outputs written
        "vms_causal_conv1d_fwd_dual": 3 * bdl * s + 2 * dim * (w + 1) * 4,
        # the fused backward tail (SSI:278-283 in one pass): x and du read, dx written (+ dx read when it
        # accumulates the other direction's gradient: not counted), dx_dbl (R + 2N rows) read, the weights and fp32 accumulators
        "vms_proj_conv_bwd": 3 * bdl * s + batch * (-(-dim // 16) + 2 * n) * seqlen * s + (dim * (w + 1) + (-(-dim // 16) + 2 * n) * dim) * 4,
    }


# The scans are bound by vector-ALU issue, not by HBM (profiles/r02_sq_scan_*.md).  Their VALU floor:
# cycles one SIMD needs per 64 (element, state) pairs when every instruction issues at its standalone rate --
# v_exp_f32 8.3 cycles (quarter rate), any other fp32 lane-operation 2.0 (packed or not: 32 lanes per clock) --
#   forward : 1 exp + 5 lane-ops (delta*A, delta*u*B, local recurrence, seeded recurrence, y += C x)     = 18.3 cycles
#   backward: 1.25 exp + 15 lane-ops (a, b, c, x re-scan + x chain, g chain, g*a*x, S1, S2, dA, dB, dC ..) = 40.4 cycles
# on 1,024 SIMDs at the 2.4 GHz peak clock.  valu_frac = valu_floor_us / measured: the binding resource's own fraction.
VALU_CYCLES_PER_64 = {"vms_selective_scan_fwd": 8.3 + 5 * 2.0, "vms_selective_scan_bwd": 1.25 * 8.3 + 15 * 2.0,
                      "vms_selective_scan_bwd_dual": 2 * (1.25 * 8.3 + 15 * 2.0)}   # two directions per call
N_SIMD, PEAK_CLOCK_HZ = 1024, 2.4e9


def valu_floor_us(kernel, batch, dim, seqlen, n=D_STATE):
    cyc = VALU_CYCLES_PER_64.get(kernel)
    if cyc is None:
        return None
    return batch * dim * seqlen * n / 64.0 * cyc / N_SIMD / PEAK_CLOCK_HZ * 1e6


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return platform.processor() or "unknown"


def usable_cores():
    """Host cores this process may actually run on: the scheduler affinity mask, capped by the container's CPU quota
    (os.cpu_count() reports the machine -- 256 on the GPU boxes -- whatever the cgroup allows; oversubscribing a
    small problem with 256 intra-op threads made it 1000x slower)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                quota, period = txt[0], float(txt[1])
            else:
                quota, period = txt[0], float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota not in ("max", "-1"):
                n = min(n, max(1, int(float(quota) / period + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, n)


def pin_rank_to_cores(local_rank, local_world):
    """One process per GPU on ONE node: give each rank its own slice of the cores this job may use and size torch's
    intra-op pool to it, so that N ranks do not each start a pool as wide as the machine (the GPU boxes report 256 cores,
    a container may own 16: 8 ranks x 16 threads on 16 cores made the launch-bound configs host-bound).  Honours an
    OMP_NUM_THREADS the launcher set.  Returns what it did (reported in the bench line)."""
    try:
        cores = sorted(os.sched_getaffinity(0))
    except AttributeError:
        cores = list(range(os.cpu_count() or 1))
    usable = min(len(cores), usable_cores())
    cores = cores[:usable] if usable < len(cores) else cores
    per = max(1, len(cores) // max(1, local_world))
    mine = cores[local_rank * per:(local_rank + 1) * per] if local_world > 1 and len(cores) >= local_world else cores
    info = {"cores_visible": len(cores), "cores_this_rank": len(mine), "pinned": False}
    if local_world > 1 and mine and os.environ.get("VMS_BENCH_NO_PIN") != "1":
        try:
            os.sched_setaffinity(0, mine)
            info["pinned"] = True
        except (AttributeError, OSError):
            pass
    threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or max(1, min(len(mine), 8))
    torch.set_num_threads(threads)
    info["torch_threads"] = threads
    return info


def _median_time(fn, reps=5, warm=1, budget_s=12.0):
    """median of `reps` runs after `warm` warm-ups; stops early (>= 1 run) when the budget is spent"""
    t_start = time.perf_counter()
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
        if time.perf_counter() - t_start > budget_s:
            break
    return statistics.median(ts), len(ts)


def cpu_baseline_reference_path(reps=5, threads=None, shape=(2, 256, 128, 16), budget_s=8.0):
    """north_star / BASELINE.md section 3: the reference's pure-PyTorch CPU path -- selective_scan_ref
    (mamba/mamba_ssm/ops/selective_scan_interface.py:86-152) + causal_conv1d_ref
    (causal-conv1d/causal_conv1d/causal_conv1d_interface.py:49-65), as restated in this repo's mamba_ssm / causal_conv1d
    packages and pinned to the reference by tests/golden -- on the host cores, fp32, input recipe of
    test_selective_scan.py:53-88; median of `reps` after one warm-up.  shape = (B, L, D, N): BASELINE configs[0]
    (2, 256, 128, 16) by default; the bench line's `value` uses a bounded sample of the headline rows instead."""
    from causal_conv1d.causal_conv1d_interface import causal_conv1d_ref
    from mamba_ssm.ops.selective_scan_interface import selective_scan_ref
    cores = usable_cores() if threads is None else threads
    torch.set_num_threads(cores)
    b, l, d, n = shape
    g = torch.Generator().manual_seed(0)
    r = lambda *s: torch.randn(*s, generator=g)
    x0 = r(b, d, l).requires_grad_()
    w, cb = r(d, D_CONV).requires_grad_(), r(d).requires_grad_()
    delta = (0.5 * torch.rand(b, d, l, generator=g)).requires_grad_()
    A = (-0.5 * torch.rand(d, n, generator=g)).requires_grad_()
    Bm, Cm = r(b, 1, n, l).requires_grad_(), r(b, 1, n, l).requires_grad_()
    Dv, z = r(d).requires_grad_(), r(b, d, l).requires_grad_()
    bias = (0.5 * torch.rand(d, generator=g)).requires_grad_()
    gout = r(b, d, l)
    leaves = [x0, w, cb, delta, A, Bm, Cm, Dv, z, bias]

    def fwd():
        u = causal_conv1d_ref(x0, w, cb, activation="silu")
        return selective_scan_ref(u, delta, A, Bm, Cm, Dv, z=z, delta_bias=bias, delta_softplus=True)

    def fwd_only():
        with torch.no_grad():
            fwd()

    def fwd_bwd():
        for t in leaves:
            t.grad = None
        fwd().backward(gout)
    t_f, _ = _median_time(fwd_only, reps, budget_s=budget_s / 2)
    t_fb, n_fb = _median_time(fwd_bwd, reps, budget_s=budget_s)
    return {"fwd_ms": t_f * 1e3, "fwd_bwd_ms": t_fb * 1e3, "fwd_tokens_per_s": b * l / t_f,
            "fwd_bwd_tokens_per_s": b * l / t_fb, "shape": [b, l, d, n], "dtype": "f32", "reps": n_fb,
            "threads": cores}


def cpu_baseline_c_port(seconds_budget=25.0, min_reps=5):
    """Second, labelled CPU line: the oracle port (oracle/vms_oracle.c, f32 arithmetic, OpenMP over rows) on a
    bounded sample of the headline workload: scan + conv forward and backward of ONE direction for one sequence of
    2048 tokens at all 1024 channels (the projection GEMMs are not part of the oracle); median of >= 5."""
    import numpy as np
    from oracle import oracle as orc
    orc.build()
    cores = usable_cores()
    os.environ["OMP_NUM_THREADS"] = str(cores)   # read by libgomp when the oracle library is first loaded
    rng = np.random.default_rng(0)
    d, n = D_MODEL * EXPAND, D_STATE
    sb, sl = 1, 2048
    f = lambda *shape: rng.standard_normal(shape, dtype=np.float32)
    u, z, g = f(sb, d, sl), f(sb, d, sl), f(sb, d, sl)
    delta = 0.5 * rng.random((sb, d, sl), dtype=np.float32)
    A = -0.5 * rng.random((d, n), dtype=np.float32)
    Bm, Cm = f(sb, 1, n, sl), f(sb, 1, n, sl)
    Dv, bias = f(d), 0.5 * rng.random(d, dtype=np.float32)
    w, cb = f(d, D_CONV), f(d)

    def once():
        x = orc.conv_fwd(u, w, cb, True)
        orc.scan_fwd(x, delta, A, Bm, Cm, Dv, z, bias, True)
        orc.scan_bwd(x, delta, A, Bm, Cm, Dv, z, bias, g, True)
        orc.conv_bwd(u, w, cb, g, True)
    once()
    ts, t_start = [], time.perf_counter()
    while len(ts) < min_reps or (time.perf_counter() - t_start < seconds_budget and len(ts) < 9):
        t0 = time.perf_counter()
        once()
        ts.append(time.perf_counter() - t0)
        if time.perf_counter() - t_start > 2 * seconds_budget:
            break
    t = statistics.median(ts)
    return {"tokens_per_s_one_direction": sb * sl / t, "ms": t * 1e3, "reps": len(ts), "threads": cores,
            "shape": [sb, sl, d, n], "what": "oracle C port (f32, OpenMP): conv fwd + scan fwd + scan bwd + conv bwd"}


def cpu_baseline(threads_restore=None):
    """cpu_baseline object of the bench line.  value = the reference's CPU path (pure-PyTorch selective_scan_ref +
    causal_conv1d_ref, forward + backward of one scan direction) on a BOUNDED SAMPLE OF THE HEADLINE WORKLOAD -- one
    sequence, the first 1024 of its 8192 tokens, all 1024 channels, d_state 16 (the recurrence is sequential in L, so
    tokens/s does not depend on how many tokens are sampled) -- halved for the block's two directions.  The same path at
    BASELINE configs[0] (2, 256, 128, 16) and the oracle's C port follow as labelled extras (`configs0_*`, `c_port`);
    neither is meant to be divided into `value` of the bench line.  The projection GEMMs are in none of them."""
    port = cpu_baseline_c_port()            # first: libgomp reads OMP_NUM_THREADS when the library is loaded
    hd = (1, 1024, D_MODEL * EXPAND, D_STATE)
    head = cpu_baseline_reference_path(reps=3, shape=hd, budget_s=14.0)
    head1 = cpu_baseline_reference_path(reps=3, threads=1, shape=hd, budget_s=14.0) if head["threads"] > 1 else head
    best = max((head, head1), key=lambda r: r["fwd_bwd_tokens_per_s"])
    ref = cpu_baseline_reference_path()     # configs[0], all usable cores
    ref1 = cpu_baseline_reference_path(threads=1) if ref["threads"] > 1 else ref
    if threads_restore:
        torch.set_num_threads(threads_restore)
    cores = best["threads"]
    return {
        "value": best["fwd_bwd_tokens_per_s"] / 2.0, "unit": "tokens/s", "cores": cores, "kind": "port",
        "cpu": cpu_model(), "host_cores_usable": usable_cores(), "host_cores_reported": os.cpu_count(),
        "sample": f"selective_scan_ref + causal_conv1d_ref (pure PyTorch, torch.set_num_threads({cores}), fp32) forward + "
                  f"backward on a bounded sample of the headline workload: (B,L,D,N)=({hd[0]},{hd[1]},{hd[2]},{hd[3]}) = one "
                  f"sequence's first {hd[1]} tokens at all {hd[2]} channels: median of {best['reps']} = "
                  f"{best['fwd_bwd_ms']:.1f} ms (forward only {best['fwd_ms']:.1f} ms); the better of {head['threads']} "
                  "and 1 thread(s); one direction, halved for the block's two; projection GEMMs not included",
        "headline_sample_all_cores": head, "headline_sample_1_thread": head1,
        "configs0_reference_path_all_cores": ref, "configs0_reference_path_1_thread": ref1,
        "c_port": dict(port, tokens_per_s_block=port["tokens_per_s_one_direction"] / 2.0),
    }


def ddp_buckets(model, cap_mb):
    """What the reducer really built (after its first-iteration rebuild in gradient-ready order): count and sizes."""
    try:
        d = model._get_ddp_logging_data()
        sizes = [int(v) for v in str(d.get("rebuilt_bucket_sizes") or d.get("bucket_sizes") or "").replace(",", " ").split()]
    except Exception:  # noqa: BLE001
        sizes = []
    return {"bucket_cap_mb": cap_mb, "n_buckets": len(sizes) or None, "bucket_bytes": sizes or None}


MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16, MI355X_MICROARCH.md


def projection_mfma(block, hidden, iters=20):
    """The block's two large projections (library GEMMs on the matrix cores, mamba_ssm/ops/projections.py) timed on
    their own AFTER the timed region: forward + backward of in_proj and of out_proj, as TFLOP/s against the dense
    bf16 MFMA peak (north_star: "MFMA utilisation for the projections"); the MFMA busy counters of the same GEMMs are
    in profiles/ (tools/mfma_counters.sh)."""
    from mamba_ssm.ops.projections import in_proj_fn, out_proj_fn
    out = {}
    with torch.autocast("cuda", dtype=torch.bfloat16):
        cases = {
            "in_proj": (lambda x: in_proj_fn(x, block.in_proj.weight, block.in_proj.bias),
                        hidden.detach().clone().requires_grad_(), block.in_proj.weight),
            "out_proj": (lambda y: out_proj_fn(y, block.out_proj.weight, block.out_proj.bias),
                         # y in the layout the scans leave it in (channel-slowest, strides (L, B L, 1)): the path of the step
                         torch.randn(block.out_proj.weight.shape[1], hidden.shape[0], hidden.shape[1], device=hidden.device,
                                     dtype=torch.bfloat16).permute(1, 0, 2).requires_grad_(),
                         block.out_proj.weight),
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


class _Stack(torch.nn.Module):
    """configs[2]: n Blocks (Add -> RMSNorm -> ViM mixer, fused add+norm kernels, fp32 residual stream) closed by the
    fused add + norm_f, the way the suite's ViViM / TimeMamba backbones run them."""

    def __init__(self, d_model, n_layers, expand):
        super().__init__()
        from functools import partial
        from mamba_ssm.modules.mamba_simple import Block, Mamba
        from mamba_ssm.ops.triton.layernorm import RMSNorm
        mixer = partial(Mamba, d_state=D_STATE, d_conv=D_CONV, expand=expand, bimamba_type="v2")
        self.layers = torch.nn.ModuleList([Block(d_model, mixer, norm_cls=partial(RMSNorm, eps=1e-5), fused_add_norm=True,
                                                 residual_in_fp32=True) for _ in range(n_layers)])
        self.norm_f = RMSNorm(d_model, eps=1e-5)

    def forward(self, h):
        from mamba_ssm.ops.triton.layernorm import rms_norm_fn
        res = None
        for blk in self.layers:
            h, res = blk(h, res)
        return rms_norm_fn(h, self.norm_f.weight, self.norm_f.bias, eps=self.norm_f.eps, residual=res, prenorm=False,
                           residual_in_fp32=True)


class _DbmPyramid(torch.nn.Module):
    """configs[3] secondary line: the DBM mixers of the temporal-action-localization backbone (MambaBackbone, arch (2, 2, 5):
    backbones.py:239-330; MaskMambaBlock: blocks.py:899-942) -- per block LayerNorm -> DBM(expand=1) -> residual, the five branch
    blocks each followed by the stride-2 max pool (kernel 3, padding 1), so the mixers see L, L, L, L/2, L/4, L/8, L/16.  (The
    mask multiply and the stochastic depth of the task code are identities on full-length eval-style inputs and left out.)"""

    def __init__(self, d_model, n_stem=2, n_branch=5):
        super().__init__()
        from mamba_ssm.modules.mamba_new import Mamba as DBM
        n = n_stem + n_branch
        self.norms = torch.nn.ModuleList([torch.nn.LayerNorm(d_model) for _ in range(n)])
        self.mixers = torch.nn.ModuleList([DBM(d_model, d_state=D_STATE, d_conv=D_CONV, expand=1) for _ in range(n)])
        self.n_stem = n_stem

    def forward(self, h):                         # (B, L, d_model) -> the coarsest level (B, L / 32, d_model)
        for i, (norm, mixer) in enumerate(zip(self.norms, self.mixers)):
            h = h + mixer(norm(h))
            if i >= self.n_stem:
                h = torch.nn.functional.max_pool1d(h.transpose(1, 2), 3, 2, 1).transpose(1, 2)
        return h


def make_workload(config, device, dims=None):
    """-> (module, per-GPU batch, seqlen, d_model).  dims = (batch, seqlen, d_model) overrides the sizes (CPU tests)."""
    what, b, l, d_model, expand, layers = WORKLOADS[config]
    if dims is not None:
        b, l, d_model = dims
    if config == "dbm":
        from mamba_ssm.modules.mamba_new import Mamba as DBM
        m = DBM(d_model, d_state=D_STATE, d_conv=D_CONV, expand=expand)
    elif config == "dbm_pyramid":
        m = _DbmPyramid(d_model) if dims is None else _DbmPyramid(d_model, 1, 1)
    elif layers > 1:
        m = _Stack(d_model, layers if dims is None else 2, expand)
    else:
        from mamba_ssm.modules.mamba_simple import Mamba
        m = Mamba(d_model, d_state=D_STATE, d_conv=D_CONV, expand=expand, bimamba_type="v2")
    return m.to(device), b, l, d_model


def run(config="block", steps=50, warmup=20, device=None, backend="nccl", dims=None, autocast=True,
        cpu_base=True, projections=True, graph=False, x_policy=None):
    """The timed loop.  device=None: cuda:LOCAL_RANK; a CPU device (tests: gloo + checker-backed fake extensions)
    runs the same DDP / timing / reporting code on tiny sizes and skips the GPU-only instrumentation.
    Returns the result dict on rank 0, None elsewhere."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # VMS_BENCH_DDP_WORLD1=1 (tests, one-GPU boxes): the DDP / RCCL path also at world size 1
    distributed = world > 1 or os.environ.get("VMS_BENCH_DDP_WORLD1") == "1"
    on_gpu = device is None or torch.device(device).type == "cuda"
    if on_gpu:
        assert torch.cuda.is_available(), "bench.py needs a GPU (the HIP path has no CPU fallback)"
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    else:
        dev = torch.device(device)
    if distributed and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if on_gpu:
            dist.init_process_group(backend=backend, device_id=dev)  # "nccl" = RCCL over xGMI
        else:
            dist.init_process_group(backend=backend)
    if distributed:   # --gpus N is a contract (ensure_world): the process group really has that many ranks
        assert dist.get_world_size() == world, f"process group has {dist.get_world_size()} ranks, WORLD_SIZE says {world}"
    sync = torch.cuda.synchronize if on_gpu else (lambda: None)
    host = pin_rank_to_cores(local_rank, int(os.environ.get("LOCAL_WORLD_SIZE", world))) if on_gpu else None

    import vms_hip
    if x_policy is not None:   # checkpoint layout of the forward scans for this run ("coarse" = 128-element, what a training
        vms_hip.set_x_layout_policy(x_policy)   # job that fills the device's memory gets from the default "auto" policy)

    torch.manual_seed(0 + rank)
    block, b, l, d_model = make_workload(config, dev, dims)
    model = block
    # --graph under torch.distributed: the BARE module is captured and the gradients leave as one flat all-reduce per replay
    # (mamba_ssm/utils/hip_graph.py); DDP's hooks are host callbacks and cannot be replayed
    use_graph = bool(graph) and on_gpu
    bucket_mb = float(os.environ.get("VMS_DDP_BUCKET_MB", DDP_BUCKET_MB))
    if distributed and not use_graph:
        kw = dict(device_ids=[local_rank]) if on_gpu else {}
        model = torch.nn.parallel.DistributedDataParallel(block, bucket_cap_mb=bucket_mb, gradient_as_bucket_view=True, **kw)
    act_dtype = torch.bfloat16 if autocast else torch.float32
    # the block sits inside a network: its input gradient is part of the backward
    hidden = torch.randn(b, l, d_model, device=dev, dtype=act_dtype, requires_grad=True)
    # fixed upstream gradient: the step is exactly the block's forward + backward (no loss kernels)
    out_len = l
    if config == "dbm_pyramid":     # every branch block halves the length (max pool, kernel 3, stride 2, padding 1)
        for _ in range(len(block.mixers) - block.n_stem):
            out_len = (out_len - 1) // 2 + 1
    gout = torch.randn(b, out_len, d_model, device=dev, dtype=act_dtype)

    def step():
        model.zero_grad(set_to_none=True)
        hidden.grad = None
        with torch.autocast(dev.type, dtype=torch.bfloat16, enabled=autocast):
            out = model(hidden)
        out.backward(gout)

    # --graph: the same step recorded once as a HIP graph and replayed with one launch per step (+ one flat all-reduce per
    # replay when distributed) -- what the launch-bound shapes (configs[3]) need; kernels are then timed in eager steps
    if use_graph:
        from mamba_ssm.utils.hip_graph import GraphedStep
        gs = GraphedStep(block, hidden, autocast_dtype=torch.bfloat16 if autocast else None,
                         process_group=dist.group.WORLD if distributed else None,
                         allreduce=os.environ.get("VMS_GRAPH_ALLREDUCE", "after"))
        gs.gout.copy_(gout)
        eager_step, step = step, gs.replay

    # the GPU's clocks need ~20 steps (0.1 s) to settle: with fewer the first timed steps run 1-2 % slow.  The extra
    # untimed steps below are reported as config.clock_ramp_steps; the timed region is exactly --steps steps.
    ramp = max(0, 20 - warmup) if on_gpu else 0
    # Per-kernel durations: two event records per C-ABI launch on the launch stream.  Every record is a marker in the stream:
    # ~6 us of idle GPU on either side of a timed launch (rocprofv3 trace of this file, profiles/r03z_step_trace.txt: the
    # library's own kernels run back to back, every timed one sits between two gaps) = 0.1 ms per block step with all ten
    # launches timed.  So: the headline block times EVERY entry point during `kt_pre` of its untimed warm-up steps, picks the
    # dominant kernel from that, and inside the timed region times only that one (the contract's live roofline measurement:
    # 2 launches per step).  The other configs are launch-bound on the host (configs[3]: ~50 launches in < 1 ms): their kernels
    # are timed in `kt_steps` extra steps AFTER the timed region (reported as config.kernel_timing).
    inline_timing = on_gpu and config == "block" and not use_graph
    kt_pre = min(5, warmup + ramp) if inline_timing else 0
    for _ in range(warmup + ramp - kt_pre):
        step()
    pre_ms, dom_pre = {}, None
    if kt_pre:
        sync()
        vms_hip.start_timing(reserve=kt_pre * 16 * WORKLOADS[config][5])
        for _ in range(kt_pre):
            step()
        pre_ms = vms_hip.stop_timing()
        ab_names = algorithmic_bytes()
        dom_pre = max((k for k in pre_ms if k in ab_names), key=lambda k: sum(pre_ms[k]), default=None)
    if distributed:
        dist.barrier()
    sync()
    if inline_timing and dom_pre:
        vms_hip.start_timing(reserve=steps * 4 * WORKLOADS[config][5], only=dom_pre)   # events pre-created here
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    sync()
    if distributed:
        dist.barrier()
    sync()
    elapsed = time.perf_counter() - t0
    kernel_ms, kernel_steps = {}, {}
    kt_steps = steps
    if inline_timing:
        kernel_ms = dict(pre_ms)
        kernel_steps = {k: kt_pre for k in pre_ms}
        if dom_pre:
            live = vms_hip.stop_timing()
            if live.get(dom_pre):
                kernel_ms[dom_pre], kernel_steps[dom_pre] = live[dom_pre], steps
    if on_gpu and not inline_timing:
        kt_steps = min(steps, 10)
        if use_graph:
            step = eager_step   # event records cannot be replayed: the kernels are timed in eager steps
        vms_hip.start_timing(reserve=kt_steps * 16 * WORKLOADS[config][5])
        for _ in range(kt_steps):
            step()
        kernel_ms = vms_hip.stop_timing()
    if distributed:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()

    res = None
    if rank == 0:
        what = WORKLOADS[config][0]
        ms_per_step = elapsed / steps * 1e3
        tokens = world * b * l * steps
        d_inner = d_model * WORKLOADS[config][4]
        # the DBM block runs its two halves as ONE node on a batch of 2 b (vms_hip.h reverse_from) unless vms_hip.debug.dbm_two_nodes;
        # every other config's scans cover (b, d_inner, l) per launch
        scan_b = 2 * b if config in ("dbm", "dbm_pyramid") and not __import__("vms_hip").debug.dbm_two_nodes else b
        # (the pyramid's launches cover five different lengths: per-kernel times only, no per-launch byte count)
        ab = algorithmic_bytes(batch=scan_b, dim=d_inner, seqlen=l) if config != "dbm_pyramid" else {}
        kern = {}
        for name, ts in kernel_ms.items():
            avg = sum(ts) / len(ts)
            ks = kernel_steps.get(name, kt_steps)
            kern[name] = {"calls_per_step": len(ts) / ks, "avg_ms": avg, "ms_per_step": sum(ts) / ks}
            if inline_timing:
                kern[name]["timed_in"] = "the timed region" if ks == steps and name == dom_pre else f"{ks} untimed warm-up steps"
            if name in ab:
                kern[name]["algorithmic_GBs"] = ab[name] / (avg * 1e-3) / 1e9
                kern[name]["hbm_frac"] = kern[name]["algorithmic_GBs"] / HBM_PEAK_GBS
            vf = valu_floor_us(name, scan_b, d_inner, l) if ab else None
            if vf is not None:
                kern[name]["valu_floor_us"] = vf
                kern[name]["valu_frac"] = vf / (avg * 1e3)
        comm = {"backend": backend if distributed else None, "world_size": dist.get_world_size() if distributed else 1}
        if distributed:
            comm.update(ddp_buckets(model, bucket_mb) if not use_graph else
                        {"gradient_exchange": f"one flat all-reduce per replay ({gs.allreduce} the graph), "
                                              f"{sum(t.numel() * t.element_size() for t in gs.flat.values())} bytes"})
        if on_gpu:
            try:
                comm["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
            except Exception:  # noqa: BLE001
                comm["rccl_version"] = None
        res = {
            "metric": (f"Mamba-block fwd+bwd tokens/s at (B,L,D,d_state)=({b},{l},{d_model},{D_STATE}); % HBM roofline"
                       if WORKLOADS[config][5] == 1 else
                       f"{WORKLOADS[config][5]}-layer {'DBM pyramid' if config == 'dbm_pyramid' else 'ViM stack'} fwd+bwd tokens/s at "
                       f"(B,L,D,d_state)=({b},{l},{d_model},{D_STATE}); % HBM roofline"),
            "value": tokens / elapsed, "unit": "tokens/s", "n_gpus": world, "steps": steps,
            "warmup": warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16" if autocast else "f32", "data": "synthetic",
            "config": {"workload": f"{what}: fwd+bwd, autocast {'bf16' if autocast else 'off'}, B={b} per GPU, L={l}, d_model={d_model}, "
                                   f"expand={WORKLOADS[config][4]} (d_inner={d_inner}), d_state={D_STATE}, d_conv={D_CONV}",
                       "name": config,
                       "step": "fwd+bwd" + (" + DDP all-reduce" if distributed else ""),
                       "global_batch": world * b, "seq_len": l, "parallelism": f"dp{world}",
                       "input_grad": True, "clock_ramp_steps": ramp,
                       "hip_graph": use_graph,   # --graph: the timed steps are replays of one captured graph
                       "kernel_timing": (f"{dom_pre} inside the timed region (every launch of it), the other entry points in {kt_pre} of the "
                                         "untimed warm-up steps" if inline_timing else f"{kt_steps} extra steps after the timed region"),
                       "checkpoint_lvl": int(os.environ.get("VMS_CHECKPOINT_LVL", "0")),
                       # vms_hip.h x_has_sub: 3 = the forward scan leaves the state after every 8 elements for the backward
                       # scan (8 B D L more bytes per launch of either, NOT counted in the algorithmic bytes below)
                       "x_layout": 1 if (os.environ.get("VMS_X_LAYOUT") == "1" or x_policy == "coarse") else 3, "comm": comm, "host": host},
            "kernels": kern,
        }
        if any(k in ab for k in kern):
            dom = max((k for k in kern if k in ab), key=lambda k: kern[k]["ms_per_step"])
            traffic, src = profiled_traffic(dom, config, res["config"]["x_layout"])
            res["roofline"] = {"kernel": dom, "bound": "hbm", "binding_resource": "valu" if "valu_frac" in kern[dom] else "hbm",
                               "valu_frac": kern[dom].get("valu_frac"), "achieved": kern[dom]["algorithmic_GBs"],
                               "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": kern[dom]["hbm_frac"], "traffic": traffic,
                               "peak_achievable": HBM_ACHIEVABLE_GBS, "frac_achievable": kern[dom]["algorithmic_GBs"] / HBM_ACHIEVABLE_GBS,
                               "traffic_source": (f"{src}: rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE of this kernel at this "
                                                  "size, committed profile (not measured in this run)") if src else None,
                               "algorithmic_bytes_per_launch": ab[dom], "avg_launch_ms": kern[dom]["avg_ms"]}
            if dom.startswith("vms_selective_scan") and res["config"]["x_layout"] == 3:
                # the design's own extra traffic (8-element checkpoints, fp32): in `traffic`, not in `achieved`
                res["roofline"]["checkpoint_bytes_per_launch"] = scan_b * d_inner * (l // 8) * D_STATE * 4 * (2 if dom.endswith("_dual") else 1)
            if dom.endswith("_dual"):
                res["roofline"]["launch"] = "one call = the backward scans of BOTH directions of the block (vms_selective_scan_bwd_dual)"
            if dom in ("vms_selective_scan_bwd", "vms_selective_scan_bwd_dual"):   # SURVEY 8d also counts an out_z rewrite (9 B D L s per direction) the blocks' nodes never ask for
                ab8 = algorithmic_bytes(batch=scan_b, dim=d_inner, seqlen=l, bwd_out_z=True)[dom]
                res["roofline"]["algorithmic_bytes_survey_8d"] = ab8
                res["roofline"]["frac_survey_8d"] = ab8 / (kern[dom]["avg_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS
        if on_gpu and world == 1 and projections and config == "block":
            res["projections"] = projection_mfma(block, hidden)
        if world == 1 and cpu_base:
            res["cpu_baseline"] = cpu_baseline(threads_restore=host["torch_threads"] if host else None)
    return res


def ensure_world(gpus, argv):
    """--gpus N is a contract, not a comment: the job must BE N ranks.
    * WORLD_SIZE set (launched by torch.distributed.run / torchrun): it must equal N, else SystemExit.
    * WORLD_SIZE unset and N > 1 (`python bench.py --gpus 8`): re-launch this very command line under
      `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free>` and
      return its exit code (the ranks print the one JSON line); VMS_BENCH_NO_SELF_LAUNCH=1 fails loudly instead.
    * N == 1 without WORLD_SIZE: run here.  -> None when the caller should go on in this process."""
    ws = os.environ.get("WORLD_SIZE")
    if ws is not None:
        if int(ws) != gpus:
            raise SystemExit(f"bench.py: --gpus {gpus} but WORLD_SIZE={ws}: launch with --nproc-per-node {gpus} "
                             f"(or pass --gpus {ws})")
        return None
    if gpus <= 1:
        return None
    if os.environ.get("VMS_BENCH_NO_SELF_LAUNCH") == "1":
        raise SystemExit(f"bench.py: --gpus {gpus} needs one process per GPU: python -m torch.distributed.run --nnodes=1 "
                         f"--nproc-per-node {gpus} --master-addr 127.0.0.1 --master-port P bench.py --gpus {gpus} ...")
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port)] + list(argv)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    print(f"bench.py: --gpus {gpus} without WORLD_SIZE: re-launching as {' '.join(cmd)}", file=sys.stderr, flush=True)
    return subprocess.call(cmd, env=env)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", choices=sorted(WORKLOADS), default="block")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-configs", action="store_true", help="only the judged block (skip the stack / long / dbm lines attached as extra_configs)")
    ap.add_argument("--extra-steps", type=int, default=10, help="timed steps of each extra config")
    ap.add_argument("--no-projections", action="store_true", help="skip the separate projection GEMM timing (profiling runs)")
    ap.add_argument("--graph", action="store_true", default=None, help="replay the step as one HIP graph (launch-bound shapes); "
                    "under torch.distributed the gradients leave as one flat all-reduce per replay.  Default: on for --config dbm "
                    "(0.25 ms of kernels behind 0.8 ms of host work), off elsewhere")
    ap.add_argument("--no-graph", dest="graph", action="store_false", help="eager steps (DDP hooks) also for --config dbm")
    # tests only (tests/test_ddp_gloo.py): the same command line on CPU / gloo with tiny sizes
    ap.add_argument("--device", default=None, help=argparse.SUPPRESS)
    ap.add_argument("--backend", default="nccl", help=argparse.SUPPRESS)
    ap.add_argument("--dims", default=None, help=argparse.SUPPRESS)
    args = ap.parse_args(argv)
    rc = ensure_world(args.gpus, sys.argv if argv is None else [sys.argv[0]] + list(argv))
    if rc is not None:
        return rc
    dims = tuple(int(v) for v in args.dims.split(",")) if args.dims else None
    if args.graph is None:
        args.graph = args.config == "dbm" and args.device is None
    res = run(args.config, args.steps, args.warmup, device=args.device, backend=args.backend, dims=dims,
              autocast=args.device is None, cpu_base=not args.no_cpu_baseline and args.device is None,
              projections=not args.no_projections and args.device is None, graph=args.graph)
    # The other BASELINE configs, driver-observed: AFTER the judged line's timed region and outside it, a few steps
    # each of configs[2] / [3] / [4] (and of the judged block with the 128-element checkpoint layout a memory-filling training job
    # gets) on the same ranks; attached to the same JSON line, whose metric / config / value stay the judged block's.
    if args.config == "block" and not args.no_extra_configs and args.device is None:
        extra = {}
        # each extra run is guarded: an out-of-memory stack, a failed graph capture or a collective hiccup in one of them
        # is recorded under its name and the judged line -- already measured -- is still printed
        for name, kw in (("stack", {}), ("long", {}), ("dbm", {"graph": True}), ("block_coarse_checkpoints", {"x_policy": "coarse"}),
                         ("block_expand2", {}), ("vivim_s", {}), ("dbm_pyramid", {"graph": True})):
            try:
                torch.cuda.empty_cache()
                r = run("block" if name == "block_coarse_checkpoints" else name, args.extra_steps, 5, backend=args.backend, cpu_base=False,
                        projections=False, **kw)
            except Exception as e:  # noqa: BLE001
                r = None
                if res is not None:
                    extra[name] = {"error": f"{type(e).__name__}: {e}"[:500]}
                try:
                    torch.cuda.synchronize()
                except Exception:  # noqa: BLE001
                    pass
            finally:
                if kw.get("x_policy"):
                    __import__("vms_hip").set_x_layout_policy("auto")
            if r is not None:
                rf = r.get("roofline", {})
                extra[name] = {"workload": r["config"]["workload"], "ms_per_step": r["ms_per_step"], "tokens_per_s": r["value"],
                               "steps": r["steps"], "hip_graph": r["config"]["hip_graph"], "x_layout": r["config"]["x_layout"],
                               "roofline": {k: rf.get(k) for k in ("kernel", "frac", "frac_achievable", "achieved", "traffic", "avg_launch_ms",
                                                                  "algorithmic_bytes_per_launch", "valu_frac")},
                               "kernels_ms_per_step": {k: round(v["ms_per_step"], 4) for k, v in r.get("kernels", {}).items()}}
        if res is not None:
            res["extra_configs"] = extra
    if res is not None:
        print(json.dumps(res), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
