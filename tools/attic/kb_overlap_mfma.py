"""Round-3 overlap probe: does matrix-pipe work co-execute with the VALU-bound scans on MI355X?

  part 1  scan (bwd / fwd) on one stream, an in_proj-sized bf16 library GEMM (65536 x 1024 @ 1024 x 2048, hipBLASLt through
          torch) on another: serial vs two streams, both launch orders.
  part 2  the same scans beside tools/mfma_burner.hip -- a 4-wave, ~136-register, barrier-free workgroup per CU that only
          issues v_mfma_f32_32x32x16_bf16 (variants: + one ds_read_b128 per MFMA, + global loads): what a GEMM WRITTEN to
          fit beside a scan workgroup could get, and what it costs the scan.

Every figure is event-timed on the stream the kernel runs on; run the script under `rocprofv3 --kernel-trace` for the
start / end timestamps (tools/overlap_trace.py condenses them).   usage: python tools/kb_overlap_mfma.py [quick]
"""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-mamba-suite_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from kb_dual import problem, bwd, fwd

LIB = os.path.join(ROOT, "tools", "build", "libmfma_burner.so")


def burner_lib():
    if not os.path.exists(LIB):
        os.makedirs(os.path.dirname(LIB), exist_ok=True)
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-shared", "-fPIC",
                               os.path.join(ROOT, "tools", "mfma_burner.hip"), "-o", LIB])
    lib = ctypes.CDLL(LIB)
    lib.burner_launch.restype = ctypes.c_int
    lib.burner_launch.argtypes = [ctypes.c_int] * 4 + [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    return lib


def ev():
    return torch.cuda.Event(enable_timing=True)


def timed_pair(first, second, s1, s2, reps=5):
    """first() on s1, second() on s2 launched right behind it; returns per-kernel and total times (us), median of reps."""
    rows = []
    for _ in range(reps):
        torch.cuda.synchronize()
        a0, a1, b0, b1 = ev(), ev(), ev(), ev()
        with torch.cuda.stream(s1):
            a0.record(); first(); a1.record()
        with torch.cuda.stream(s2):
            b0.record(); second(); b1.record()
        torch.cuda.synchronize()
        t0 = min(0.0, a0.elapsed_time(b0))          # b0 relative to a0 (ms)
        total = max(a0.elapsed_time(a1), a0.elapsed_time(b1)) - t0
        rows.append((a0.elapsed_time(a1) * 1e3, b0.elapsed_time(b1) * 1e3, total * 1e3, a0.elapsed_time(b0) * 1e3))
    rows.sort(key=lambda r: r[2])
    return rows[len(rows) // 2]


def alone(fn, stream, reps=7):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        e0, e1 = ev(), ev()
        with torch.cuda.stream(stream):
            e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


def main():
    quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
    dev = "cuda"
    p = problem(0)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    x = torch.randn(65536, 1024, device=dev, dtype=torch.bfloat16)
    w = torch.randn(2048, 1024, device=dev, dtype=torch.bfloat16) * 0.03
    gemm_out = torch.empty(2048, 65536, device=dev, dtype=torch.bfloat16)
    gemm_flop = 2.0 * 65536 * 1024 * 2048

    def gemm():
        torch.matmul(w, x.t(), out=gemm_out)      # the in_proj formulation of ops/projections.py

    def warm():
        for _ in range(6):
            bwd(p, False); fwd(p, False); gemm()
        torch.cuda.synchronize()
    warm()

    print("== part 1: scan beside an in_proj-sized hipBLASLt GEMM (275 GFLOP) ==")
    for name, op in () if quick else (("scan_bwd", lambda: bwd(p, False)), ("scan_fwd", lambda: fwd(p, False))):
        warm()
        ts, tg = alone(op, s1), alone(gemm, s2)
        a = timed_pair(op, gemm, s1, s2)
        b = timed_pair(gemm, op, s2, s1)
        print(f"{name}: alone {ts:7.1f} us | gemm alone {tg:7.1f} us ({gemm_flop / tg * 1e-6:6.1f} TFLOP/s) | serial sum {ts + tg:7.1f}")
        print(f"   scan first : scan {a[0]:7.1f}  gemm {a[1]:7.1f}  total {a[2]:7.1f} us  (gemm start +{a[3]:6.1f} us)  total / serial = {a[2] / (ts + tg):.3f}")
        print(f"   gemm first : gemm {b[0]:7.1f}  scan {b[1]:7.1f}  total {b[2]:7.1f} us  (scan start +{b[3]:6.1f} us)  total / serial = {b[2] / (ts + tg):.3f}")

    print("== part 2: scan beside the MFMA burner (256 workgroups x 4 waves, one per CU) ==")
    lib = burner_lib()
    sink = torch.zeros(1 << 22, device=dev)
    src = torch.randn(64 << 20, device=dev)            # 256 MB: the burner's variant 2 streams it
    flop_per_mfma = 2.0 * 32 * 32 * 16
    # 16 KB: fits beside a backward-scan workgroup (56 KB) and beside ONE forward-scan workgroup (64 KB; two of those and
    # the burner's ~136 registers per lane do not fit a SIMD).  (A first run with 100 KB -- to pin one burner workgroup
    # per CU -- kept the scans' workgroups out altogether: r03a in profiles/r03_overlap.md.)
    lds = int(os.environ.get("BURNER_LDS_KB", "16")) * 1024

    def burner_clock():
        """effective shader clock (GHz) of the last burner launch, median over its workgroups: s_memtime cycles / wall"""
        torch.cuda.synchronize()
        t = sink[:1024].view(torch.int64).view(256, 2).cpu().double()
        return float((t[:, 0] / (t[:, 1] * 10.0)).median())   # wall_clock64 ticks at 100 MHz

    def burner(variant, iters, s):
        n = lib.burner_launch(variant, 256, iters, lds, sink.data_ptr(), src.data_ptr(), src.numel() // 4, s.cuda_stream)
        assert n > 0
        return n

    for variant, vname in ((0, "mfma only, 4 accumulators"), (3, "mfma only, 1 accumulator (dependent chain)"),
                           (4, "mfma + s_sleep 1"), (5, "mfma + s_sleep 2"), (1, "+ ds_read_b128 per mfma"),
                           (2, "+ ds_read + global loads")):
        if quick and variant in (2, 5):
            continue
        iters = {0: 30000, 3: 60000, 4: 12000, 5: 8000, 1: 20000, 2: 8000}[variant]
        tb = alone(lambda: burner(variant, iters, s2), s2, reps=3)
        n_mfma = (1 if variant == 3 else 4) * iters
        rate = lambda t: n_mfma * 4 * 256 * flop_per_mfma / t * 1e-6
        print(f"-- burner [{vname}] alone: {tb:8.1f} us = {rate(tb):7.1f} TFLOP/s ({n_mfma * 32 / (tb * 1e-6) / 1e9:4.2f} GHz-equivalent at 32 cyc/MFMA; shader clock {burner_clock():4.2f} GHz)")
        for name, op in (("scan_bwd", lambda: bwd(p, False)), ("scan_fwd", lambda: fwd(p, False))):
            warm()
            ts = alone(op, s1)
            # burner first (it is resident on every CU when the scan's workgroups arrive); the scan runs inside its window
            r = timed_pair(lambda: burner(variant, iters, s2), op, s2, s1, reps=3)
            # MFMA rate while the scan ran ~ work the burner finished during the scan's window, from its slowdown
            extra = r[0] - tb
            clk = burner_clock()
            print(f"   {name}: alone {ts:7.1f} us | beside burner {r[1]:7.1f} us ({r[1] / ts:5.3f}x) | burner {tb:8.1f} -> {r[0]:8.1f} us (+{extra:6.1f})"
                  f" | burner rate inside the scan window ~ {max(0.0, 1.0 - extra / r[1]) * 100:5.1f} % of its own | total/serial {r[2] / (ts + tb):.3f} | clock {clk:4.2f} GHz")
            # scan first, burner behind it
            r2 = timed_pair(op, lambda: burner(variant, iters, s2), s1, s2, reps=3)
            print(f"   {name} first: scan {r2[0]:7.1f} us ({r2[0] / ts:5.3f}x) | burner {r2[1]:8.1f} us | total/serial {r2[2] / (ts + tb):.3f} | clock {burner_clock():4.2f} GHz")


if __name__ == "__main__":
    main()
