"""Host time of a step at a size where the GPU work is negligible: cProfile of N eager steps of bench.py's workload.
usage: python tools/host_profile.py [config] [batch,seqlen,d_model] [steps]"""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "video-mamba-suite_amd"))
import torch  # noqa: E402
import bench  # noqa: E402

config = sys.argv[1] if len(sys.argv) > 1 else "block"
dims = tuple(int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "1,64,768").split(","))
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 300
dev = torch.device("cuda", 0)
torch.manual_seed(0)
block, b, l, d_model = bench.make_workload(config, dev, dims)
hidden = torch.randn(b, l, d_model, device=dev, dtype=torch.bfloat16, requires_grad=True)
gout = torch.randn(b, l, d_model, device=dev, dtype=torch.bfloat16)
tf = tb = 0.0


def step():
    global tf, tb
    block.zero_grad(set_to_none=True)
    hidden.grad = None
    t0 = time.perf_counter()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out = block(hidden)
    t1 = time.perf_counter()
    out.backward(gout)
    t2 = time.perf_counter()
    tf += t1 - t0
    tb += t2 - t1


for _ in range(30):
    step()
torch.cuda.synchronize()
tf = tb = 0.0
t0 = time.perf_counter()
for _ in range(steps):
    step()
torch.cuda.synchronize()
t = time.perf_counter() - t0
print(f"{config} {dims}: {t / steps * 1e3:.3f} ms per step on the host clock (forward call {tf / steps * 1e3:.3f}, backward call {tb / steps * 1e3:.3f})")
pr = cProfile.Profile()
pr.enable()
for _ in range(steps):
    step()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(35)
if os.environ.get("HOST_PROFILE_OPS", "1") == "1":
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU]) as prof:
        for _ in range(50):
            step()
        torch.cuda.synchronize()
    print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=45, max_name_column_width=60))
