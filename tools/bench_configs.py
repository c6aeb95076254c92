"""Secondary lines: block forward+backward at the shapes of BASELINE configs 3-5 and the suite's class-token length
(one GPU, bf16 autocast, synthetic tokens).  Not the judged metric (bench.py is); shows the same path at other sizes."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "video-mamba-suite_amd"))
from mamba_ssm.modules.mamba_simple import Mamba as ViM
from mamba_ssm.modules.mamba_new import Mamba as DBM

def run(name, block, B, L, d_model, steps=20, warmup=10):
    dev = "cuda"
    block = block.to(dev)
    params = list(block.parameters())
    x = torch.randn(B, L, d_model, device=dev, dtype=torch.bfloat16, requires_grad=True)
    g = torch.randn(B, L, d_model, device=dev, dtype=torch.bfloat16)

    def step():
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = block(x)
        return torch.autograd.grad(y, [x] + params, g)

    def timed(fn):
        for _ in range(warmup): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(steps): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / steps
    eager = timed(step)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3): step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        step()
    replay = timed(graph.replay)
    print(f"{name:62s} eager {eager:7.3f} ms = {B * L / eager / 1e3:6.2f} M tok/s   as one HIP graph {replay:7.3f} ms = "
          f"{B * L / replay / 1e3:6.2f} M tok/s", flush=True)
    del graph


torch.manual_seed(0)
run("cfg 2  ViM block  (8, 8192, d_model 1024, expand 1)", ViM(1024, expand=1, bimamba_type="v2"), 8, 8192, 1024)
run("cfg 2' ViM block  (8, 8192, d_model 1024, expand 2)", ViM(1024, expand=2, bimamba_type="v2"), 8, 8192, 1024)
run("cfg 3  ViM block  (8, 3136, d_model 768, expand 1)", ViM(768, expand=1, bimamba_type="v2"), 8, 3136, 768)
run("       ViM block  (8, 1569 = 8x196+1, d_model 768, expand 1)", ViM(768, expand=1, bimamba_type="v2"), 8, 1569, 768)
run("       ViM block  (8, 3152 = 16x197, d_model 384, expand 2)", ViM(384, expand=2, bimamba_type="v2"), 8, 3152, 384)
run("cfg 4  DBM block  (2, 2304, d_model 512, expand 1)", DBM(512, expand=1), 2, 2304, 512)
run("cfg 5  ViM block  (1, 65536, d_model 768, expand 1)", ViM(768, expand=1, bimamba_type="v2"), 1, 65536, 768)
