import os, sys, time, torch
sys.path.insert(0, "video-mamba-suite_amd")
from mamba_ssm.modules.mamba_new import Mamba as DBM
from mamba_ssm.modules.mamba_simple import Mamba as ViM
def run(name, block, B, L, dm, steps=100):
    block = block.cuda()
    x = torch.randn(B, L, dm, device="cuda", dtype=torch.bfloat16, requires_grad=True)
    g = torch.randn(B, L, dm, device="cuda", dtype=torch.bfloat16)
    def step():
        block.zero_grad(set_to_none=True); x.grad = None
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = block(x)
        y.backward(g)
    for _ in range(20): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): step()
    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / steps * 1e3
    # host-only time: how long the CPU needs to ENQUEUE a step (GPU far behind on a long queue is fine here: measure enqueue of 20 steps then sync)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): step()
    th = (time.perf_counter() - t0) / 20 * 1e3
    torch.cuda.synchronize()
    print(f"{name:40s} inner_ext={'off' if os.environ.get('VMS_NO_INNER_EXT')=='1' else 'on '}  step {t:6.3f} ms   host enqueue {th:6.3f} ms", flush=True)
torch.manual_seed(0)
run("DBM (2, 2304, 512)", DBM(512, expand=1), 2, 2304, 512)
run("ViM (8, 1569, 768)", ViM(768, expand=1, bimamba_type="v2"), 8, 1569, 768)
run("ViM (2, 1568, 384)", ViM(384, expand=1, bimamba_type="v2"), 2, 1568, 384)
