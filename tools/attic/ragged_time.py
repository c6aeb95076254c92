import os, sys, time, torch
sys.path.insert(0, "video-mamba-suite_amd")
from mamba_ssm.modules.mamba_simple import Mamba as ViM
def run(name, block, B, L, dm, steps=50):
    block = block.cuda()
    params = list(block.parameters())
    x = torch.randn(B, L, dm, device="cuda", dtype=torch.bfloat16, requires_grad=True)
    g = torch.randn(B, L, dm, device="cuda", dtype=torch.bfloat16)
    def step():
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = block(x)
        return torch.autograd.grad(y, [x] + params, g)
    for _ in range(10): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): step()
    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / steps * 1e3
    print(f"{name:40s} NO_INNER_EXT={os.environ.get('VMS_NO_INNER_EXT','0')} step {t:6.3f} ms", flush=True)
torch.manual_seed(0)
if len(sys.argv) > 1:
    run("ViM (8, 8192, 1024) first", ViM(1024, expand=1, bimamba_type="v2"), 8, 8192, 1024, 10)
run("ViM (8, 1569, 768)", ViM(768, expand=1, bimamba_type="v2"), 8, 1569, 768)
run("ViM (8, 1568, 768)", ViM(768, expand=1, bimamba_type="v2"), 8, 1568, 768)
