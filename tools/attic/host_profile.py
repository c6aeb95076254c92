"""Where the host time of one small block step goes (cProfile, GPU idle most of the time at this size)."""
import cProfile, pstats, os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "video-mamba-suite_amd"))
from mamba_ssm.modules.mamba_new import Mamba as DBM
block = DBM(512, expand=1).cuda()
x = torch.randn(2, 2304, 512, device="cuda", dtype=torch.bfloat16, requires_grad=True)
g = torch.randn(2, 2304, 512, device="cuda", dtype=torch.bfloat16)
def step():
    block.zero_grad(set_to_none=True); x.grad = None
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = block(x)
    y.backward(g)
for _ in range(10): step()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(50): step()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(45)
