import sys, torch
sys.path.insert(0, 'video-mamba-suite_amd')
import vms_hip
torch.manual_seed(0)
b, rows, k, L = 1, 128, 64, 256
w = (torch.randn(rows, k, device='cuda') * 0.2).bfloat16()
x = torch.randn(b, k, L, device='cuda').bfloat16()
out = torch.randn(b, rows, L, device='cuda').bfloat16()
old = out.clone()
T = (w.double() @ x.double())
vms_hip.proj_apply(w, x, out, True)
d = out.double() - T
print("err vs T+old", (d - old.double()).abs().max().item(), " vs T", d.abs().max().item(), " vs T+2old", (d - 2*old.double()).abs().max().item())
e = (d - old.double()).abs()[0]
idx = (e > 0.1).nonzero()
print("bad count", idx.shape[0], "of", e.numel())
print(idx[:20].tolist())
# does d equal old at some permuted position?
r, c = idx[0].tolist() if idx.shape[0] else (0, 0)
print("at", r, c, "d", d[0, r, c].item(), "old", old[0, r, c].item())
m = (old.double()[0] - d[0, r, c]).abs()
print("old positions matching d:", (m < 1e-2).nonzero()[:8].tolist())
