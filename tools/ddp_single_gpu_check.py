import os, sys, torch, torch.distributed as dist
sys.path.insert(0, "video-mamba-suite_amd")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29511")
dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from mamba_ssm.modules.mamba_simple import Mamba
torch.manual_seed(0)
block = Mamba(1024, expand=1, bimamba_type="v2").cuda()
ref = {k: None for k, _ in block.named_parameters()}
model = torch.nn.parallel.DistributedDataParallel(block, device_ids=[0], bucket_cap_mb=32, gradient_as_bucket_view=True)
x = torch.randn(8, 8192, 1024, device="cuda", dtype=torch.bfloat16, requires_grad=True)
g = torch.randn_like(x)
for it in range(3):
    model.zero_grad(set_to_none=True); x.grad = None
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = model(x)
    y.backward(g)
torch.cuda.synchronize()
ddp_grads = {k: p.grad.clone() for k, p in block.named_parameters()}
block.zero_grad(set_to_none=True); x.grad = None
with torch.autocast("cuda", dtype=torch.bfloat16):
    y = block(x)
y.backward(g)
bad = 0
for k, p in block.named_parameters():
    d = (p.grad - ddp_grads[k]).abs().max().item(); s = p.grad.abs().max().item()
    if d > 2e-2 * max(s, 1e-6): bad += 1; print("MISMATCH", k, d, s)
print("DDP(world=1, nccl) on GPU: params", len(ddp_grads), "mismatches", bad)
dist.destroy_process_group()
