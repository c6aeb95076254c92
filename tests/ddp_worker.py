"""Worker for tests/test_ddp_gloo.py: one rank of a world_size-2 gloo job on CPU.

Each rank wraps the ViM block in DistributedDataParallel and runs forward+backward on its own
shard of the batch (batch-axis sharding).  The two extension modules are replaced by
checker-backed fakes (the product has no CPU path); what is under test is that the fused autograd
nodes behave under DDP: every parameter receives its gradient exactly once, the reducer's buckets
fire, and the all-reduced gradients equal the average of the per-shard gradients."""
import os
import sys
import types

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "video-mamba-suite_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def install_fakes():
    from oracle import oracle as orc
    from mamba_ssm.ops import selective_scan_interface as ssi
    import causal_conv1d.causal_conv1d_interface as cci

    from fake_ext import make_fakes
    fs, fc = make_fakes(orc)
    ssi.selective_scan_cuda = fs
    ssi.causal_conv1d_cuda = fc
    cci.causal_conv1d_cuda = fc


def main():
    rank, world, port, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    os.environ["OMP_NUM_THREADS"] = "2"
    install_fakes()
    from mamba_ssm.modules.mamba_simple import Mamba
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)  # same weights on every rank
    block = Mamba(32, d_state=8, expand=1, bimamba_type="v2")
    ddp = torch.nn.parallel.DistributedDataParallel(block, bucket_cap_mb=1, gradient_as_bucket_view=True)
    torch.manual_seed(100)
    full = torch.randn(2 * world, 24, 32)  # global batch, identical on all ranks
    gfull = torch.randn(2 * world, 24, 32)
    shard = slice(2 * rank, 2 * rank + 2)
    y = ddp(full[shard])
    y.backward(gfull[shard])
    grads = {k: p.grad.clone() for k, p in block.named_parameters()}
    # reference: the same module, un-wrapped, on every shard, averaged
    ref = {k: torch.zeros_like(p) for k, p in block.named_parameters()}
    for r in range(world):
        block.zero_grad()
        s = slice(2 * r, 2 * r + 2)
        block(full[s]).backward(gfull[s])
        for k, p in block.named_parameters():
            ref[k] += p.grad / world
    worst = max(((grads[k] - ref[k]).abs().max() / (ref[k].abs().max() + 1e-12)).item() for k in grads)
    # every rank must hold identical (all-reduced) gradients
    flat = torch.cat([g.flatten() for g in grads.values()])
    gathered = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    same = all(torch.equal(gathered[0], g) for g in gathered)
    np.save(out + f".rank{rank}.npy", np.array([worst, float(same), float(len(grads))]))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
