"""Worker for tests/test_ddp_nccl_gpu.py: ONE process, world_size 1, backend nccl (= RCCL) on the one GPU of the box.
What can be checked with one rank: the process group comes up on RCCL, DistributedDataParallel's hooks and bucket views work
with the fused nodes (eager DDP step == bare step), the graphed data-parallel step (mamba_ssm/utils/hip_graph.py: bare module
captured + one flat all-reduce per replay, issued behind or captured inside the graph) == the eager DDP step, and bench.run()
reports the backend / bucket layout on the DDP path and on the --graph path."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "video-mamba-suite_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch
import torch.distributed as dist


def rel(a, b):
    return ((a.float() - b.float()).abs().max() / b.float().abs().max().clamp_min(1e-6)).item()


def main():
    out_path, port = sys.argv[1], sys.argv[2]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
                      HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=dev)
    res = {"backend": dist.get_backend(), "world": dist.get_world_size()}
    from mamba_ssm.modules.mamba_new import Mamba as DBM
    from mamba_ssm.modules.mamba_simple import Mamba
    from mamba_ssm.utils.hip_graph import GraphedStep
    for name, make, (b, l, dm) in (("vim", lambda: Mamba(256, d_state=16, expand=1, bimamba_type="v2"), (2, 512, 256)),
                                    ("dbm", lambda: DBM(256, d_state=16, expand=1), (2, 768, 256))):
        torch.manual_seed(0)
        block = make().to(dev)
        x = torch.randn(b, l, dm, device=dev, dtype=torch.bfloat16, requires_grad=True)
        g = torch.randn(b, l, dm, device=dev, dtype=torch.bfloat16)

        def step(m):
            block.zero_grad(set_to_none=True)
            x.grad = None
            with torch.autocast("cuda", dtype=torch.bfloat16):
                y = m(x)
            y.backward(g)
            torch.cuda.synchronize()
            return y.detach().clone(), x.grad.clone(), {k: p.grad.clone() for k, p in block.named_parameters()}
        y0, dx0, g0 = step(block)                                  # bare module
        ddp = torch.nn.parallel.DistributedDataParallel(block, device_ids=[0], bucket_cap_mb=1, gradient_as_bucket_view=True)
        for _ in range(2):                                         # the reducer rebuilds its buckets after the first step
            y1, dx1, g1 = step(ddp)
        worst_ddp = max(rel(g1[k], g0[k]) for k in g0)
        res[name + "_ddp_vs_bare"] = max(worst_ddp, rel(y1, y0), rel(dx1, dx0))
        res[name + "_n_params"] = len(g0)
        res[name + "_ddp_buckets"] = str(ddp._get_ddp_logging_data().get("rebuilt_bucket_sizes"))
        del ddp
        block.zero_grad(set_to_none=True)
        for mode in ("after", "captured"):
            gs = GraphedStep(block, x, process_group=dist.group.WORLD, allreduce=mode)
            for _ in range(2):
                yg, dxg = gs(x.detach(), g)
            torch.cuda.synchronize()
            gg = {k: p.grad for k, p in block.named_parameters()}
            res[f"{name}_graph_{mode}_vs_ddp"] = max(max(rel(gg[k], g1[k]) for k in g1), rel(yg, y1), rel(dxg, dx1))
            # every p.grad is a view of the flat buffer the all-reduce ran on
            flat = next(iter(gs.flat.values()))
            lo, hi = flat.data_ptr(), flat.data_ptr() + flat.numel() * flat.element_size()
            res[f"{name}_graph_{mode}_views"] = all(lo <= p.grad.data_ptr() < hi for p in gs.params)
            # optimizer.zero_grad(set_to_none=True) must not detach the parameters from the replayed gradients
            block.zero_grad(set_to_none=True)
            gs.replay()
            torch.cuda.synchronize()
            res[f"{name}_graph_{mode}_rebinds"] = all(p.grad is gr for p, gr in zip(gs.params, gs.dparams))
            del gs
            block.zero_grad(set_to_none=True)
    # bench.run() on the DDP path and on the graphed data-parallel path (world size 1 on RCCL)
    os.environ["VMS_BENCH_DDP_WORLD1"] = "1"
    import bench
    r = bench.run("dbm", steps=3, warmup=2, cpu_base=False, projections=False)
    res["bench_ddp"] = {"comm": r["config"]["comm"], "hip_graph": r["config"]["hip_graph"], "step": r["config"]["step"], "value": r["value"]}
    r = bench.run("dbm", steps=3, warmup=2, cpu_base=False, projections=False, graph=True)
    res["bench_graph"] = {"comm": r["config"]["comm"], "hip_graph": r["config"]["hip_graph"], "value": r["value"]}
    with open(out_path, "w") as f:
        json.dump(res, f)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
