"""Worker for tests/test_ddp_nccl_gpu.py::test_ddp_multi_rank_on_rccl: ONE RANK of an N-rank job (N = the box's GPU count, >= 2),
launched by `python -m torch.distributed.run --nproc-per-node N` (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the launcher),
backend nccl (= RCCL over xGMI), one process per GPU -- the reference's only parallel strategy
(action-recognition/run_class_finetuning.py:570-582, egocentric-understanding/engine/main_lavila_pretrain.py:147).

Checked on every rank, reported by rank 0:
  * the process group has N ranks on N distinct devices (an all-gather of (rank, device index));
  * eager DistributedDataParallel gradients of the ViM and DBM blocks == the average of the per-shard gradients of the BARE
    module (each rank runs its shard un-wrapped, the sums are all-reduced), and are bit-identical on every rank;
  * GraphedStep(process_group=...) (the bare module captured + one flat all-reduce per replay) == the eager DDP step;
  * bench.run() on the DDP path reports backend nccl, world_size N and >= 2 buckets with a small bucket cap."""
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
    out_path = sys.argv[1]
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    dist.init_process_group(backend="nccl", device_id=dev)
    res = {"backend": dist.get_backend(), "world": dist.get_world_size()}
    ids = torch.tensor([rank, torch.cuda.current_device()], device=dev)
    got = [torch.empty_like(ids) for _ in range(world)]
    dist.all_gather(got, ids)
    res["ranks"] = sorted(int(t[0]) for t in got)
    res["devices"] = sorted(int(t[1]) for t in got)
    from mamba_ssm.modules.mamba_new import Mamba as DBM
    from mamba_ssm.modules.mamba_simple import Mamba
    from mamba_ssm.utils.hip_graph import GraphedStep
    for name, make, (b, l, dm) in (("vim", lambda: Mamba(256, d_state=16, expand=1, bimamba_type="v2"), (2, 512, 256)),
                                    ("dbm", lambda: DBM(256, d_state=16, expand=1), (2, 768, 256))):
        torch.manual_seed(0)                       # the same weights on every rank
        block = make().to(dev)
        torch.manual_seed(100 + rank)              # this rank's shard of the global batch
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
        _, _, g_bare = step(block)                 # un-wrapped, this shard only
        ref = {}
        for k, t in g_bare.items():                # the average over the shards, by hand
            t = t.float().clone()
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            ref[k] = t / world
        ddp = torch.nn.parallel.DistributedDataParallel(block, device_ids=[local], bucket_cap_mb=0.05, gradient_as_bucket_view=True)
        for _ in range(2):                         # the reducer rebuilds its buckets after the first step
            y1, dx1, g1 = step(ddp)
        res[name + "_ddp_vs_sharded_average"] = max(rel(g1[k], ref[k]) for k in ref)
        flat = torch.cat([t.flatten().float() for t in g1.values()])
        allf = [torch.empty_like(flat) for _ in range(world)]
        dist.all_gather(allf, flat)
        res[name + "_same_on_all_ranks"] = all(torch.equal(allf[0], t) for t in allf)
        import bench
        res[name + "_n_buckets"] = bench.ddp_buckets(ddp, 0.05)["n_buckets"] or 0
        del ddp
        block.zero_grad(set_to_none=True)
        for mode in ("after", "captured"):
            gs = GraphedStep(block, x, process_group=dist.group.WORLD, allreduce=mode)
            for _ in range(2):
                yg, dxg = gs(x.detach(), g)
            torch.cuda.synchronize()
            gg = {k: p.grad for k, p in block.named_parameters()}
            res[f"{name}_graph_{mode}_vs_ddp"] = max(max(rel(gg[k], g1[k]) for k in g1), rel(yg, y1), rel(dxg, dx1))
            del gs
            block.zero_grad(set_to_none=True)
    os.environ["VMS_DDP_BUCKET_MB"] = "0.05"
    r = bench.run("dbm", steps=3, warmup=2, cpu_base=False, projections=False)
    if rank == 0:
        res["bench_ddp"] = {"comm": r["config"]["comm"], "n_gpus": r["n_gpus"], "global_batch": r["config"]["global_batch"], "value": r["value"]}
    r = bench.run("dbm", steps=3, warmup=2, cpu_base=False, projections=False, graph=True)
    if rank == 0:
        res["bench_graph"] = {"comm": r["config"]["comm"], "hip_graph": r["config"]["hip_graph"]}
        with open(out_path, "w") as f:
            json.dump(res, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
