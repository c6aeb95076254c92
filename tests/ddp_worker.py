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

    def np_(t):
        return None if t is None else t.detach().float().cpu().numpy()

    def fwd(u, delta, A, B, C, D_, z_, delta_bias_, delta_softplus):
        r = orc.scan_fwd(np_(u), np_(delta), np_(A), np_(B), np_(C), np_(D_), np_(z_), np_(delta_bias_), delta_softplus, prec="f64")
        res = [torch.empty_like(delta).copy_(torch.from_numpy(r["out"])), torch.from_numpy(r["x"])]
        if z_ is not None:
            res.append(torch.empty_like(z_).copy_(torch.from_numpy(r["out_z"])))
        return res

    def bwd(u, delta, A, B, C, D_, z_, delta_bias_, dout, x_, out_, dz_, delta_softplus, recompute_out_z):
        r = orc.scan_bwd(np_(u), np_(delta), np_(A), np_(B), np_(C), np_(D_), np_(z_), np_(delta_bias_), np_(dout), delta_softplus, prec="f64")
        tt = lambda a, like: torch.from_numpy(a).to(like.dtype)
        res = [tt(r["du"], u), torch.empty_like(delta).copy_(tt(r["ddelta"], delta)), tt(r["dA"], A), tt(r["dB"], B), tt(r["dC"], C),
               tt(r["dD"], D_) if D_ is not None else None, tt(r["ddelta_bias"], delta_bias_) if delta_bias_ is not None else None]
        if z_ is not None:
            dz = dz_ if dz_ is not None else torch.empty_like(z_)
            dz.copy_(tt(r["dz"], z_))
            res.append(dz)
        if recompute_out_z:
            f = orc.scan_fwd(np_(u), np_(delta), np_(A), np_(B), np_(C), np_(D_), np_(z_), np_(delta_bias_), delta_softplus, prec="f64")
            res.append(torch.from_numpy(f["out_z"]).to(u.dtype))
        return res

    def cfwd(x, w, b, silu):
        return torch.from_numpy(orc.conv_fwd(np_(x), np_(w), np_(b), silu, prec="f64")).to(x.dtype)

    def cbwd(x, w, b, dout, dx_, silu):
        r = orc.conv_bwd(np_(x), np_(w), np_(b), np_(dout), silu, prec="f64")
        dx = dx_ if dx_ is not None else torch.empty_like(x)
        dx.copy_(torch.from_numpy(r["dx"]))
        return [dx, torch.from_numpy(r["dweight"]), torch.from_numpy(r["dbias"]) if b is not None else None]

    fs = types.SimpleNamespace(fwd=fwd, bwd=bwd)
    fc = types.SimpleNamespace(causal_conv1d_fwd=cfwd, causal_conv1d_bwd=cbwd)
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
