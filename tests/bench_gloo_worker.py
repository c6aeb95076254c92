"""Worker for tests/test_ddp_gloo.py::test_bench_configs_under_gloo: one rank of a world_size-2 gloo job that runs
bench.run() -- the very function bench.py's command line calls -- for one --config on CPU: tiny sizes, the two extension
modules (and the norm extension) replaced by checker-backed fakes.  What is under test is bench.py's multi-process path
for every config: DDP wrapping of each workload, barrier + max-over-ranks timing, the whole-job token count, rank-0-only
reporting."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "video-mamba-suite_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def main():
    rank, world, port, out, config = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), OMP_NUM_THREADS="2")
    torch.set_num_threads(2)
    from ddp_worker import install_fakes
    install_fakes()
    from oracle import oracle as orc
    from fake_ext import make_norm_fake
    import mamba_ssm.ops.triton.layernorm as lnm
    lnm.layer_norm_cuda = make_norm_fake(orc)
    import bench
    dims = {"block": (2, 24, 32), "stack": (1, 20, 32), "dbm": (2, 24, 32), "long": (1, 48, 32), "block_expand2": (2, 24, 32),
            "vivim_s": (1, 20, 32), "dbm_pyramid": (2, 24, 32)}[config]
    res = bench.run(config, steps=2, warmup=1, device="cpu", backend="gloo", dims=dims, autocast=False,
                    cpu_base=False, projections=False)
    assert (res is not None) == (rank == 0)
    if rank == 0:
        with open(out + f".{config}.json", "w") as f:
            json.dump(res, f)
    torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
