"""bench.py's COMMAND LINE on CPU for tests/test_ddp_gloo.py: the extension modules are replaced by checker-backed fakes
(the product has no CPU path), then bench.main() parses sys.argv exactly as `python bench.py ...` does -- including the
`--gpus N` contract: without WORLD_SIZE it re-launches THIS script under torch.distributed.run, one process per rank."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "video-mamba-suite_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def main():
    import torch
    torch.set_num_threads(2)
    from ddp_worker import install_fakes
    install_fakes()
    from oracle import oracle as orc
    from fake_ext import make_norm_fake
    import mamba_ssm.ops.triton.layernorm as lnm
    lnm.layer_norm_cuda = make_norm_fake(orc)
    import bench
    return bench.main()


if __name__ == "__main__":
    sys.exit(main())
