"""Multi-GPU path on CPU: world_size-2 gloo job (one process per rank, rendezvous on 127.0.0.1)
running the ViM block under DistributedDataParallel with batch-axis sharding."""
import os
import socket
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return str(s.getsockname()[1])


def test_ddp_gradients_match_sharded_average(tmp_path, oracle):
    port, out = _free_port(), str(tmp_path / "ddp")
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "ddp_worker.py"), str(r), "2", port, out])
             for r in range(2)]
    for p in procs:
        assert p.wait(timeout=600) == 0
    for r in range(2):
        worst, same, n = np.load(out + f".rank{r}.npy")
        assert n == 16          # every parameter of the ViM block got a gradient
        assert same == 1.0      # all ranks hold the same reduced gradients
        assert worst < 1e-5, worst


import json

import pytest


@pytest.mark.parametrize("config", ["block", "stack", "dbm", "long"])
def test_bench_configs_under_gloo(tmp_path, oracle, config):
    """bench.py --config {block, stack, dbm, long} as a world_size-2 job (gloo, CPU, fake extensions, tiny sizes): the
    same bench.run() the driver launches with torch.distributed.run; rank 0 reports the whole-job aggregate."""
    port, out = _free_port(), str(tmp_path / "bench")
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "bench_gloo_worker.py"), str(r), "2", port, out, config])
             for r in range(2)]
    for p in procs:
        assert p.wait(timeout=600) == 0
    res = json.load(open(out + f".{config}.json"))
    b, l = {"block": (2, 24), "stack": (1, 20), "dbm": (2, 24), "long": (1, 48)}[config]
    assert res["n_gpus"] == 2 and res["config"]["global_batch"] == 2 * b and res["config"]["name"] == config
    assert res["config"]["comm"] == {"backend": "gloo", "world_size": 2}
    assert res["scaling"] == "weak" and res["config"]["parallelism"] == "dp2"
    # value = tokens of ALL ranks / max-over-ranks time
    assert abs(res["value"] - 2 * b * l * res["steps"] / (res["ms_per_step"] * 1e-3 * res["steps"])) <= 1e-6 * res["value"]
