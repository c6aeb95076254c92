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
