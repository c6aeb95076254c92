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


@pytest.mark.parametrize("config", ["block", "stack", "dbm", "long", "block_expand2", "vivim_s", "dbm_pyramid"])
def test_bench_configs_under_gloo(tmp_path, oracle, config):
    """bench.py --config {block, stack, dbm, long} as a world_size-2 job (gloo, CPU, fake extensions, tiny sizes): the
    same bench.run() the driver launches with torch.distributed.run; rank 0 reports the whole-job aggregate."""
    port, out = _free_port(), str(tmp_path / "bench")
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "bench_gloo_worker.py"), str(r), "2", port, out, config])
             for r in range(2)]
    for p in procs:
        assert p.wait(timeout=600) == 0
    res = json.load(open(out + f".{config}.json"))
    b, l = {"block": (2, 24), "stack": (1, 20), "dbm": (2, 24), "long": (1, 48), "block_expand2": (2, 24), "vivim_s": (1, 20),
            "dbm_pyramid": (2, 24)}[config]
    assert res["n_gpus"] == 2 and res["config"]["global_batch"] == 2 * b and res["config"]["name"] == config
    comm = res["config"]["comm"]
    assert comm["backend"] == "gloo" and comm["world_size"] == 2
    assert comm["bucket_cap_mb"] == 4.0 and comm["n_buckets"] >= 1 and sum(comm["bucket_bytes"]) > 0   # what the reducer built
    assert res["scaling"] == "weak" and res["config"]["parallelism"] == "dp2"
    # value = tokens of ALL ranks / max-over-ranks time
    assert abs(res["value"] - 2 * b * l * res["steps"] / (res["ms_per_step"] * 1e-3 * res["steps"])) <= 1e-6 * res["value"]


def test_bench_gpus_flag_is_a_contract(monkeypatch):
    """bench.py --gpus N: WORLD_SIZE must equal N (the flag was parsed and never read)."""
    import bench
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    assert bench.ensure_world(1, ["bench.py"]) is None                      # one GPU, no launcher: run here
    monkeypatch.setenv("WORLD_SIZE", "2")
    assert bench.ensure_world(2, ["bench.py", "--gpus", "2"]) is None       # launched with the right size
    with pytest.raises(SystemExit, match="--gpus 8 but WORLD_SIZE=2"):
        bench.ensure_world(8, ["bench.py", "--gpus", "8"])
    with pytest.raises(SystemExit, match="--gpus 1 but WORLD_SIZE=2"):
        bench.ensure_world(1, ["bench.py"])
    monkeypatch.delenv("WORLD_SIZE")
    monkeypatch.setenv("VMS_BENCH_NO_SELF_LAUNCH", "1")
    with pytest.raises(SystemExit, match="needs one process per GPU"):
        bench.ensure_world(4, ["bench.py", "--gpus", "4"])


def test_bench_self_launches_under_torch_distributed_run(oracle):
    """`python bench.py --gpus 2` with no launcher re-launches itself as a 2-rank torch.distributed.run job (gloo, CPU, fake
    extensions, tiny sizes): one JSON line from rank 0 with the whole-job aggregate, DDP buckets reported."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(OMP_NUM_THREADS="2", MASTER_ADDR="127.0.0.1", VMS_DDP_BUCKET_MB="0.001")
    r = subprocess.run([sys.executable, os.path.join(HERE, "bench_cli_worker.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--config", "block", "--device", "cpu", "--backend", "gloo", "--dims", "2,24,32"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "re-launching as" in r.stderr and "--nproc-per-node=2" in r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["config"]["global_batch"] == 4 and res["config"]["parallelism"] == "dp2"
    comm = res["config"]["comm"]
    assert comm["backend"] == "gloo" and comm["world_size"] == 2
    # a 1 KB cap over ~20 KB of gradients: the reducer's rebuilt buckets are several, and their bytes add up to the gradients'
    assert comm["n_buckets"] and comm["n_buckets"] > 1 and sum(comm["bucket_bytes"]) > 0
