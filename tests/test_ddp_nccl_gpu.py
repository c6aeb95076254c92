"""The data-parallel path on RCCL with the one GPU a test box has (no -m gpu test initialised nccl).
Reference sites: action-recognition/run_class_finetuning.py:570-582 (DistributedDataParallel), temporal-action-localization/
train_eval.py:76 (nn.DataParallel around the DBM model)."""
import json
import os
import socket
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.gpu
def test_ddp_and_graphed_ddp_on_rccl_world1(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = str(s.getsockname()[1])
    out = str(tmp_path / "nccl.json")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, os.path.join(HERE, "ddp_nccl_worker.py"), out, port]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    if r.returncode < 0 and os.environ.get("GRAFT_REPO_ROOT"):
        # killed by a signal (round 5 saw SIGABRT on a cold box: the NCCL watchdog's event queries against a graph capture in "global"
        # mode, fixed in mamba_ssm/utils/hip_graph.py with capture_error_mode="thread_local").  No second chance since round 6
        # (VERDICT r5 6b): the log is kept where the GPU box's outputs are collected and the test FAILS, so the next one is seen.
        d = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out")
        os.makedirs(d, exist_ok=True)
        open(os.path.join(d, "ddp_nccl_worker_signal.log"), "w").write(f"rc {r.returncode}\n--- stdout\n{r.stdout}\n--- stderr\n{r.stderr}")
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    res = json.load(open(out))
    assert res["backend"] == "nccl" and res["world"] == 1
    for name in ("vim", "dbm"):
        # fp32 atomics make the small parameter gradients order-dependent in the last bits; bf16 activations
        assert res[name + "_ddp_vs_bare"] < 2e-2, res
        for mode in ("after", "captured"):
            assert res[f"{name}_graph_{mode}_vs_ddp"] < 2e-2, res
            assert res[f"{name}_graph_{mode}_views"] and res[f"{name}_graph_{mode}_rebinds"], res
    assert res["vim_n_params"] == 16
    c = res["bench_ddp"]["comm"]
    assert c["backend"] == "nccl" and c["world_size"] == 1 and c["n_buckets"] >= 1 and c["rccl_version"]
    assert res["bench_ddp"]["step"].endswith("DDP all-reduce") and not res["bench_ddp"]["hip_graph"]
    c = res["bench_graph"]["comm"]
    assert res["bench_graph"]["hip_graph"] and c["backend"] == "nccl" and "one flat all-reduce per replay" in c["gradient_exchange"]


@pytest.mark.gpu
def test_ddp_multi_rank_on_rccl(tmp_path):
    """N = min(GPUs of the box, 8) ranks when that is >= 2 (skipped on the one-GPU test boxes): the data-parallel path as the
    driver's scaling bench launches it -- torch.distributed.run, one process per GPU, RCCL -- so that the first multi-GPU run is
    not also the first execution of this code.  Reference: action-recognition/run_class_finetuning.py:570."""
    import torch
    n = min(torch.cuda.device_count(), 8)
    if n < 2:
        pytest.skip(f"{torch.cuda.device_count()} GPU(s) visible: the multi-rank RCCL test needs >= 2")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = str(s.getsockname()[1])
    out = str(tmp_path / "nccl_multi.json")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
                        "--master-port", port, os.path.join(HERE, "ddp_nccl_multi_worker.py"), out], env=env, capture_output=True,
                       text=True, timeout=1800)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    res = json.load(open(out))
    assert res["backend"] == "nccl" and res["world"] == n
    assert res["ranks"] == list(range(n)) and res["devices"] == list(range(n))   # RCCL sees N ranks on N distinct GPUs
    for name in ("vim", "dbm"):
        assert res[name + "_ddp_vs_sharded_average"] < 2e-2, res   # bf16 activations, fp32 atomics: the bar of the world-1 test
        assert res[name + "_same_on_all_ranks"], res
        assert res[name + "_n_buckets"] >= 2, res
        for mode in ("after", "captured"):
            assert res[f"{name}_graph_{mode}_vs_ddp"] < 2e-2, res
    c = res["bench_ddp"]["comm"]
    assert c["backend"] == "nccl" and c["world_size"] == n and c["n_buckets"] >= 2 and c["rccl_version"]
    assert res["bench_ddp"]["n_gpus"] == n and res["bench_ddp"]["global_batch"] == 2 * n
    assert res["bench_graph"]["hip_graph"] and "one flat all-reduce per replay" in res["bench_graph"]["comm"]["gradient_exchange"]
