"""The N > 1 product path on real hardware: two fresh child processes, one rank each, 4 + 4 clips,
must gather exactly what one process produces for the 8 clips — tokens bit for bit (Philox noise is keyed by the global
clip index, `clip_base`) and the waveform bit for bit (same kernels on the same tokens).  On a 1-GPU box both ranks
share cuda:0 and the collectives run over gloo on host copies; with >= 2 GPUs the same test runs over RCCL."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("total", [8, 5])
def test_two_ranks_gather_what_one_rank_generates(tmp_path, total):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    worker = os.path.join(HERE, "shard_worker.py")
    one = tmp_path / "one.npz"
    r = subprocess.run([sys.executable, worker, "--out", str(one), "--total", str(total)], env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    two = tmp_path / "two.npz"
    multi_gpu = torch.cuda.device_count() >= 2
    env2 = dict(env)
    if not multi_gpu:
        env2.update(VAURA_BENCH_BACKEND="gloo", VAURA_BENCH_SHARE_GPU="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), worker, "--out", str(two),
                        "--total", str(total)], env=env2, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    a, b = np.load(one), np.load(two)
    assert int(b["world"]) == 2 and int(a["world"]) == 1
    assert a["codes"].shape == (total, 9, 24) and a["codes"].min() >= 0 and a["codes"].max() < 1024
    assert np.array_equal(a["codes"], b["codes"])
    assert np.array_equal(a["wav"], b["wav"]) and np.isfinite(a["wav"]).all() and np.abs(a["wav"]).max() > 0


def test_rccl_world_of_one_executes_the_product_backend(tmp_path):
    """What a 1-GPU box CAN execute of the N > 1 product path: the RCCL backend itself.  One rank, VAURA_DIST_FORCE_GROUP=1:
    ``init_process_group("nccl", device_id=cuda:0)`` creates the communicator on the device, and the final gathers (``all_gather`` of
    DEVICE tensors), the barrier, ``all_gather_object`` and the MAX all-reduce all go through RCCL instead of being skipped — the
    branches of vaura_amd/dist.py that gloo runs never touch.  Result == the plain single-process run, bit for bit."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    worker = os.path.join(HERE, "shard_worker.py")
    one = tmp_path / "one.npz"
    r = subprocess.run([sys.executable, worker, "--out", str(one), "--total", "4"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    forced = tmp_path / "forced.npz"
    envf = dict(env, VAURA_DIST_FORCE_GROUP="1", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    r = subprocess.run([sys.executable, worker, "--out", str(forced), "--total", "4"], env=envf, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    a, b = np.load(one), np.load(forced)
    assert str(a["backend"]) == "none" and str(b["backend"]) == "nccl" and int(b["n_seen"]) == 1 and float(b["worst"]) == 0.0
    assert np.array_equal(a["codes"], b["codes"]) and np.array_equal(a["wav"], b["wav"])


def _torchrun(world, args, env, timeout):
    return subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
                           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), *args], env=env,
                          capture_output=True, text=True, timeout=timeout)


def test_eight_ranks_control_flow_configs2_shape(tmp_path):
    """BASELINE configs[2]'s control flow at world 8: 64 clips -> 8 per rank (and a ragged 61 -> 8,8,8,8,8,7,7,7), tiny model,
    every rank a fresh process running the product path, the final gather of DEVICE tensors in global clip order == what one
    process generates.  With fewer than 8 GPUs all ranks share cuda:0 and the collectives run over gloo on host copies; with 8
    GPUs the same test runs over RCCL (that run is the driver's)."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    worker = os.path.join(HERE, "shard_worker.py")
    shared = torch.cuda.device_count() < 8
    env8 = dict(env, VAURA_BENCH_BACKEND="gloo", VAURA_BENCH_SHARE_GPU="1") if shared else env
    for total in (64, 61):
        one = tmp_path / f"one{total}.npz"
        r = subprocess.run([sys.executable, worker, "--out", str(one), "--total", str(total), "--layers", "2", "--frames", "8"],
                           env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        eight = tmp_path / f"eight{total}.npz"
        r = _torchrun(8, [worker, "--out", str(eight), "--total", str(total), "--layers", "2", "--frames", "8"], env8, 1500)
        assert r.returncode == 0, r.stderr[-3000:]
        a, b = np.load(one), np.load(eight)
        assert int(b["world"]) == 8 and a["codes"].shape == (total, 9, 8)
        assert np.array_equal(a["codes"], b["codes"]) and np.array_equal(a["wav"], b["wav"])


def test_bench_launches_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` from a plain shell (WORLD_SIZE unset) starts its two ranks as fresh child processes and relays
    rank 0's JSON line (the form the round-end driver uses for N = 1).  On a 1-GPU box the ranks share the GPU (control flow only,
    flagged in the record); with >= 2 GPUs this is the real RCCL run."""
    import json
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    if torch.cuda.device_count() < 2:
        env.update(VAURA_BENCH_BACKEND="gloo", VAURA_BENCH_SHARE_GPU="1")
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(HERE), "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--no-extras"], env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    rec = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert rec["n_gpus"] == 2 and rec["config"]["global_batch"] == 16 and rec["value"] > 0
    assert rec.get("shared_gpu_control_flow_only", False) == (torch.cuda.device_count() < 2)
