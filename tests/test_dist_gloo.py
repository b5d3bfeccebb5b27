"""world_size-2 gloo run of the only exchange on the path: the final gather of per-rank results
(vaura_amd/dist.py).  CPU tensors stand in for the tokens / waveforms a rank produced."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from vaura_amd import dist as vdist


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    r, _, w = vdist.init("gloo")
    assert (r, w) == (rank, world)
    counts = [vdist.shard(total, i, world)[1] for i in range(world)]
    first, n = vdist.shard(total, rank, world)
    # "tokens" of clip c are all equal to c: global order must come back regardless of raggedness
    local = torch.arange(first, first + n, dtype=torch.int32)[:, None, None].expand(n, 9, 5).contiguous()
    full = vdist.gather_clips(local, counts)
    wav = torch.arange(first, first + n, dtype=torch.float32)[:, None, None].expand(n, 1, 7).contiguous()
    fullw = vdist.gather_clips(wav, counts)
    t = vdist.max_over_ranks(float(rank + 1), "cpu")
    seen = vdist.ranks_seen("cpu")                      # who took part (the bench record's `ranks_seen` / `ranks`)
    per_rank = vdist.gather_floats(10.0 * (rank + 1), "cpu")
    vdist.barrier()
    ok = (full.shape == (total, 9, 5) and bool((full[:, 0, 0] == torch.arange(total, dtype=torch.int32)).all())
          and bool((fullw[:, 0, 0] == torch.arange(total, dtype=torch.float32)).all()) and t == float(world)
          and [d["rank"] for d in seen] == list(range(world)) and per_rank == [10.0 * (i + 1) for i in range(world)])
    out[rank] = ok
    dist.destroy_process_group()


@pytest.mark.parametrize("total", [8, 5])
def test_final_gather_two_ranks(total):
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), total, out), nprocs=world, join=True)
    assert all(out[r] for r in range(world))


def test_gpu_count_without_hip_never_touches_torch_cuda():
    """bench.py's launcher parent counts GPUs from sysfs (it must not initialise HIP before it spawns the ranks): an int or None,
    and no CUDA context afterwards."""
    n = vdist.count_gpus_without_hip()
    assert n is None or (isinstance(n, int) and n >= 0)
    assert not torch.cuda.is_initialized()
    assert vdist.ranks_seen("cpu")[0]["rank"] == 0 and vdist.gather_floats(1.5, "cpu") == [1.5]     # single process: no group needed
