"""Batch-parallel sharding of independent clips over the GPUs of one node.

The reference has no inference-time parallelism (single device, sequential batches:
configs/generate_vgg.yaml:18, scripts/generate.py:264).  Clips are independent
(SURVEY.md §8e), so each rank owns a contiguous slice of the global batch, keeps a full weight
replica, and the only exchange is ONE all_gather of the results at the end (tokens and
waveform).  Sampling noise is keyed by global clip index (``clip_base``) so the result does
not depend on the world size.  No collective sits on the data path.
"""
from __future__ import annotations

import os
from typing import List, Tuple

import torch
import torch.distributed as dist


def env_rank() -> Tuple[int, int, int]:
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def _single() -> bool:
    """True when the collectives may be skipped: no group, or a group of ONE rank that was not asked for explicitly.  With
    VAURA_DIST_FORCE_GROUP=1 a world of one rank still creates its RCCL communicator and sends every helper through the real
    collective calls — the only way a 1-GPU box can execute ``init_process_group("nccl", device_id=...)`` and the device-tensor branches
    of the gathers at all (tests/test_gpu_multirank.py::test_rccl_world_of_one...)."""
    if not dist.is_initialized():
        return True
    return dist.get_world_size() == 1 and os.environ.get("VAURA_DIST_FORCE_GROUP") != "1"


def init(backend: str = "nccl", device_index: int | None = None) -> Tuple[int, int, int]:
    """Join the job's process group.  With the product backend ("nccl" = RCCL) the rank first binds its own GPU
    (``device_index``, default LOCAL_RANK) and hands it to ``init_process_group(device_id=...)``: the communicator is then
    created eagerly on that device instead of on whatever device the first collective happens to see."""
    rank, local, world = env_rank()
    if (world > 1 or os.environ.get("VAURA_DIST_FORCE_GROUP") == "1") and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        kw = {}
        if backend == "nccl":
            idx = local if device_index is None else device_index
            torch.cuda.set_device(idx)
            kw["device_id"] = torch.device("cuda", idx)
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, local, world


def count_gpus_without_hip() -> int | None:
    """GPU agents of this node read from the KFD topology in sysfs — no HIP / torch.cuda call, so a launcher parent that must
    not initialise the GPU before it spawns its ranks can still refuse an impossible --gpus N.  None when sysfs has no answer."""
    root = "/sys/class/kfd/kfd/topology/nodes"
    try:
        n = 0
        for node in os.listdir(root):
            with open(os.path.join(root, node, "properties")) as f:
                props = dict(line.split()[:2] for line in f if len(line.split()) >= 2)
            if int(props.get("simd_count", "0")) > 0:
                n += 1
        return n
    except (OSError, ValueError):
        return None


def ranks_seen(device) -> List[dict]:
    """One all_gather over the job's backend of who took part: rank, local device index and device name of every rank
    (the bench record carries it, so a reader can check that the collective really spanned N devices)."""
    dev = torch.device(device)
    me = {"rank": env_rank()[0], "device": str(dev), "name": torch.cuda.get_device_name(dev) if dev.type == "cuda" else "cpu"}
    if dev.type == "cuda":
        import socket
        props = torch.cuda.get_device_properties(dev)
        me["uuid"] = str(getattr(props, "uuid", ""))
        # a stable hardware id that survives per-rank HIP_VISIBLE_DEVICES isolation (every rank then sees ITS gpu as cuda:0) and runtimes
        # that report one uuid for all devices: host + PCI domain:bus:device
        bus = getattr(props, "pci_bus_id", None)
        if bus is not None:
            me["pci"] = f"{socket.gethostname()}/{int(getattr(props, 'pci_domain_id', 0)):04x}:{int(bus):02x}:{int(getattr(props, 'pci_device_id', 0)):02x}"
    if _single():
        return [me]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, me)
    return out


def assert_distinct_devices(seen: List[dict]) -> None:
    """Every rank of a clip-parallel job must sit on its OWN GPU: two ranks with the same device uuid means a launcher bound them
    to one device, and the aggregate rate would be a shared-GPU artefact (and the one-launch MLP's in-launch hand-off would starve,
    csrc/mlp_engine.h).  Raises on every rank that sees the list (all of them: ``ranks_seen`` is an all-gather)."""
    uu, dv, pci = [r.get("uuid") for r in seen], [r.get("device") for r in seen], [r.get("pci") for r in seen]
    ids = [(u or d) for u, d in zip(uu, dv)]
    if len(seen) > 1 and all(pci):
        # host + PCI address: what physically distinguishes two GPUs, whatever the ranks' device indices or the runtime's uuids say
        ids = pci
    elif len(seen) > 1 and all(uu) and len(set(uu)) == 1 and len(set(dv)) == len(seen):
        # a runtime that reports ONE uuid for every device (seen on some ROCm builds) while every rank sits on its own device index:
        # the uuid does not distinguish anything there — judge by the device index instead of failing a correct launch
        ids = dv
    dup = sorted({i for i in ids if ids.count(i) > 1})
    if len(seen) > 1 and dup:
        raise RuntimeError(f"{len(seen)} ranks but only {len(set(ids))} distinct GPU(s): device id(s) {dup} are shared by several ranks "
                           f"({[(r['rank'], r.get('device')) for r, i in zip(seen, ids) if i in dup]})")


def gather_floats(x: float, device) -> List[float]:
    """The same scalar from every rank (per-rank timings of the bench record)."""
    if _single():
        return [x]
    t = torch.tensor([x], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else device)
    outs = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(outs, t)
    return [float(o.item()) for o in outs]


def shard(total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous split of `total` clips: returns (first_clip, count) of `rank`; remainders go to
    the lowest ranks."""
    base, rem = divmod(total, world)
    count = base + (1 if rank < rem else 0)
    first = rank * base + min(rank, rem)
    return first, count


def gather_clips(local: torch.Tensor, counts: List[int]) -> torch.Tensor:
    """All-gather per-rank results (clip dim 0, possibly ragged) into global clip order."""
    if _single():
        return local
    world = dist.get_world_size()
    mx = max(counts)
    pad = local
    if local.shape[0] < mx:
        pad = torch.cat([local, local.new_zeros((mx - local.shape[0],) + tuple(local.shape[1:]))], dim=0)
    src = pad.contiguous()
    if dist.get_backend() == "gloo" and src.is_cuda:   # debugging aid: gloo collectives on host copies
        host = src.cpu()
        outs = [torch.empty_like(host) for _ in range(world)]
        dist.all_gather(outs, host)
        out = [o.to(src.device) for o in outs]
    else:
        out = [torch.empty_like(src) for _ in range(world)]
        dist.all_gather(out, src)
    return torch.cat([o[:c] for o, c in zip(out, counts)], dim=0)


def barrier():
    if not _single():
        dist.barrier()


def max_over_ranks(x: float, device) -> float:
    if _single():
        return x
    t = torch.tensor([x], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
