"""One process per GPU over torch.distributed (backend "nccl" = RCCL over xGMI on ROCm; "gloo" for
the CPU tests).  The hot path has exactly ONE collective: a sum all-reduce of the flat fp32
gradient bucket between backward and the Adam step of a DAgger update; the
1/world mean is folded into the Adam kernel.  Rollout / eval steps need no communication: envs and
scenes are sharded round-robin like `construct_envs` does per process (env_utils.py:77-99)."""
import os

import torch
import torch.distributed as dist


def world_info():
    local = 0 if os.environ.get("IVLN_ONE_DEVICE") else int(os.environ.get("LOCAL_RANK", "0"))
    return int(os.environ.get("RANK", "0")), local, int(os.environ.get("WORLD_SIZE", "1"))


def init(backend: str = None):
    rank, local_rank, world = world_info()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            # IVLN_DIST_BACKEND=gloo + IVLN_ONE_DEVICE=1: every rank on one GPU - a control-flow smoke test of the
            # multi-rank paths on a 1-GPU box (RCCL refuses two ranks per device); never a measurement
            backend = os.environ.get("IVLN_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            kw["device_id"] = torch.device("cuda", local_rank)
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, local_rank, world


def _require_group(what: str) -> bool:
    """True when a collective is needed.  A multi-rank launch (WORLD_SIZE > 1 in the environment) without an
    initialised process group is an error, never a silent single-rank result: a trainer entry point that forgot
    `init()` would otherwise write its local shard as if it were the whole job."""
    world = world_info()[2]
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size() > 1
    if world > 1:
        raise RuntimeError(f"{what}: WORLD_SIZE={world} but torch.distributed is not initialised (call dist.init())")
    return False


# bench.py sets this to a list for the timed updates: every gradient all-reduce then leaves a (start, stop) HIP event pair
# recorded on the calling stream around the ONE collective call - the collective's own cost inside an update
# (`update_step.allreduce`); None = no events (the normal case)
ALLREDUCE_EVENTS = None


def allreduce_sum_(flat: torch.Tensor):
    """In-place sum all-reduce of the flat gradient bucket (26.3 MB fp32 for MapCMA): ONE collective call on the calling
    stream (the update runs on ops.eager_work_stream; torch's RCCL work is ordered against it on both sides)."""
    if _require_group("allreduce_sum_"):
        timed = ALLREDUCE_EVENTS is not None and flat.is_cuda
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        if timed:
            e1.record()
            ALLREDUCE_EVENTS.append((e0, e1))
    return flat


def broadcast_(flat: torch.Tensor, src: int = 0):
    """Rank `src`'s tensor to every rank (identical initial weights)."""
    if _require_group("broadcast_"):
        dist.broadcast(flat, src=src)
    return flat


def allreduce_min_int(value: int, device) -> int:
    """MIN over ranks of a python int (number of update batches a rank can supply: every rank must run the same
    number of `update_agent` calls, each of which contains one gradient all-reduce)."""
    if _require_group("allreduce_min_int"):
        t = torch.tensor([int(value)], dtype=torch.int64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return int(t.item())
    return int(value)


def shard(items, rank: int, world: int):
    """Round-robin split of envs / scenes / tours across ranks."""
    return [x for i, x in enumerate(items) if i % world == rank]


def gather_objects(obj):
    if _require_group("gather_objects"):
        out = [None] * dist.get_world_size()
        dist.all_gather_object(out, obj)
        return out
    return [obj]
