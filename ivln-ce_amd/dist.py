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


def allreduce_sum_(flat: torch.Tensor):
    """In-place sum all-reduce of the flat gradient bucket (26.3 MB fp32 for MapCMA)."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    return flat


def shard(items, rank: int, world: int):
    """Round-robin split of envs / scenes / tours across ranks."""
    return [x for i, x in enumerate(items) if i % world == rank]


def gather_objects(obj):
    if dist.is_initialized() and dist.get_world_size() > 1:
        out = [None] * dist.get_world_size()
        dist.all_gather_object(out, obj)
        return out
    return [obj]
