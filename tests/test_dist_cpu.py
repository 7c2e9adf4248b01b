"""CPU (gloo, world_size 2) coverage of the N>1 path: rank sharding of envs / trajectories and the
flat-bucket gradient all-reduce that FlatAdam.step issues before the fused Adam kernel."""
import os
import sys

import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import ivln_ce_amd  # noqa: F401
    from ivln_ce_amd import dist as D
    from ivln_ce_amd.config import get_config
    from ivln_ce_amd.envs import SyntheticVectorEnv

    r, _, w = D.init("gloo")
    flat = torch.full((1000,), float(rank + 1))
    D.allreduce_sum_(flat)
    envs = SyntheticVectorEnv(get_config(), num_envs=2, rank=r, world=w)
    ids = [e.idx for e in envs._envs]
    gathered = D.gather_objects(ids)
    q.put((rank, float(flat[0]), ids, gathered, D.shard(list(range(7)), r, w)))
    torch.distributed.destroy_process_group()


def test_gloo_world2_allreduce_and_sharding():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, 2, 29731, q)) for r in range(2)]
    for p in ps:
        p.start()
    out = sorted(q.get(timeout=120) for _ in range(2))
    for p in ps:
        p.join(30)
    assert out[0][1] == 3.0 and out[1][1] == 3.0          # sum over ranks
    assert out[0][2] == [0, 2] and out[1][2] == [1, 3]    # envs round-robin over ranks
    assert out[0][3] == [[0, 2], [1, 3]]
    assert out[0][4] == [0, 2, 4, 6] and out[1][4] == [1, 3, 5]


def _worker_min(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import ivln_ce_amd  # noqa: F401
    from ivln_ce_amd import dist as D

    # a multi-rank launch without a process group must fail loudly, not return the local shard
    errs = []
    for fn in (lambda: D.gather_objects(rank), lambda: D.allreduce_sum_(torch.ones(2)),
               lambda: D.broadcast_(torch.ones(2)), lambda: D.allreduce_min_int(3, torch.device("cpu"))):
        try:
            fn()
            errs.append(None)
        except RuntimeError as e:
            errs.append(str(e))
    D.init("gloo")
    n = D.allreduce_min_int(5 + rank, torch.device("cpu"))  # what DaggerTrainer does with its batch count
    t = torch.full((3,), float(rank))
    D.broadcast_(t, src=0)
    q.put((rank, errs, n, t.tolist()))
    torch.distributed.destroy_process_group()


def test_collectives_refuse_to_run_ungrouped_and_min_reduce():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker_min, args=(r, 2, 29741, q)) for r in range(2)]
    for p in ps:
        p.start()
    out = sorted(q.get(timeout=120) for _ in range(2))
    for p in ps:
        p.join(30)
    for rank, errs, n, t in out:
        assert all(e is not None and "not initialised" in e for e in errs), errs
        assert n == 5 and t == [0.0, 0.0, 0.0]
