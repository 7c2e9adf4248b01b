"""CPU (gloo, world_size 2 and 8) coverage of the N>1 path: rank sharding of envs / trajectories, the flat-bucket
gradient all-reduce that FlatAdam.step issues before the fused Adam kernel, the MIN-reduced batch count with unequal
shards, and the rank-0 merge of per-tour evaluation records (t-nDTW equal to the single-process value)."""
import os
import sys

import torch
import torch.multiprocessing as mp
from conftest import free_port

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
    port = free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
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
    port = free_port()
    ps = [ctx.Process(target=_worker_min, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    out = sorted(q.get(timeout=120) for _ in range(2))
    for p in ps:
        p.join(30)
    for rank, errs, n, t in out:
        assert all(e is not None and "not initialised" in e for e in errs), errs
        assert n == 5 and t == [0.0, 0.0, 0.0]


# ---- world 8: configs[3] / configs[4]'s layout (64 envs -> 8 per rank; scenes / tours by index mod 8) -------------------
def _play_all(envs, iterative):
    """Every env plays all its episodes under a deterministic stand-in policy (the expert's action, except a FORWARD every
    third step) - no GPU, no policy: what is under test is who owns which env and what rank 0 makes of the gathered
    records.  Returns what the evaluation loops hand to `gather_objects`."""
    from ivln_ce_amd.envs import FORWARD

    def act(obs, t):
        return FORWARD if t % 3 == 2 else int(obs["shortest_path_sensor"][0])

    if not iterative:
        for e in envs._envs:
            obs, t = e.reset_episodic(), 0
            while not e.exhausted:
                obs, _, done, _ = e.step_episodic(act(obs, t), True)
                t = 0 if done else t + 1
        stats = {ep.episode_id: {"steps": float(len(ep.script))} for e in envs._envs for ep in e.episodes}
        return stats, (envs.dtw_data(), envs.gt_paths())
    dtw_data, stats_tours = {}, {}
    for e in envs._envs:
        (obs, _, produce), t = e.reset(), 0
        while not e.exhausted:
            tour, ep_id = e.current_episode.tour_id, e.current_episode.episode_id
            obs, _, agent_done, sim_done, _, produce2, info = e.step(act(obs, t) if produce else 0, False)
            if "dtw_data" in info and sim_done:
                dtw_data.setdefault(tour, []).extend(info["dtw_data"])
                stats_tours.setdefault(tour, {})[ep_id] = {"steps": float(t)}
            t = t + 1 if produce else t
            produce = produce2
            if sim_done:
                (obs, _, produce), t = e.reset(), 0
    return stats_tours, dtw_data, envs.gt_paths()


def _small_config():
    from ivln_ce_amd.config import get_config

    cfg = get_config()
    cfg.defrost()
    for s in (cfg.TASK_CONFIG.SIMULATOR.DEPTH_SENSOR, cfg.TASK_CONFIG.SIMULATOR.RGB_SENSOR):
        s.HEIGHT = s.WIDTH = 8  # (the frames are not read here)
    cfg.freeze()
    return cfg


def _worker8(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import ivln_ce_amd  # noqa: F401
    from ivln_ce_amd import dist as D
    from ivln_ce_amd import trainers
    from ivln_ce_amd.envs import SyntheticVectorEnv
    from ivln_ce_amd.tour_ndtw import compute_tour_ndtw

    r, _, w = D.init("gloo")
    cfg = _small_config()
    # the update's one collective: flat bucket, sum over 8 ranks
    flat = torch.arange(4096, dtype=torch.float32) * (rank + 1)
    D.allreduce_sum_(flat)
    # unequal shards (a rank's store can hold fewer trajectories): every rank runs the MIN batch count
    n_batches = D.allreduce_min_int((9, 7, 8, 12, 7, 10, 11, 9)[rank], torch.device("cpu"))
    out = {"rank": rank, "sum_ok": bool(torch.equal(flat, torch.arange(4096, dtype=torch.float32) * 36)), "n_batches": n_batches}
    for iterative in (False, True):
        envs = SyntheticVectorEnv(cfg, num_envs=8, rank=r, world=w, iterative=iterative, n_episodes=4, episodes_per_tour=2)
        out[("ids", iterative)] = [e.idx for e in envs._envs]
        gathered = D.gather_objects(_play_all(envs, iterative))
        if rank == 0:
            if iterative:
                stats, agent, gt = trainers.merge_iterative_shards(gathered)
            else:
                stats, agent, gt = trainers.merge_episodic_shards(gathered)
            out[("tndtw", iterative)] = compute_tour_ndtw(agent, gt, 3.0)
            out[("tours", iterative)] = sorted(agent)
            out[("n_stats", iterative)] = len(stats)
    q.put(out)
    torch.distributed.destroy_process_group()


def test_gloo_world8_sharding_min_batches_and_rank0_tour_ndtw():
    """64 envs over 8 ranks (env i on rank i mod 8, 8 per rank); the MIN-reduced batch count of unequal shards; rank 0's
    merge of the 8 ranks' tour records gives the t-nDTW of the single process that plays all 64 envs itself."""
    sys.path.insert(0, ROOT)
    import ivln_ce_amd  # noqa: F401
    from ivln_ce_amd import trainers
    from ivln_ce_amd.envs import SyntheticVectorEnv
    from ivln_ce_amd.tour_ndtw import compute_tour_ndtw

    world = 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    ps = [ctx.Process(target=_worker8, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    outs = sorted((q.get(timeout=300) for _ in range(world)), key=lambda o: o["rank"])
    for p in ps:
        p.join(60)
    for r, o in enumerate(outs):
        assert o["sum_ok"] and o["n_batches"] == 7
        for it in (False, True):
            assert o[("ids", it)] == [r + world * k for k in range(8)]  # 64 envs -> 8 per rank, index mod 8
    cfg = _small_config()
    for it in (False, True):
        single = SyntheticVectorEnv(cfg, num_envs=64, rank=0, world=1, iterative=it, n_episodes=4, episodes_per_tour=2)
        played = _play_all(single, it)
        if it:
            stats, agent, gt = trainers.merge_iterative_shards([played])
        else:
            stats, agent, gt = trainers.merge_episodic_shards([played])
        ref = compute_tour_ndtw(agent, gt, 3.0)
        assert outs[0][("tours", it)] == sorted(agent) and len(agent) == 128  # 64 envs x 2 tours each
        assert outs[0][("n_stats", it)] == len(stats)
        assert 0.0 < ref < 1.0 and abs(outs[0][("tndtw", it)] - ref) < 1e-12, (it, outs[0][("tndtw", it)], ref)
