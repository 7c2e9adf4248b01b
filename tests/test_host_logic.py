"""CPU tests of the host-side logic that mirrors the reference: config surface, registry names,
collate / inflection weights / block shuffle, trajectory store, t-nDTW known answers, synthetic env."""
import os
import random
import sys

import numpy as np
import pytest
import torch

import ivln_ce_amd  # noqa: F401

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
from ivln_ce_amd.config import Config, get_config


def test_config_defaults_and_yaml_merge(tmp_path):
    cfg = get_config()
    assert cfg.IL.lr == 2.5e-4 and cfg.IL.batch_size == 5 and cfg.IL.inflection_weight_coef == 3.2
    assert cfg.MODEL.STATE_ENCODER.hidden_size == 512 and cfg.MODEL.SEMANTIC_MAP_ENCODER.last_ch_mult == 4
    assert cfg.RL.POLICY.OBS_TRANSFORMS.EGOCENTRIC_MAPPER.resolution_meters == 0.1
    assert cfg.EVAL.ITERATIVE_MAP_RESET == "iterative"
    with pytest.raises(AttributeError):
        cfg.IL.lr = 1.0  # frozen like yacs
    task = tmp_path / "task.yaml"
    task.write_text("SIMULATOR:\n  DEPTH_SENSOR:\n    WIDTH: 128\n    HEIGHT: 128\n")
    exp = tmp_path / "exp.yaml"
    # the reference's eval YAMLs carry the lower-case key (quirk Q8): accepted as an unused key
    exp.write_text(f"BASE_TASK_CONFIG_PATH: {task}\nNUM_ENVIRONMENTS: 8\nEVAL:\n  SPLIT: val_unseen\n"
                   "  iterative_map_reset: episodic\nMODEL:\n  policy_name: MapCMAPolicy\n"
                   "RL:\n  POLICY:\n    OBS_TRANSFORMS:\n      ENABLED_TRANSFORMS: [GTSemanticsIterativeMapper]\n")
    cfg = get_config(str(exp), ["IL.lr", "1e-3", "TORCH_GPU_ID", 1])
    assert cfg.NUM_ENVIRONMENTS == 8 and cfg.EVAL.SPLIT == "val_unseen" and cfg.IL.lr == 1e-3 and cfg.TORCH_GPU_ID == 1
    assert cfg.EVAL.ITERATIVE_MAP_RESET == "iterative" and cfg.EVAL.iterative_map_reset == "episodic"
    assert cfg.TASK_CONFIG.SIMULATOR.DEPTH_SENSOR.WIDTH == 128
    assert cfg.CMD_TRAILING_OPTS == ["IL.lr", "1e-3", "TORCH_GPU_ID", 1]
    import pickle

    c2 = pickle.loads(pickle.dumps(cfg))  # checkpoints pickle the config (base_il_trainer.py:158-168)
    assert isinstance(c2, Config) and c2.MODEL.policy_name == "MapCMAPolicy" and c2.is_frozen()


def test_reference_yaml_files_load_when_present():
    ref = "/root/reference/ivlnce_baselines/config/map_cma/gt_semantics/iterative_maps/2_eval_episodic.yaml"
    if not os.path.exists(ref):
        pytest.skip("reference tree not mounted (GPU box)")
    cfg = get_config(ref)
    assert cfg.MODEL.policy_name == "MapCMAPolicy" and cfg.NUM_ENVIRONMENTS == 4
    assert cfg.RL.POLICY.OBS_TRANSFORMS.ENABLED_TRANSFORMS == ["GTSemanticsIterativeMapper"]


def test_registry_names_match_reference():
    from ivln_ce_amd import latent_policy, obs_transforms, policy, trainers  # noqa: F401
    from ivln_ce_amd.registry import baseline_registry as reg

    assert reg.get_policy("MapCMAPolicy") is policy.MapCMAPolicy
    assert reg.get_policy("LatentCMAPolicy") is latent_policy.LatentCMAPolicy
    for n in ["GTSemanticsIterativeMapper", "PredictedSemanticsIterativeMapper", "GTSemanticsKnownMapper",
              "PredictedSemanticsKnownMapper"]:
        assert reg.get_obs_transformer(n) is getattr(obs_transforms, n)
    assert reg.get_trainer("dagger") is trainers.DaggerTrainer
    assert reg.get_trainer("iterative_collection_dagger") is trainers.IterativeCollectionDaggerTrainer


def test_obs_transform_observation_space():
    from ivln_ce_amd.obs_transforms import GTSemanticsIterativeMapper
    from ivln_ce_amd.spaces import Box, Dict

    tr = GTSemanticsIterativeMapper.from_config(get_config())
    sp = Dict({"depth": Box(0, 1, (256, 256, 1)), "semantic12": Box(0, 12, (256, 256, 1), np.uint8),
               "world_robot_pose": Box(-1, 1, (3,)), "world_robot_orientation": Box(-1, 1, (2,), np.float64)})
    out = tr.transform_observation_space(sp)
    assert out.spaces["occupancy_map"].shape == (64, 64) and out.spaces["semantic_map"].dtype == np.uint8
    assert "semantic12" not in out.spaces and "world_robot_pose" not in out.spaces and "depth" in out.spaces
    assert abs(tr.camera_parameters.vertical_fov_radians - np.pi / 2) < 1e-12


def test_collate_and_inflection_weights(tmp_path):
    """dagger_trainer.py:42-117,193-214: pad obs with 1.0 / actions+weights with 0, time-major flatten,
    masks zero on the first step; inflection weights [1, 3.2][a_t != a_{t-1}], first step inflection."""
    from ivln_ce_amd.trainers import IWTrajectoryDataset, TrajectoryStore, collate_fn

    store = TrajectoryStore(str(tmp_path / "traj"))
    acts = [[1, 1, 2, 2, 0], [1, 3, 0]]
    for i, a in enumerate(acts):
        T = len(a)
        store.put(i, {"occupancy_map": np.full((T, 4, 4), i + 2, np.uint8), "progress": np.arange(T, dtype=np.float64).reshape(T, 1)},
                  [0] + a[:-1], a)
    assert len(store) == 2
    random.seed(0)
    ds = IWTrajectoryDataset(store, use_iw=True, inflection_weight_coef=3.2, batch_size=2)
    items = list(iter(ds))
    by_len = {len(it[1]): it for it in items}
    assert torch.allclose(by_len[5][3], torch.tensor([3.2, 1.0, 3.2, 1.0, 3.2]))
    assert torch.allclose(by_len[3][3], torch.tensor([3.2, 3.2, 3.2]))
    obs, prev, nd, corr, w = collate_fn([by_len[5], by_len[3]])
    assert obs["occupancy_map"].shape == (10, 4, 4) and prev.shape == (10, 1) and nd.shape == (10, 1)
    assert corr.shape == (5, 2) and w.shape == (5, 2)
    assert nd.view(5, 2)[0].tolist() == [0, 0] and nd.view(5, 2)[1:].min() == 1
    om = obs["occupancy_map"].view(5, 2, 4, 4)
    assert float(om[4, 1].max()) == 1.0 and float(om[2, 1].min()) == 3.0  # padded steps of the short one = 1.0
    assert w[3:, 1].tolist() == [0.0, 0.0] and corr[3:, 1].tolist() == [0, 0]
    # rank sharding of the record indices
    assert IWTrajectoryDataset(store, True, 3.2, 1, rank=1, world=2).indices == [1]


def test_tour_ndtw_known_answers():
    from ivln_ce_amd.tour_ndtw import compute_tour_ndtw, dtw_symmetric1, novel_only, window_align

    a = np.array([[0, 0, 0], [1, 0, 0], [2, 0, 0], [3, 0, 0]], float)
    assert dtw_symmetric1(a, a) == 0.0
    b = a + np.array([0, 0, 1.0])
    assert abs(dtw_symmetric1(a, b) - 4.0) < 1e-12  # diagonal path, 4 unit costs
    # python reference of symmetric1 on a ragged pair
    x, y = np.random.RandomState(0).rand(7, 3), np.random.RandomState(1).rand(5, 3)
    D = np.full((7, 5), np.inf)
    for i in range(7):
        for j in range(5):
            d = np.linalg.norm(x[i] - y[j])
            best = 0.0 if i == j == 0 else min(D[i - 1, j - 1] if i and j else np.inf, D[i - 1, j] if i else np.inf,
                                              D[i, j - 1] if j else np.inf)
            D[i, j] = d + best
    assert abs(dtw_symmetric1(x, y) - D[-1, -1]) < 1e-12
    # a window that forbids every path -> inf
    w = np.zeros((4, 4), np.uint8)
    assert np.isinf(dtw_symmetric1(a, a, w))
    # identical tours -> 1.0; single-episode tour == exp(-dtw/(n*3))
    path = [{"position": [0.25 * i, 0, 0], "phase": "agent", "episode_id": f"e{i // 4}"} for i in range(12)]
    assert compute_tour_ndtw({"t": path}, {"t": path}) == 1.0
    g = [{"position": [0.25 * i, 0, 0], "phase": "agent", "episode_id": f"e{i // 3}"} for i in range(6)]
    ag = [{"position": [0.25 * i, 0.5, 0], "phase": "agent", "episode_id": f"e{i // 3}"} for i in range(6)]
    # diagonal path is optimal and passes through both alignment points: 6 * 0.5
    assert abs(compute_tour_ndtw({"t": ag}, {"t": g}) - np.exp(-3.0 / (6 * 3.0))) < 1e-12
    # the reference weights tours by episode TRANSITIONS (tour_ndtw.py:8-16), so a split made of
    # single-episode tours divides by zero there too; kept
    one = [{"position": [0.25 * i, 0, 0], "phase": "agent", "episode_id": "e"} for i in range(4)]
    with pytest.raises(ZeroDivisionError):
        compute_tour_ndtw({"t": one}, {"t": one})
    # the alignment window pins episode boundaries: columns of alignment points admit one row only
    win = window_align(5, 5, [(1, 2)])
    assert win[:, 2].tolist() == [0, 1, 0, 0, 0] and win[:, 0].min() == 1
    assert novel_only([1, 1, 2, 2, 3]) == [1, 2, 3]


def test_synthetic_env_protocol_and_sharding():
    from ivln_ce_amd.envs import SyntheticVectorEnv

    cfg = get_config()
    envs = SyntheticVectorEnv(cfg, num_envs=2, n_episodes=2, min_len=3, max_len=4)
    obs = envs.reset()
    assert obs[0]["depth"].shape == (256, 256, 1) and obs[0]["depth"].dtype == np.float32
    assert obs[0]["world_robot_orientation"].dtype == np.float64 and obs[0]["instruction"].shape == (200,)
    done_count = 0
    for _ in range(20):
        expert = [int(o["shortest_path_sensor"][0]) for o in obs]
        out = envs.step(expert)
        obs = [o[0] for o in out]
        done_count += sum(o[2] for o in out)
        for o in out:
            if o[2]:  # the expert's own path: nDTW == SDTW == 1 (fastdtw of identical paths is exact)
                assert abs(o[3]["ndtw"] - 1.0) < 1e-12 and abs(o[3]["sdtw"] - 1.0) < 1e-12
            else:
                assert o[3]["ndtw"] == 0.0
    assert done_count >= 4
    from ivln_ce_amd.tour_ndtw import compute_tour_ndtw

    # following the expert reproduces the gt paths: t-nDTW == 1 on the tours played once
    agent, gt = envs.dtw_data(), envs.gt_paths()
    assert set(agent) == set(gt) and abs(compute_tour_ndtw(agent, gt) - 1.0) < 1e-9
    a = SyntheticVectorEnv(cfg, num_envs=2, rank=1, world=4)
    assert [e.idx for e in a._envs] == [1, 5]


def test_dtw_recurrence_matches_reference_golden():
    """csrc/dtw.cpp (the DTW under nDTW / SDTW / t-nDTW) against distances produced by the reference's own
    exact DTW (habitat_extensions/utils.py:155-221), tests/golden/gen_dtw_golden.py; 1e-9 relative."""
    from ivln_ce_amd.measures import ndtw
    from ivln_ce_amd.tour_ndtw import dtw_symmetric1

    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "dtw.npz"))
    for k in range(int(g["n_cases"])):
        x, y = g[f"x_{k}"], g[f"y_{k}"]
        d = dtw_symmetric1(x, y)
        assert abs(d - float(g[f"d_{k}"])) <= 1e-9 * max(1.0, abs(float(g[f"d_{k}"]))), (k, d, float(g[f"d_{k}"]))
        assert abs(ndtw(x.tolist(), y.tolist(), 3.0, fdtw=False) - float(g[f"ndtw_{k}"])) < 1e-12


def test_fastdtw_restatement_known_answers():
    """FastDTW (third-party, unpinned): identical paths -> 0; sequences shorter than radius+2 -> exact DTW;
    never below the exact DTW distance and close to it on smooth trajectories; NDTW/SDTW measure plumbing."""
    from ivln_ce_amd.measures import NDTW, SDTW, fastdtw, ndtw
    from ivln_ce_amd.tour_ndtw import dtw_symmetric1

    rng = np.random.RandomState(3)
    a = np.cumsum(rng.randn(50, 3) * 0.25, axis=0)
    assert fastdtw(a, a)[0] == 0.0
    s1, s2 = rng.rand(2, 3), rng.rand(2, 3)
    assert abs(fastdtw(s1, s2)[0] - dtw_symmetric1(s1, s2)) < 1e-12
    b = a[::2] + rng.randn(25, 3) * 0.05
    exact, fast = dtw_symmetric1(a, b), fastdtw(a, b)[0]
    assert exact - 1e-9 <= fast <= 1.25 * exact
    path = fastdtw(a, b)[1]
    assert path[0] == (0, 0) and path[-1] == (49, 24)
    m = NDTW(3.0, fdtw=False)
    m.reset_metric(b.tolist(), a[0])
    for p in a[1:]:
        m.update_metric(p)
        m.update_metric(p)  # an unchanged position is not appended (measures.py:196-200)
    assert len(m.locations) == 50 and abs(m.get_metric() - ndtw(a.tolist(), b.tolist(), 3.0)) < 1e-12
    assert SDTW.get_metric(1.0, m.get_metric()) == m.get_metric() and SDTW.get_metric(0.0, m.get_metric()) == 0.0


def test_prefetch_loader_preserves_order_and_surfaces_errors():
    from ivln_ce_amd.trainers import PrefetchLoader

    def batches(n, fail_at=None):
        for i in range(n):
            if i == fail_at:
                raise RuntimeError("broken record")
            yield ({"occupancy_map": torch.full((4, 2), float(i))}, torch.full((4, 1), i), torch.ones(4, 1, dtype=torch.uint8),
                   torch.zeros(2, 2, dtype=torch.long), torch.ones(2, 2))

    got = [int(b[1][0, 0]) for b in PrefetchLoader(batches(7), torch.device("cpu"))]
    assert got == list(range(7))
    out = next(iter(PrefetchLoader(batches(1), torch.device("cpu"))))
    assert out[0]["occupancy_map"].dtype == torch.float32 and out[3] is None
    with pytest.raises(RuntimeError, match="broken record"):
        list(PrefetchLoader(batches(5, fail_at=2), torch.device("cpu")))


def _deser(d):
    import torch

    dt = getattr(torch, d["dtype"].split(".")[1])
    return torch.tensor(d["data"], dtype=dt).view(d["shape"])


def test_tour_sampler_and_collate_match_reference_golden(golden_dir):
    """TourSampler row order / tour starts / transposition / drop_last cut and the tour collate against the
    reference's own tour_dataset.py (tests/golden/gen_tour_golden.py; the bin packing itself is a restated
    third-party function, unpinned)."""
    import json

    import numpy as np
    import torch

    from ivln_ce_amd.tour_batches import TourSampler, to_constant_bin_number, tour_collate

    g = json.load(open(os.path.join(golden_dir, "tour_batches.json")))
    for c in g["sampler"]:
        table, nxt = {}, 1
        for t, n in enumerate(c["sizes"]):
            table[f"tour{t}"] = list(range(nxt, nxt + n))
            nxt += n
        np.random.seed(c["seed"])
        s = TourSampler(table, batch_size=c["batch_size"], shuffle=c["shuffle"], drop_last=c["drop_last"])
        assert s.batched_idxs == c["batches"], c
        assert sorted(s.get_tour_done_idxs()) == c["tour_done_idxs"], c
        assert s.get_num_batches() == len(c["batches"]) == len(s)
        assert [list(b) for b in s] == c["iterated"] and list(s) == []  # single pass
    # greedy partition: heaviest first into the lightest bin, ties to the lowest index
    bins = to_constant_bin_number({"a": 5, "b": 3, "c": 4, "d": 2, "e": 6, "f": 1, "g": 3}, 3)
    assert [list(b) for b in bins] == [["e", "d"], ["a", "g"], ["c", "b", "f"]]
    co = g["collate"]
    samples = [({k: _deser(v) for k, v in s["obs"].items()}, _deser(s["prev"]), _deser(s["expert"]), _deser(s["weights"]),
                _deser(s["tour"])) for s in co["samples"]]
    obs_b, prev_b, ep_b, tour_b, corr_b, w_b = tour_collate(samples)
    ref = co["out"]
    for k in ref["obs"]:
        r = _deser(ref["obs"][k])
        assert obs_b[k].dtype == r.dtype and torch.equal(obs_b[k], r), k
    for got, name in [(prev_b, "prev"), (ep_b, "episode"), (tour_b, "tour"), (corr_b, "expert"), (w_b, "weights")]:
        r = _deser(ref[name])
        assert got.dtype == r.dtype and got.shape == r.shape and torch.equal(got, r), name


def test_tour_trajectory_dataset_and_store_tour_index(tmp_path):
    import numpy as np
    import torch

    from ivln_ce_amd.tour_batches import TourSampler, TourTrajectoryDataset, tour_collate
    from ivln_ce_amd.trainers import TrajectoryStore

    store = TrajectoryStore(str(tmp_path / "traj"))
    table = {"t0": [1, 2, 3], "t1": [4, 5], "t2": [6, 7, 8]}
    rs = np.random.RandomState(0)
    for tour, idxs in table.items():
        for i in idxs:
            T = 2 + i % 3
            expert = rs.randint(0, 4, size=T)
            store.put(i, {"feat": rs.rand(T, 4).astype(np.float32)}, np.concatenate([[0], expert[:-1]]), expert, tour_id=tour)
    store.put_tour_index(table)
    assert store.get_tour_index() == table and len(store) == 8  # the table is not a trajectory record
    ds = TourTrajectoryDataset(store, use_iw=True, inflection_weight_coef=3.2)
    with pytest.raises(AssertionError):
        ds[1]
    sampler = TourSampler({k: list(v) for k, v in table.items()}, batch_size=2, shuffle=False, drop_last=False)
    ds.set_tour_done_idxs(sampler.get_tour_done_idxs())
    assert sampler.get_tour_done_idxs() == {1, 4, 6}
    obs, prev, expert, w, tour = ds[4]
    assert tour.tolist() == [0] + [1] * (len(prev) - 1) and ds[5][4].tolist() == [1] * len(ds[5][1])
    exp_w = [3.2] + [3.2 if expert[i] != expert[i - 1] else 1.0 for i in range(1, len(expert))]
    assert torch.allclose(w, torch.tensor(exp_w))
    loader = torch.utils.data.DataLoader(ds, batch_sampler=sampler, collate_fn=tour_collate)
    batches = list(loader)
    assert len(batches) == len(sampler.batched_idxs)
    obs_b, prev_b, ep_b, tour_b, corr_b, w_b = batches[0]
    T, N = corr_b.shape
    assert N == 2 and obs_b["feat"].shape == (T * N, 4) and ep_b.view(T, N)[0].tolist() == [0, 0]
    assert tour_b.view(T, N)[0].tolist() == [0, 0]  # both rows open a tour in the first batch
    store.clear()
    assert len(store) == 0 and store.get_tour_index() == {}


def test_greedy_bin_packing_properties():
    """to_constant_bin_number (restated `binpacking` function): every key lands in exactly one bin, the bin count
    is constant, and the greedy rule bounds the spread of the bin loads by the largest item."""
    import numpy as np

    from ivln_ce_amd.tour_batches import TourSampler, to_constant_bin_number

    rs = np.random.RandomState(3)
    for n_bins in (1, 2, 5, 8):
        w = {f"t{i}": int(v) for i, v in enumerate(rs.randint(1, 40, size=rs.randint(n_bins, 60)))}
        bins = to_constant_bin_number(w, n_bins)
        assert len(bins) == n_bins
        keys = [k for b in bins for k in b]
        assert sorted(keys) == sorted(w) and all(b[k] == w[k] for b in bins for k in b)
        loads = [sum(b.values()) for b in bins]
        assert max(loads) - min(loads) <= max(w.values())
    # the sampler never repeats or invents a record, whatever the tour sizes
    table, nxt = {}, 1
    for t, n in enumerate(rs.randint(1, 9, size=11)):
        table[f"tour{t}"] = list(range(nxt, nxt + int(n)))
        nxt += int(n)
    np.random.seed(1)
    s = TourSampler({k: list(v) for k, v in table.items()}, batch_size=4, shuffle=True, drop_last=False)
    flat = [i for b in s.batched_idxs for i in b]
    assert sorted(flat) == list(range(1, nxt)) and len(s.get_tour_done_idxs()) == len(table)
    # row r of consecutive batches walks bin r in order: each tour's episodes stay contiguous within their row
    rows = [[b[r] for b in s.batched_idxs if len(b) == 4] for r in range(4)]  # full batches: position r = bin r
    tour_of = {i: t for t, idxs in table.items() for i in idxs}
    for row in rows:
        seen, last = set(), None
        for i in row:
            t = tour_of[i]
            if t != last:
                assert t not in seen, "a tour was split inside its row"
                seen.add(t)
                last = t


def test_bench_gpus_flag_spawns_that_many_ranks():
    """`python bench.py --gpus 2` with no launcher around it must start two ranks itself (torch.distributed.run,
    before the parent touches a GPU) and rank 0's line must say n_gpus 2.  --plumbing-only keeps the ranks off the
    GPU (there is none here): rendezvous, barrier and the max-over-ranks reduction still run."""
    import json
    import subprocess
    import sys

    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--plumbing-only"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [json.loads(line) for line in r.stdout.splitlines() if line.startswith("{")]
    assert len(lines) == 1 and lines[0]["n_gpus"] == 2 and lines[0]["plumbing_only"] is True
    assert lines[0]["value"] is None  # never mistaken for a measurement
    # under a launcher (WORLD_SIZE set) the flag does not spawn again
    r1 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--plumbing-only"], env=env,
                        capture_output=True, text=True, timeout=300)
    assert r1.returncode == 0 and json.loads(r1.stdout.splitlines()[-1])["n_gpus"] == 1


def _np(d):
    import numpy as np

    return np.array(d["data"], dtype=np.dtype(d["dtype"])).reshape(d["shape"])


@pytest.mark.parametrize("case", ["teacher_forcing_unique", "beta_quarter", "policy_only"])
def test_dagger_rollout_logic_matches_reference_run(case, tmp_path, monkeypatch):
    """A18: `DaggerTrainer._update_dataset` against what the REFERENCE's own `_update_dataset` + `_pause_envs`
    stored, stepped and fed to `policy.act` on the same scripted env / stand-in policy
    (tests/golden/gen_rollout_golden.py -> rollout_golden.json): beta-mixing, expert -1 skip, pause compaction at
    beta == 1, stored-trajectory contents.  Host logic only - runs on the CPU."""
    import json

    import numpy as np
    import torch

    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import rollout_script as RS

    from ivln_ce_amd import trainers
    from ivln_ce_amd.config import get_config

    g = json.load(open(os.path.join(ROOT, "tests", "golden", "rollout_golden.json")))[case]
    p, data_it, update_size, seed = RS.CASES[case]
    assert (g["p"], g["data_it"], g["update_size"], g["seed"]) == (p, data_it, update_size, seed)
    cfg = get_config(opts=["IL.DAGGER.p", p, "IL.DAGGER.update_size", update_size])
    tr = trainers.DaggerTrainer.__new__(trainers.DaggerTrainer)  # no GPU needed for the host logic
    tr.config, tr.device, tr.obs_transforms = cfg, torch.device("cpu"), []
    tr.rank, tr.local_rank, tr.world = 0, 0, 1
    tr.policy = RS.ScriptedPolicy()
    tr.store = trainers.TrajectoryStore(str(tmp_path / "traj"))
    envs = RS.ScriptedEnvs(RS.SCRIPTS)
    monkeypatch.setattr(trainers, "construct_envs", lambda *a, **k: envs)
    torch.manual_seed(seed)
    n = tr._update_dataset(data_it)
    assert n == len(g["records"]) == len(tr.store)
    assert envs.action_log == g["env_actions"]          # what the simulator was told to do, step by step
    assert tr.policy.calls == g["policy_calls"]         # rows / masks / previous actions / compacted state per act()
    for i, rec in enumerate(g["records"]):
        obs, prev, oracle = tr.store.get(i)
        assert sorted(obs) == sorted(rec["obs"]), (sorted(obs), sorted(rec["obs"]))
        for k, v in rec["obs"].items():
            ref = _np(v)
            assert obs[k].dtype == ref.dtype and np.array_equal(obs[k], ref), (i, k)
        assert np.array_equal(prev, _np(rec["prev_actions"])) and np.array_equal(oracle, _np(rec["oracle_actions"]))


def test_collate_block_shuffle_and_iw_dataset_match_reference_run(tmp_path):
    """A19: `collate_fn`, `_block_shuffle`, `IWTrajectoryDataset` against the reference's own functions
    (gen_rollout_golden.py -> collate_golden.json): padded time-major batch, inflection weights, and the ORDER a
    seeded dataset yields trajectories in (incl. the reference's quirk of sorting a preload by len(obs dict))."""
    import json
    import random

    import numpy as np
    import torch

    from ivln_ce_amd import trainers

    g = json.load(open(os.path.join(ROOT, "tests", "golden", "collate_golden.json")))
    t = lambda d: torch.from_numpy(_np(d))  # noqa: E731
    samples = [({k: t(v) for k, v in s["obs"].items()}, t(s["prev"]), t(s["oracle"]), t(s["weights"]))
               for s in g["collate"]["samples"]]
    obs_b, prev_b, nd_b, corr_b, w_b = trainers.collate_fn(samples)
    out = g["collate"]["out"]
    for k, v in out["obs"].items():
        assert torch.equal(obs_b[k], t(v)) and obs_b[k].dtype == t(v).dtype, k
    for got, key in [(prev_b, "prev"), (nd_b, "not_done"), (corr_b, "oracle"), (w_b, "weights")]:
        assert torch.equal(got, t(out[key])) and got.dtype == t(out[key]).dtype, key
    for c in g["block_shuffle"]:
        random.seed(c["seed"])
        assert trainers._block_shuffle(list(range(c["n"])), c["block"]) == c["out"]
    store = trainers.TrajectoryStore(str(tmp_path / "iw"))
    for i, tj in enumerate(g["trajectories"]):
        store.put(i, {k: _np(v) for k, v in tj["obs"].items()}, _np(tj["prev"]), _np(tj["oracle"]))
    for c in g["iw_dataset"]:
        random.seed(c["seed"])
        ds = trainers.IWTrajectoryDataset(store, c["use_iw"], c["coef"], batch_size=c["batch_size"])
        assert ds.length == c["length"]
        got = [{"id": int(obs["instruction"][0, 0]) - 1, "weights": [float(x) for x in w]} for obs, _, _, w in ds]
        assert got == c["yielded"]


def _iter_trainer(cls, cfg, tmp_path, with_rgb=True):
    """A trainer of this package wired to the scripted stand-ins of tests/golden/iterative_script.py, on the CPU
    (host logic only: the policy / mapper stand-ins carry no arithmetic)."""
    import torch

    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import iterative_script as IS

    from ivln_ce_amd import trainers

    tr = cls.__new__(cls)
    tr.config, tr.device = cfg, torch.device("cpu")
    tr.rank, tr.local_rank, tr.world = 0, 0, 1
    tr.obs_transforms = [IS.ScriptedMapper()]
    tr.policy = IS.ScriptedIterativePolicy(with_rgb=with_rgb)
    tr.store = trainers.TrajectoryStore(str(tmp_path / "traj"))
    return tr, IS


@pytest.mark.parametrize("case", ["tf_unique_oracle", "beta_half_oracle", "policy_no_oracle"])
def test_iterative_collection_matches_reference_run(case, tmp_path, monkeypatch):
    """A18: `IterativeCollectionDaggerTrainer._update_dataset(save_tour_idx_data=True)` - the trainer every MapCMA
    YAML names - against what the REFERENCE's own `_update_dataset`, `masks_to_tensors`, `add_map_to_observations` and
    `_pause_iterative_envs` stored, stepped and fed to `policy.act_iterative` on the same scripted iterative env
    (oracle phases, four masks, tour ids, an expert -1, env pauses; tests/golden/gen_iterative_golden.py ->
    iterative_golden.json): records numbered from 1, the tour table under "0", only agent-phase steps stored, maps
    carried across the episodes of a tour, a second collection continuing the numbering, quirk Q12 included."""
    import json

    import numpy as np
    import torch

    from ivln_ce_amd import trainers
    from ivln_ce_amd.config import get_config

    g = json.load(open(os.path.join(ROOT, "tests", "golden", "iterative_golden.json")))["collect"][case]
    cfg = get_config(opts=["IL.DAGGER.p", g["p"], "IL.DAGGER.update_size", g["update_size"]])
    tr, IS = _iter_trainer(trainers.IterativeCollectionDaggerTrainer, cfg, tmp_path)
    assert IS.COLLECT_CASES[case][:3] == (g["p"], g["runs"][0]["data_it"], g["update_size"])
    for run in g["runs"]:
        tr.obs_transforms = [IS.ScriptedMapper()]
        envs = IS.ScriptedVectorEnv(IS.scripts(), iterative=True, auto_reset=True, oracle_phases=g["oracle_phases"])
        monkeypatch.setattr(trainers, "construct_envs", lambda *a, **k: envs)
        del tr.policy.calls[:], tr.policy.deleted[:]
        torch.manual_seed(run["seed"])
        table = tr._update_dataset(run["data_it"], save_tour_idx_data=True)
        assert table == run["tour_table"] == tr.store.get_tour_index()
        assert envs.action_log == run["env_actions"]           # what the simulator was told to do, step by step
        assert tr.policy.calls == run["policy_calls"]          # four masks / previous actions / state rows / map per call
        assert tr.policy.deleted == run["deleted_batch_idx"]   # rows reported to net.delete_batch_idx on pause
        assert len(tr.store) == len(run["records"]) and tr.store.entries() == len(run["records"]) + 1
        for idx, rec in run["records"].items():
            obs, prev, oracle = tr.store.get(int(idx))
            assert sorted(obs) == sorted(rec["obs"]), (idx, sorted(obs), sorted(rec["obs"]))
            for k, v in rec["obs"].items():
                ref = _np(v)
                assert obs[k].dtype == ref.dtype and np.array_equal(obs[k], ref), (idx, k)
            assert np.array_equal(prev, _np(rec["prev_actions"])) and np.array_equal(oracle, _np(rec["oracle_actions"]))


@pytest.mark.parametrize("case", ["episodic", "iterative_tour_maps", "iterative_episode_maps", "iterative_no_oracle"])
def test_eval_loops_match_reference_run(case, tmp_path, monkeypatch):
    """A18: `_eval_checkpoint` (episodic) and `_eval_checkpoint_iterative` against runs of the REFERENCE's own loops
    (base_il_trainer.py:313-583, 585-928) on the scripted env: the actions the envs received, every `reset_at`, what
    each `act` / `act_iterative` call saw (masks, previous actions, compacted state rows, the map under either
    ITERATIVE_MAP_RESET), rows reported on pause, the report files and TensorBoard scalars, and the ARGUMENTS handed
    to `compute_tour_ndtw` (dtw-python itself is absent from the image)."""
    import json

    from ivln_ce_amd import trainers
    from ivln_ce_amd.config import get_config

    g = json.load(open(os.path.join(ROOT, "tests", "golden", "iterative_golden.json")))["eval"][case]
    gt = {"val_seen": {"T0": [[0.0, 0.0, 0.0]], "T1": [[1.0, 0.0, 0.0]]}}
    gt_file = str(tmp_path / "gt.json")
    json.dump(gt, open(gt_file, "w"))
    res_dir = str(tmp_path / "results")
    cfg = get_config(opts=["RESULTS_DIR", res_dir, "EVAL.SPLIT", "val_seen", "EVAL.ITERATIVE_MAP_RESET", g["map_reset"],
                           "EVAL.ITERATIVE_GT_PATHS", gt_file, "EVAL.SAVE_RESULTS", True,
                           "TASK_CONFIG.ENVIRONMENT.ITERATIVE.ENABLED", g["iterative"], "VIDEO_OPTION", []])
    tr, IS = _iter_trainer(trainers.BaseVLNCETrainer, cfg, tmp_path, with_rgb=False)
    tr._load_eval_policy = lambda *a, **k: None
    envs = IS.ScriptedVectorEnv(IS.scripts(), iterative=g["iterative"], auto_reset=False, oracle_phases=g["oracle_phases"])
    monkeypatch.setattr(trainers, "construct_envs", lambda *a, **k: envs)
    ndtw_calls = []

    def record(agent_paths, gt_paths, success_distance):
        ndtw_calls.append({"agent_paths": json.loads(json.dumps(agent_paths)), "gt_paths": gt_paths,
                           "success_distance": success_distance})
        return 0.4242

    monkeypatch.setattr(trainers, "compute_tour_ndtw", record)

    class Writer:
        scalars = []

        def add_scalar(self, k, v, step):
            self.scalars.append([k, float(v), int(step)])

    w = Writer()
    res = tr._eval_checkpoint("data/checkpoints/ckpt.3.pth", w, checkpoint_index=0)
    assert envs.action_log == g["env_actions"]
    assert [list(x) for x in envs.reset_at_log] == g["reset_at"]
    assert tr.policy.calls == g["policy_calls"]
    assert tr.policy.deleted == g["deleted_batch_idx"]
    assert ndtw_calls == g["tour_ndtw_calls"]
    assert sorted(os.listdir(res_dir)) == sorted(g["files"])
    for f, content in g["files"].items():
        assert json.load(open(os.path.join(res_dir, f))) == content, f
    assert w.scalars == g["scalars"]
    stats = g["files"][("iterative_stats" if g["iterative"] else "stats") + "_ckpt_3_val_seen.json"]
    assert {k: res[k] for k in stats} == stats and res["episodes"] == 10


def test_tour_ndtw_matches_reference_bookkeeping():
    """t-nDTW against the reference's own `compute_tour_ndtw` on seeded random tours (oracle phases, in-place turns,
    one-step episodes that close the alignment window; tests/golden/gen_tour_ndtw_golden.py): the reference's
    filtering / alignment / window / weighting ran around a plain numpy DTW standing in for dtw-python."""
    import json

    from ivln_ce_amd.tour_ndtw import compute_tour_ndtw

    for c in json.load(open(os.path.join(ROOT, "tests", "golden", "tour_ndtw.json"))):
        assert abs(compute_tour_ndtw(c["agent"], c["gt"], c["success_distance"]) - c["score"]) < 1e-12


def test_synthetic_iterative_env_walks_the_reference_phases():
    """The synthetic env's iterative protocol (environments.py:36-356 restated on a kinematic agent): 7-tuple steps,
    agent -> oracle_goal -> oracle_start -> agent, `agent_episode_done` in every oracle step, `produce_action` off
    while the oracle drives, `tour_done` only from a reset that changes tour, `dtw_data` at episode ends; following
    the expert (env 0) or stopping early (env 1) and letting the oracle do its part scores a sensible t-nDTW against
    the env's own expert paths."""
    from ivln_ce_amd.envs import SyntheticVectorEnv
    from ivln_ce_amd.tour_ndtw import compute_tour_ndtw

    cfg = get_config(opts=["ENV_NAME", "VLNCEIterativeEnv"])
    for auto_reset in (True, False):
        envs = SyntheticVectorEnv(cfg, num_envs=2, n_episodes=5, episodes_per_tour=2, min_len=3, max_len=6,
                                  auto_reset_done=auto_reset)
        assert envs.iterative
        out = envs.reset()
        obs = [o[0] for o in out]
        assert all(o[1] is True and o[2] is True for o in out)  # first reset: "tour done", the agent acts
        dtw, phases_seen, tour_resets, finished = {}, set(), 0, [0, 0]
        acting = [True, True]
        for _ in range(600):
            if min(finished) >= 5:
                break
            eps = envs.current_episodes()
            acts = [int(o["shortest_path_sensor"][0]) for o in obs]
            if acting[1] and float(obs[1]["progress"][0]) > 0.3:
                acts[1] = 0  # env 1 stops early: the oracle has to convey it to the goal
            out = envs.step(acts)
            assert all(len(o) == 7 for o in out)
            for i, (o, _, agent_done, sim_done, tour_done, produce, info) in enumerate(out):
                if not acting[i]:   # an oracle step: the agent's episode stays "done", its action was ignored
                    assert agent_done and len(info) <= 1
                if agent_done or sim_done:
                    assert "dtw_data" in info
                    phases_seen.update(p["phase"] for p in info["dtw_data"])
                if sim_done:
                    if finished[i] < 5:
                        dtw.setdefault(eps[i].tour_id, []).extend(info["dtw_data"])
                    finished[i] += 1
                    if not auto_reset:
                        o, tour_done, produce = envs.reset_at(i)[0]
                    nxt = envs.current_episodes()[i]
                    assert tour_done == (nxt.tour_id != eps[i].tour_id)
                    tour_resets += int(tour_done)
                else:
                    assert tour_done is False
                obs[i], acting[i] = o, bool(produce)
        assert min(finished) >= 5 and phases_seen == {"agent", "oracle_goal", "oracle_start"} and tour_resets >= 4
        gt = envs.gt_paths()
        score = compute_tour_ndtw(dtw, {k: gt[k] for k in dtw})
        assert 0.6 < score < 1.0, score


def test_requeue_resumes_at_the_checkpointed_iteration_and_epoch(tmp_path, monkeypatch):
    """a run requeued from a checkpoint written after (dagger_it = 1, epoch = 1) continues with epoch 2 of
    iteration 1 on the stored trajectories (no new collection), then runs iteration 2 from epoch 0 with a fresh
    collection - and never touches the checkpoints of what was already done."""
    import numpy as np
    import torch

    from ivln_ce_amd import trainers
    from ivln_ce_amd.config import get_config

    cfg = get_config(opts=["IL.DAGGER.iterations", 3, "IL.epochs", 3, "IL.batch_size", 2, "IL.load_from_ckpt", True,
                           "IL.is_requeue", True])
    tr = trainers.DaggerTrainer.__new__(trainers.DaggerTrainer)
    tr.config, tr.device = cfg, torch.device("cpu")
    tr.rank, tr.local_rank, tr.world = 0, 0, 1
    tr.step_id, tr.start_epoch, tr.start_dagger_it = 40, 2, 1   # what _initialize_policy restores from the checkpoint
    tr.store = trainers.TrajectoryStore(str(tmp_path / "traj"))
    for i in range(4):
        tr.store.put(i, {"feat": np.zeros((3, 2), np.float32), "instruction": np.ones((3, 4), np.int64)},
                     np.zeros(3, np.int64), np.array([1, 1, 0]))
    collected, saved = [], []
    monkeypatch.setattr(tr, "_update_dataset", lambda it, **k: collected.append(it), raising=False)
    monkeypatch.setattr(tr, "_update_agent", lambda *a, **k: (0.0, 0.0, 0.0), raising=False)
    monkeypatch.setattr(tr, "save_checkpoint", lambda name, dagger_it, epoch, step_id: saved.append((name, dagger_it, epoch)),
                        raising=False)
    log = tr._train_loop()
    assert collected == [3]                       # only iteration 2 collects (data_it = dagger_it + 1 after a load)
    assert saved == [("ckpt.5.pth", 1, 2), ("ckpt.6.pth", 2, 0), ("ckpt.7.pth", 2, 1), ("ckpt.8.pth", 2, 2)]
    assert [(e["dagger_it"], e["epoch"]) for e in log] == [(1, 2)] * 2 + [(2, 0)] * 2 + [(2, 1)] * 2 + [(2, 2)] * 2
    assert tr.step_id == 48
