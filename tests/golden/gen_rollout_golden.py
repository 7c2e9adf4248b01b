"""Golden vectors for the host-side logic of the DAgger trainer, produced by the REFERENCE's own code
(build container only; /root/reference never travels):

  rollout_golden.json   `DaggerTrainer._update_dataset` (ivlnce_baselines/trainers/dagger_trainer.py:251-504)
                        + `BaseVLNCETrainer._pause_envs` (common/base_il_trainer.py:221-256) driven by the scripted
                        env / stand-in policy of rollout_script.py: beta-mixing, the expert -1 skip, pause
                        compaction at beta == 1, what a stored trajectory contains - the stored records, the
                        actions the envs received and what every `policy.act` call saw
  collate_golden.json   `collate_fn`, `_block_shuffle` and `IWTrajectoryDataset` (dagger_trainer.py:42-234): padded
                        time-major batch, inflection weights, and the order in which a seeded dataset yields
                        trajectories

lmdb / msgpack_numpy / tensorflow / habitat are absent from the image: they are replaced by in-memory stand-ins
that carry no arithmetic (a dict-backed key-value store, identity (un)packing, empty modules)."""
import copy
import importlib.util
import json
import os
import random
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_shim  # noqa: E402
import rollout_script as RS  # noqa: E402

_ref_shim.install()
REF = _ref_shim.REF


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


# ---- in-memory lmdb ---------------------------------------------------------------------------
class _Txn:
    def __init__(self, db):
        self.db = db

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False

    def put(self, k, v, overwrite=True):
        self.db[k] = v

    def get(self, k):
        return self.db[bytes(k)]

    def commit(self):
        pass


class _Env:
    def __init__(self, db):
        self.db = db

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False

    def stat(self):
        return {"entries": len(self.db)}

    def begin(self, write=False, buffers=False):
        return _Txn(self.db)


_DBS = {}
_mod("lmdb", open=lambda path, **kw: _Env(_DBS.setdefault(path, {})))
_mod("msgpack_numpy", packb=lambda obj, use_bin_type=True: copy.deepcopy(obj),
     unpackb=lambda obj, raw=False: copy.deepcopy(obj))
_mod("tensorflow")

# ---- habitat pieces the trainers import ---------------------------------------------------------
_mod("habitat.utils")
_mod("habitat.utils.visualizations")
_mod("habitat.utils.visualizations.utils", append_text_to_image=None)


class BaseILTrainer:
    def __init__(self, config=None):
        self.config = config


_mod("habitat_baselines.common.base_il_trainer", BaseILTrainer=BaseILTrainer)
_mod("habitat_baselines.common.environments", get_env_class=lambda name: None)
ot = sys.modules["habitat_baselines.common.obs_transformers"]
ot.apply_obs_transforms_batch = lambda batch, ts: batch
ot.apply_obs_transforms_obs_space = lambda space, ts: space
ot.get_active_obs_transforms = lambda cfg: []
_mod("habitat_baselines.common.tensorboard_utils", TensorboardWriter=None)
_mod("habitat_baselines.rl.ddppo.algo")
_mod("habitat_baselines.rl.ddppo.algo.ddp_utils", is_slurm_batch_job=lambda: False)
_mod("habitat_extensions")
_mod("habitat_extensions.tour_ndtw", compute_tour_ndtw=None)
_mod("habitat_extensions.utils", generate_video=None, observations_to_image=None)
_mod("ivlnce_baselines.common.mapping_module.visualize_semantic_map", append_image_horizontally=None,
     append_image_vertically=None)
_ENVS = {}
_mod("ivlnce_baselines.common.env_utils", construct_envs=lambda cfg, cls=None, **kw: _ENVS["next"],
     construct_envs_auto_reset_false=lambda cfg, cls=None, **kw: _ENVS["next"])
pk = types.ModuleType("ivlnce_baselines.trainers")
pk.__path__ = [os.path.join(REF, "ivlnce_baselines", "trainers")]
sys.modules["ivlnce_baselines.trainers"] = pk

spec = importlib.util.spec_from_file_location("ivlnce_baselines.trainers.dagger_trainer",
                                              os.path.join(REF, "ivlnce_baselines", "trainers", "dagger_trainer.py"))
ref = importlib.util.module_from_spec(spec)
sys.modules[spec.name] = ref
spec.loader.exec_module(ref)


def ser(x):
    if isinstance(x, np.ndarray):
        return {"dtype": str(x.dtype), "shape": list(x.shape), "data": x.flatten().tolist()}
    if torch.is_tensor(x):
        return {"dtype": str(x.dtype).replace("torch.", ""), "shape": list(x.shape), "data": x.flatten().tolist()}
    raise TypeError(type(x))


def rollout_case(name):
    p, data_it, update_size, seed = RS.CASES[name]
    cfgmod = sys.modules["ivln_cfg_for_shim"]
    cfg = cfgmod.get_config(opts=["IL.DAGGER.p", p, "IL.DAGGER.update_size", update_size,
                                  "IL.DAGGER.lmdb_features_dir", f"mem://{name}", "IL.DAGGER.lmdb_fp16", False,
                                  "IL.DAGGER.lmdb_commit_frequency", 3])
    tr = ref.DaggerTrainer.__new__(ref.DaggerTrainer)
    tr.config, tr.device, tr.obs_transforms = cfg, torch.device("cpu"), []
    tr.lmdb_features_dir = f"mem://{name}"
    tr.policy = RS.ScriptedPolicy()
    envs = RS.ScriptedEnvs(RS.SCRIPTS)
    _ENVS["next"] = envs
    torch.manual_seed(seed)
    tr._update_dataset(data_it)
    db = _DBS[f"mem://{name}"]
    records = []
    for i in range(len(db)):
        obs, prev, oracle = db[str(i).encode()]
        records.append({"obs": {k: ser(np.asarray(v)) for k, v in sorted(obs.items())}, "prev_actions": ser(prev),
                        "oracle_actions": ser(oracle)})
    return {"p": p, "data_it": data_it, "update_size": update_size, "seed": seed, "records": records,
            "env_actions": envs.action_log, "policy_calls": tr.policy.calls}


def collate_cases():
    g = torch.Generator().manual_seed(5)
    lens = [3, 5, 2, 4]
    samples = []
    for T in lens:
        obs = {"feat": torch.randn(T, 2, 3, generator=g), "instruction": torch.randint(0, 9, (T, 4), generator=g)}
        prev = torch.randint(0, 4, (T,), generator=g)
        oracle = torch.randint(0, 4, (T,), generator=g)
        w = torch.where(torch.rand(T, generator=g) < 0.5, torch.tensor(3.2), torch.tensor(1.0))
        samples.append((obs, prev, oracle, w))
    out = ref.collate_fn([({k: v.clone() for k, v in s[0].items()}, s[1].clone(), s[2].clone(), s[3].clone())
                          for s in samples])
    obs_b, prev_b, nd_b, corr_b, w_b = out
    collate = {"samples": [{"obs": {k: ser(v) for k, v in s[0].items()}, "prev": ser(s[1]), "oracle": ser(s[2]),
                            "weights": ser(s[3])} for s in samples],
               "out": {"obs": {k: ser(v) for k, v in obs_b.items()}, "prev": ser(prev_b), "not_done": ser(nd_b),
                       "oracle": ser(corr_b), "weights": ser(w_b)}}

    shuffles = []
    for seed, n, bs in [(0, 10, 3), (1, 7, 2), (2, 4, 8), (3, 0, 2)]:
        random.seed(seed)
        shuffles.append({"seed": seed, "n": n, "block": bs, "out": ref._block_shuffle(list(range(n)), bs)})

    # a small trajectory database in the fake lmdb: record i = [obs, prev_actions, oracle_actions]
    db = _DBS.setdefault("mem://iw", {})
    rs = np.random.RandomState(4)
    trajs = []
    for i in range(11):
        T = int(rs.randint(2, 7))
        oracle = rs.randint(0, 4, size=T).astype(np.int64)
        prev = np.concatenate([[0], oracle[:-1]]).astype(np.int64)
        obs = {"feat": rs.rand(T, 2).astype(np.float32), "instruction": np.tile(np.array([i + 1, 2, 0], np.int64), (T, 1))}
        if i % 3 == 0:  # an extra sensor on some records: the reference sorts by len(obs dict), see below
            pass
        db[str(i).encode()] = [obs, prev, oracle]
        trajs.append({"obs": {k: ser(v) for k, v in obs.items()}, "prev": ser(prev), "oracle": ser(oracle)})
    iw = []
    for seed, use_iw, coef, bs in [(0, True, 3.2, 2), (7, True, 3.2, 3), (3, False, 3.2, 2)]:
        random.seed(seed)
        ds = ref.IWTrajectoryDataset("mem://iw", use_iw, inflection_weight_coef=coef, lmdb_map_size=1e6, batch_size=bs)
        order = []
        for obs, prev, oracle, w in ds:
            order.append({"id": int(obs["instruction"][0, 0]) - 1, "weights": [float(x) for x in w]})
        iw.append({"seed": seed, "use_iw": use_iw, "coef": coef, "batch_size": bs, "length": ds.length, "yielded": order})
    return {"collate": collate, "block_shuffle": shuffles, "trajectories": trajs, "iw_dataset": iw}


if __name__ == "__main__":
    roll = {name: rollout_case(name) for name in RS.CASES}
    json.dump(roll, open(os.path.join(HERE, "rollout_golden.json"), "w"))
    for name, c in roll.items():
        print(name, "records", len(c["records"]), "lens", [r["prev_actions"]["shape"][0] for r in c["records"]],
              "steps", len(c["env_actions"]), "rows per act call", [x["rows"] for x in c["policy_calls"]])
    col = collate_cases()
    json.dump(col, open(os.path.join(HERE, "collate_golden.json"), "w"))
    for c in col["iw_dataset"]:
        print("iw", c["seed"], c["batch_size"], [y["id"] for y in c["yielded"]])
