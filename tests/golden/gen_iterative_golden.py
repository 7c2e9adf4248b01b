"""Golden runs of the REFERENCE's own tour-by-tour collection and evaluation loops (build container only;
/root/reference never travels):

  iterative_golden.json
    collect/<case>   `IterativeCollectionDaggerTrainer._update_dataset(data_it, save_tour_idx_data=True)`
                     (ivlnce_baselines/trainers/iterative_collection_dagger_trainer.py:131-397) with its helpers
                     `masks_to_tensors` :82-114, `add_map_to_observations` :28-58, `batch_and_transform` :116-129,
                     `save_episode_to_disk` :60-80 and `_pause_iterative_envs` (common/base_il_trainer.py:258-311):
                     the stored records (numbered from 1), the {tour: [record ids]} table under key "0", the actions
                     the envs received, what every `policy.act_iterative` call saw (four masks, previous actions,
                     compacted state rows, the map), `delete_batch_idx` calls.  One case collects twice into the
                     same database (the second call continues the numbering and extends the table).
    eval/<case>      `BaseVLNCETrainer._eval_checkpoint` (:313-583, episodic) and `_eval_checkpoint_iterative`
                     (:585-928): env action / `reset_at` logs, per-call policy logs, and the report files
                     (`stats_ckpt_…`, `iterative_stats_ckpt_…`, `iterative_all_stats_ckpt_…`, `dtw_data_ckpt_…`).
                     `compute_tour_ndtw` (dtw-python, absent from the image) is replaced by a recorder: its
                     ARGUMENTS are part of the golden, its value is a sentinel.

The scripted env / policy / mapper stand-ins live in iterative_script.py and are shared with the tests.  lmdb,
msgpack_numpy, tensorflow and habitat are replaced by the arithmetic-free stand-ins of gen_rollout_golden.py."""
import importlib.util
import json
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_rollout_golden as G  # noqa: E402  (installs the stand-ins, imports the reference's dagger_trainer)
import iterative_script as IS  # noqa: E402

REF = G.REF
ser = G.ser

ot = sys.modules["habitat_baselines.common.obs_transformers"]


def _apply(batch, transforms):
    for t in transforms:
        batch = t(batch)
    return batch


ot.apply_obs_transforms_batch = _apply
# the names were bound at import time inside the already-imported reference modules
bt = sys.modules["ivlnce_baselines.common.base_il_trainer"]
bt.apply_obs_transforms_batch = _apply
NDTW_CALLS = []


def _record_tour_ndtw(agent_paths, gt_paths, success_distance):
    NDTW_CALLS.append({"agent_paths": json.loads(json.dumps(agent_paths)), "gt_paths": json.loads(json.dumps(gt_paths)),
                       "success_distance": success_distance})
    return 0.4242


bt.compute_tour_ndtw = _record_tour_ndtw
bt.is_slurm_batch_job = lambda: True  # use_pbar False: progress goes to the logger, not tqdm

spec = importlib.util.spec_from_file_location(
    "ivlnce_baselines.trainers.iterative_collection_dagger_trainer",
    os.path.join(REF, "ivlnce_baselines", "trainers", "iterative_collection_dagger_trainer.py"))
ict = importlib.util.module_from_spec(spec)
sys.modules[spec.name] = ict
spec.loader.exec_module(ict)
ict.apply_obs_transforms_batch = _apply


def _config(opts):
    return sys.modules["ivln_cfg_for_shim"].get_config(opts=opts)


def _records(db):
    out = {}
    for key, val in db.items():
        k = key.decode()
        if k == "0":
            continue
        obs, prev, oracle = val
        out[k] = {"obs": {n: ser(np.asarray(v)) for n, v in sorted(obs.items())}, "prev_actions": ser(prev),
                  "oracle_actions": ser(oracle)}
    return out


def collect_case(name):
    p, data_it, update_size, seed, oracle = IS.COLLECT_CASES[name]
    path = f"mem://iter_{name}"
    cfg = _config(["IL.DAGGER.p", p, "IL.DAGGER.update_size", update_size, "IL.DAGGER.lmdb_features_dir", path,
                   "IL.DAGGER.lmdb_fp16", False])
    tr = ict.IterativeCollectionDaggerTrainer.__new__(ict.IterativeCollectionDaggerTrainer)
    tr.config, tr.device = cfg, torch.device("cpu")
    tr.lmdb_features_dir = path
    tr.policy = IS.ScriptedIterativePolicy()
    runs = []
    for rep in range(2 if name == "beta_half_oracle" else 1):
        tr.obs_transforms = [IS.ScriptedMapper()]
        envs = IS.ScriptedVectorEnv(IS.scripts(), iterative=True, auto_reset=True, oracle_phases=oracle)
        G._ENVS["next"] = envs
        del tr.policy.calls[:], tr.policy.deleted[:]
        torch.manual_seed(seed + rep)
        table = tr._update_dataset(data_it + rep, save_tour_idx_data=True)
        db = G._DBS[path]
        stored = json.loads(db[b"0"].decode())
        assert stored == json.loads(json.dumps(table))
        runs.append({"data_it": data_it + rep, "seed": seed + rep, "tour_table": stored, "records": _records(db),
                     "env_actions": envs.action_log, "policy_calls": list(tr.policy.calls),
                     "deleted_batch_idx": list(tr.policy.deleted)})
    return {"p": p, "update_size": update_size, "oracle_phases": oracle, "runs": runs}


class _Writer:
    def __init__(self):
        self.scalars = []

    def add_scalar(self, k, v, step):
        self.scalars.append([k, float(v), int(step)])


def eval_case(name):
    iterative, map_reset, oracle = IS.EVAL_CASES[name]
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        gt = {"val_seen": {"T0": [[0.0, 0.0, 0.0]], "T1": [[1.0, 0.0, 0.0]]}}
        gt_file = os.path.join(tmp, "gt.json")
        json.dump(gt, open(gt_file, "w"))
        cfg = _config(["RESULTS_DIR", tmp, "EVAL.SPLIT", "val_seen", "EVAL.ITERATIVE_MAP_RESET", map_reset,
                       "EVAL.ITERATIVE_GT_PATHS", gt_file, "EVAL.SAVE_RESULTS", True,
                       "TASK_CONFIG.ENVIRONMENT.ITERATIVE.ENABLED", iterative, "VIDEO_OPTION", []])
        tr = bt.BaseVLNCETrainer.__new__(bt.BaseVLNCETrainer)
        tr.config, tr.device = cfg, torch.device("cpu")
        tr.obs_transforms = [IS.ScriptedMapper()]
        tr.policy = IS.ScriptedIterativePolicy(with_rgb=False)
        tr._get_spaces = lambda config, envs=None: (None, None)
        tr._initialize_policy = lambda *a, **k: None
        envs = IS.ScriptedVectorEnv(IS.scripts(), iterative=iterative, auto_reset=False, oracle_phases=oracle)
        G._ENVS["next"] = envs
        writer = _Writer()
        del NDTW_CALLS[:]
        tr._eval_checkpoint("data/checkpoints/ckpt.3.pth", writer, checkpoint_index=0)
        for f in sorted(os.listdir(tmp)):
            if f != "gt.json":
                out[f] = json.load(open(os.path.join(tmp, f)))
    return {"iterative": iterative, "map_reset": map_reset, "oracle_phases": oracle, "files": out,
            "env_actions": envs.action_log, "reset_at": [list(x) for x in envs.reset_at_log],
            "policy_calls": tr.policy.calls, "deleted_batch_idx": tr.policy.deleted, "scalars": writer.scalars,
            "tour_ndtw_calls": list(NDTW_CALLS)}


if __name__ == "__main__":
    gold = {"collect": {n: collect_case(n) for n in IS.COLLECT_CASES}, "eval": {n: eval_case(n) for n in IS.EVAL_CASES}}
    json.dump(gold, open(os.path.join(HERE, "iterative_golden.json"), "w"))
    for n, c in gold["collect"].items():
        for r in c["runs"]:
            print("collect", n, "records", sorted(r["records"], key=int), "tours", r["tour_table"], "steps",
                  len(r["env_actions"]), "rows", [x["rows"] for x in r["policy_calls"]][-6:], "deleted", r["deleted_batch_idx"])
    for n, c in gold["eval"].items():
        print("eval", n, "files", sorted(c["files"]), "steps", len(c["env_actions"]), "reset_at", len(c["reset_at"]),
              "ndtw calls", len(c["tour_ndtw_calls"]))
