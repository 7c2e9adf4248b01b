"""Two ranks on one GPU (gloo): DAgger train (env sharding, flat-bucket all-reduce, rank-0 checkpoint) then
eval (stats + dtw_data gathered on rank 0).  Control-flow smoke test of BASELINE configs[3]/[4]:
  IVLN_DIST_BACKEND=gloo IVLN_ONE_DEVICE=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
      --master-addr 127.0.0.1 --master-port 29531 tools/dist_smoke.py <scratch dir>"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import ivln_ce_amd  # noqa: E402,F401
from ivln_ce_amd import dist as D  # noqa: E402
from ivln_ce_amd import trainers  # noqa: E402,F401
from ivln_ce_amd.config import get_config  # noqa: E402
from ivln_ce_amd.registry import baseline_registry  # noqa: E402

out = sys.argv[1]
TRAINER = sys.argv[2] if len(sys.argv) > 2 else "dagger"  # or iterative_dagger: tour batches, MIN-reduced batch count
EVAL_ONLY = len(sys.argv) > 3 and sys.argv[3] == "eval_only"  # eval() must set the process group up by itself
if EVAL_ONLY:
    rank, _, world = D.world_info()
else:
    rank, _, world = D.init()
torch.manual_seed(0)
np.random.seed(rank)
cfg = get_config(opts=[
    "TRAINER_NAME", TRAINER, "NUM_ENVIRONMENTS", 2, "MODEL.policy_name", "MapCMAPolicy",
    "MODEL.INSTRUCTION_ENCODER.use_pretrained_embeddings", False, "MODEL.DEPTH_ENCODER.ddppo_checkpoint", "NONE",
    "RL.POLICY.OBS_TRANSFORMS.ENABLED_TRANSFORMS", ["GTSemanticsIterativeMapper"],
    "IL.DAGGER.iterations", 1, "IL.DAGGER.update_size", 4 if TRAINER == "dagger" else 16, "IL.DAGGER.p", 0.5,
    "IL.epochs", 1, "IL.batch_size", 2,
    "IL.DAGGER.lmdb_features_dir", os.path.join(out, f"traj{rank}"), "CHECKPOINT_FOLDER", os.path.join(out, "ckpt"),
    "RESULTS_DIR", os.path.join(out, "res"), "EVAL_CKPT_PATH_DIR", os.path.join(out, "ckpt"),
])
if EVAL_ONLY:
    os.makedirs(os.path.join(out, "ckpt"), exist_ok=True)
    ckpt = os.path.join(out, "ckpt", "none.pth")  # absent file: random-init weights, broadcast from rank 0
    cfg.defrost()
    cfg.EVAL_CKPT_PATH_DIR = ckpt
    cfg.freeze()
    res = [baseline_registry.get_trainer(TRAINER)(cfg)._eval_checkpoint(ckpt, None, 0)]
    assert torch.distributed.is_initialized(), "eval must initialise the process group itself"
    if rank == 0:
        # both ranks' episodes were gathered: 2 envs per rank x 8 episodes
        assert res[0]["episodes"] == 2 * world * 8, res[0]["episodes"]
        print("dist smoke ok: eval_only world", world, "episodes", res[0]["episodes"])
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()
    sys.exit(0)
tr = baseline_registry.get_trainer(TRAINER)(cfg)
log = tr.train()
assert len(log) >= 1 and all(np.isfinite(l["loss"]) for l in log)
# data-parallel replicas must hold identical parameters after the all-reduced update
flat = tr.optimizer.flat.detach().clone()
gathered = D.gather_objects(float(flat.double().sum().item()))
assert max(gathered) - min(gathered) == 0.0, gathered
res = baseline_registry.get_trainer(TRAINER)(cfg).eval()
if rank == 0:
    print("dist smoke ok:", TRAINER, "world", world, "updates", len(log), "eval", {k: round(v, 4) for k, v in res[0].items() if k in ("episodes", "t_ndtw")})
torch.distributed.barrier()
torch.distributed.destroy_process_group()
