#!/usr/bin/env python3
"""Same CLI as the reference's run.py (run.py:17-81):
    python run.py --run-type {train,eval,inference} --exp-config <yaml[,yaml]> [KEY VALUE ...]
Dispatches through the registry to the MI355X-native trainers (ivln-ce_amd/trainers.py)."""
import argparse
import os
import random
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ivln_ce_amd  # noqa: E402,F401
from ivln_ce_amd import obs_transforms, policy, trainers  # noqa: E402,F401  (register plugins)
from ivln_ce_amd.config import get_config  # noqa: E402
from ivln_ce_amd.registry import baseline_registry  # noqa: E402


def main():
    parser = argparse.ArgumentParser()
    parser.add_argument("--exp-config", type=str, required=True, help="path to config yaml containing info about experiment")
    parser.add_argument("--run-type", choices=["train", "eval", "inference"], required=True,
                        help="run type of the experiment (train, eval, inference)")
    parser.add_argument("opts", default=None, nargs=argparse.REMAINDER, help="Modify config options from command line")
    args = parser.parse_args()
    run_exp(**vars(args))


def _cap_host_threads():
    """The host side of a rollout (batch_obs, env bookkeeping) is thousands of small CPU tensor ops per second; with
    torch's default of one OpenMP thread per core and one process per GPU they oversubscribe the node (measured on a
    shared MI355X host: torch.stack of two depth frames 1.8 ms instead of 0.1 ms, the GPU test suite 400 s instead of
    90).  Default here: cores / ranks on the node, at most 8; OMP_NUM_THREADS or IVLN_HOST_THREADS decide otherwise."""
    if os.environ.get("OMP_NUM_THREADS"):
        return
    n = os.environ.get("IVLN_HOST_THREADS")
    if n is None:
        ranks = int(os.environ.get("LOCAL_WORLD_SIZE", "1") or 1)
        n = max(1, min(8, (os.cpu_count() or 1) // max(1, ranks)))
    torch.set_num_threads(int(n))


def run_exp(exp_config: str, run_type: str, opts=None) -> None:
    _cap_host_threads()
    config = get_config(exp_config, opts)
    random.seed(config.TASK_CONFIG.SEED)
    np.random.seed(config.TASK_CONFIG.SEED)
    torch.manual_seed(config.TASK_CONFIG.SEED)
    torch.backends.cudnn.benchmark = False
    torch.backends.cudnn.deterministic = False
    trainer_init = baseline_registry.get_trainer(config.TRAINER_NAME)
    assert trainer_init is not None, f"{config.TRAINER_NAME} is not supported"
    trainer = trainer_init(config)
    if run_type == "train":
        trainer.train()
    elif run_type == "eval":
        trainer.eval()
    elif run_type == "inference":
        trainer.inference()


if __name__ == "__main__":
    main()
