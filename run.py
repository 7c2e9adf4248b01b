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


def run_exp(exp_config: str, run_type: str, opts=None) -> None:
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
