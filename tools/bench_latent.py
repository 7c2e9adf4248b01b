"""Latent-CMA rollout step throughput: python tools/bench_latent.py [envs]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import ivln_ce_amd  # noqa: E402,F401
from ivln_ce_amd import latent_policy  # noqa: E402,F401
from ivln_ce_amd.config import get_config  # noqa: E402
from ivln_ce_amd.graphed import GraphedRollout  # noqa: E402
from ivln_ce_amd.registry import baseline_registry  # noqa: E402
from ivln_ce_amd.spaces import Box, Dict, Discrete  # noqa: E402
from ivln_ce_amd.synthetic import SyntheticRollout  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device("cuda:0")
cfg = get_config(opts=["MODEL.policy_name", "LatentCMAPolicy", "MODEL.INSTRUCTION_ENCODER.use_pretrained_embeddings", False,
                       "MODEL.DEPTH_ENCODER.ddppo_checkpoint", "NONE"])
space = Dict({"depth": Box(0.0, 1.0, (256, 256, 1), np.float32), "rgb": Box(0, 255, (224, 224, 3), np.uint8),
              "instruction": Box(0, 2504, (200,), np.int64)})
torch.manual_seed(0)
pol = baseline_registry.get_policy("LatentCMAPolicy").from_config(cfg, space, Discrete(4)).to(dev).eval()
roll = SyntheticRollout(B=B, seed=5, with_rgb=True)
obs = [{k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in roll.step().items()} for _ in range(16)]
for mode in ("eager", "graph"):
    if mode == "graph":
        runner = GraphedRollout(pol, [], obs[0], deterministic=True, streams=False)
        step = lambda i: runner.step(obs[i % 16])  # noqa: E731
    else:
        state = {"rnn": torch.zeros(B, 2, 512, device=dev), "prev": torch.zeros(B, 1, dtype=torch.long, device=dev)}

        def step(i):
            with torch.no_grad():
                a, state["rnn"] = pol.act(obs[i % 16], state["rnn"], state["prev"], obs[i % 16]["not_done_masks"],
                                          deterministic=True)
            state["prev"] = a
    for i in range(10):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 100
    for i in range(n):
        step(i)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    print(f"Latent-CMA B={B} {mode}: {1e3 * el / n:.3f} ms/step  {B * n / el:.0f} env-steps/s "
          f"(RGB ResNet-50 8.2 GFLOP + depth ResNet 0.7 GFLOP per env-step)")
