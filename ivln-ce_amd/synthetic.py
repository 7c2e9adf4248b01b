"""Seeded synthetic observations of the shapes/dtypes the Habitat sensors produce (SURVEY.md
section 8a row A0 / 8d): used by bench.py, smoke() and the tests, because neither Habitat-Sim nor
MP3D data exist on the GPU box.

depth  f32 (B,256,256,1) = clamp(0.2 + 0.6*U_col + 0.02*U, 0, 1) (wall-like, 100% valid)
rgb    u8  (B,224,224,3); semantic12 u8 (B,256,256,1) in [0,12]
instruction i64 (B,200): first `n_tokens` in [2, vocab), rest 0 (PAD)
pose starts (0,1.25,0); each step moves 0.25 m along the heading then turns 15 deg (fp64)
"""
import math
from typing import Dict

import torch


class SyntheticRollout:
    def __init__(self, B=4, H=256, W=256, rgb_hw=224, n_tokens=80, vocab=2504, seed=1234, with_rgb=False,
                 reset_every=0):
        self.B, self.H, self.W, self.rgb_hw = B, H, W, rgb_hw
        self.with_rgb = with_rgb
        self.g = torch.Generator().manual_seed(seed)
        self.t = 0
        self.reset_every = reset_every
        self.pose = torch.zeros(B, 3)
        self.pose[:, 1] = 1.25
        self.heading = torch.zeros(B, dtype=torch.float64)
        self.instruction = torch.zeros(B, 200, dtype=torch.int64)
        self.instruction[:, :n_tokens] = torch.randint(2, vocab, (B, n_tokens), generator=self.g)

    def step(self) -> Dict[str, torch.Tensor]:
        B, H, W, g = self.B, self.H, self.W, self.g
        col = torch.rand(B, 1, W, 1, generator=g)
        depth = (0.2 + 0.6 * col + 0.02 * torch.rand(B, H, W, 1, generator=g)).clamp(0, 1)
        obs = {
            "depth": depth,
            "semantic12": torch.randint(0, 13, (B, H, W, 1), generator=g, dtype=torch.uint8),
            "instruction": self.instruction.clone(),
            "world_robot_pose": self.pose.clone(),
            "world_robot_orientation": torch.stack(
                [torch.zeros(B, dtype=torch.float64), self.heading.clone()], 1
            ),
            "not_done_masks": torch.ones(B, 1, dtype=torch.uint8),
            "env_name": ["synthetic"] * B,
            "progress": torch.rand(B, 1, generator=g, dtype=torch.float64),
        }
        if self.t == 0 or (self.reset_every and self.t % self.reset_every == 0):
            obs["not_done_masks"][:] = 0
        if self.with_rgb:
            obs["rgb"] = torch.randint(0, 256, (B, self.rgb_hw, self.rgb_hw, 3), generator=g, dtype=torch.uint8)
        # advance: forward 0.25 m along heading (habitat: -z is forward), then heading += 15 deg
        self.pose[:, 0] += (-0.25 * torch.sin(self.heading)).float()
        self.pose[:, 2] += (-0.25 * torch.cos(self.heading)).float()
        self.heading += math.radians(15.0)
        self.t += 1
        return obs
