"""Synthetic vector environment with the `VectorEnv` surface the reference's rollout / eval loops
use (num_envs, reset(), step(list[int]), reset_at, pause_at,
current_episodes(), number_of_episodes, observation_spaces, action_spaces, close()).

Habitat-Sim, MP3D and R2R-CE do not exist on the GPU box (simulator side is out of scope,
SURVEY.md section 2 rows 8-9): when `habitat` is importable the trainers call the reference's
`construct_envs` unchanged, otherwise this stand-in feeds the hot path observations of exactly the
sensor dtypes/shapes of section 8a row A0, a scripted expert (`shortest_path_sensor`) and a
kinematic agent (0.25 m forward / 15 degree turns) so positions, tours and t-nDTW are meaningful.
"""
import math
from types import SimpleNamespace
from typing import List

import numpy as np

from .measures import SDTW, ndtw
from .spaces import Box, Dict, Discrete

STOP, FORWARD, LEFT, RIGHT = 0, 1, 2, 3


class _SynthEnv:
    def __init__(self, idx, seed, cfg, episodes_per_tour=3, n_episodes=8, min_len=6, max_len=14, with_rgb=False,
                 with_semantic=True):
        self.idx = idx
        self.rng = np.random.RandomState(seed)
        d = cfg.TASK_CONFIG.SIMULATOR.DEPTH_SENSOR
        self.H, self.W = d.HEIGHT, d.WIDTH
        r = cfg.TASK_CONFIG.SIMULATOR.RGB_SENSOR
        self.rgb_hw = (r.HEIGHT, r.WIDTH)
        self.with_rgb, self.with_semantic = with_rgb, with_semantic
        self.max_steps = cfg.TASK_CONFIG.ENVIRONMENT.MAX_EPISODE_STEPS
        nd = cfg.TASK_CONFIG.TASK.NDTW
        self.ndtw_fdtw, self.ndtw_success_distance = bool(nd.FDTW), float(nd.SUCCESS_DISTANCE)
        self.episodes = []
        for e in range(n_episodes):
            n = self.rng.randint(min_len, max_len + 1)
            script = list(self.rng.choice([FORWARD, FORWARD, LEFT, RIGHT], size=n - 1)) + [STOP]
            tokens = np.zeros(200, np.int64)
            L = self.rng.randint(10, 81)
            tokens[:L] = self.rng.randint(2, 2504, size=L)
            self.episodes.append(SimpleNamespace(
                episode_id=f"{idx}_{e}", tour_id=f"tour{idx}_{e // episodes_per_tour}", scene_id=f"scene{idx}",
                script=script, tokens=tokens, start=np.array([float(idx) * 3.0, 1.25, 0.0], np.float32),
            ))
        self.ep_i = -1
        self.dtw_data = []
        self.exhausted = False  # every episode played once: stop logging dtw data (eval pauses the env)
        self.reset()

    # -- kinematics --------------------------------------------------------------------------
    def _apply(self, a):
        if a == FORWARD:
            self.pose[0] += np.float32(-0.25 * math.sin(self.heading))
            self.pose[2] += np.float32(-0.25 * math.cos(self.heading))
        elif a == LEFT:
            self.heading += math.radians(15.0)
        elif a == RIGHT:
            self.heading -= math.radians(15.0)

    def _obs(self):
        ep = self.current_episode
        col = self.rng.rand(1, self.W, 1).astype(np.float32)
        depth = np.clip(0.2 + 0.6 * col + 0.02 * self.rng.rand(self.H, self.W, 1).astype(np.float32), 0, 1)
        expert = ep.script[self.t] if self.t < len(ep.script) else STOP
        obs = {
            "depth": depth.astype(np.float32),
            "instruction": ep.tokens.copy(),
            "world_robot_pose": self.pose.copy(),
            "world_robot_orientation": np.array([0.0, self.heading], np.float64),
            "env_name": ep.scene_id,
            "progress": np.array([min(1.0, self.t / max(1, len(ep.script)))], np.float64),
            "shortest_path_sensor": np.array([float(expert)], np.float64),
        }
        if self.with_semantic:
            obs["semantic12"] = self.rng.randint(0, 13, size=(self.H, self.W, 1)).astype(np.uint8)
        if self.with_rgb:
            obs["rgb"] = self.rng.randint(0, 256, size=(*self.rgb_hw, 3)).astype(np.uint8)
        return obs

    def _log(self):
        if self.exhausted:
            return
        self.dtw_data.append({"position": [float(x) for x in self.pose], "phase": "agent",
                              "episode_id": self.current_episode.episode_id})

    def reset(self):
        if self.ep_i + 1 >= len(self.episodes):
            self.exhausted = True
        self.ep_i = (self.ep_i + 1) % len(self.episodes)
        self.current_episode = self.episodes[self.ep_i]
        self.pose = self.current_episode.start.copy()
        self.heading = 0.0
        self.t = 0
        self._positions = [[float(x) for x in self.pose]]
        self._log()
        return self._obs()

    @staticmethod
    def _gt_positions(ep):
        pose, heading = ep.start.copy(), 0.0
        pts = [[float(x) for x in pose]]
        for a in ep.script:
            if a == FORWARD:
                pose[0] += np.float32(-0.25 * math.sin(heading))
                pose[2] += np.float32(-0.25 * math.cos(heading))
                pts.append([float(x) for x in pose])
            elif a == LEFT:
                heading += math.radians(15.0)
            elif a == RIGHT:
                heading -= math.radians(15.0)
        return pts

    def step(self, action):
        self._apply(int(action))
        self.t += 1
        done = int(action) == STOP or self.t >= min(self.max_steps, 4 * len(self.current_episode.script))
        self._log()
        success = float(done and int(action) == STOP)
        self._positions.append([float(x) for x in self.pose])
        ndtw_v = 0.0
        if done:  # per-episode nDTW / SDTW against the scripted expert's path (measures.py:152-230)
            agent = [p for i, p in enumerate(self._positions) if i == 0 or p != self._positions[i - 1]]
            ndtw_v = ndtw(agent, self._gt_positions(self.current_episode), self.ndtw_success_distance, self.ndtw_fdtw)
        info = {"distance_to_goal": 0.0, "success": success, "spl": 0.0, "ndtw": ndtw_v,
                "sdtw": SDTW.get_metric(success, ndtw_v),
                "path_length": 0.25 * self.t, "oracle_success": 0.0, "steps_taken": float(self.t)}
        obs = self.reset() if done else self._obs()
        return obs, 0.0, done, info

    def expert_path(self):
        """Positions of the scripted expert for every episode, tour-grouped (gt for t-nDTW)."""
        out = {}
        for ep in self.episodes:
            pose, heading = ep.start.copy(), 0.0
            pts = [{"position": [float(x) for x in pose], "phase": "agent", "episode_id": ep.episode_id}]
            for a in ep.script:
                if a == FORWARD:
                    pose[0] += np.float32(-0.25 * math.sin(heading))
                    pose[2] += np.float32(-0.25 * math.cos(heading))
                elif a == LEFT:
                    heading += math.radians(15.0)
                elif a == RIGHT:
                    heading -= math.radians(15.0)
                pts.append({"position": [float(x) for x in pose], "phase": "agent", "episode_id": ep.episode_id})
            out.setdefault(ep.tour_id, []).extend(pts)
        return out


class SyntheticVectorEnv:
    def __init__(self, config, num_envs=None, seed=None, rank=0, world=1, **env_kw):
        n = num_envs if num_envs is not None else config.NUM_ENVIRONMENTS
        seed = config.TASK_CONFIG.SEED if seed is None else seed
        # envs sharded round-robin over ranks like construct_envs splits scenes (env_utils.py:77-99)
        ids = [i for i in range(n * world) if i % world == rank]
        # RGB frames only for policies that read them (RedNet-predicted semantics, Latent-CMA): they are 150 KB
        # of host RNG per step and env otherwise
        needs_rgb = (config.MODEL.policy_name == "LatentCMAPolicy"
                     or any("Predicted" in t for t in config.RL.POLICY.OBS_TRANSFORMS.ENABLED_TRANSFORMS))
        env_kw.setdefault("with_rgb", needs_rgb)
        self._envs: List[_SynthEnv] = [_SynthEnv(i, seed + 1000 * i, config, **env_kw) for i in ids]
        self._paused: List[_SynthEnv] = []
        d = config.TASK_CONFIG.SIMULATOR.DEPTH_SENSOR
        sp = {
            "depth": Box(0.0, 1.0, (d.HEIGHT, d.WIDTH, 1), np.float32),
            "instruction": Box(0, 2504, (200,), np.int64),
            "world_robot_pose": Box(-1e6, 1e6, (3,), np.float32),
            "world_robot_orientation": Box(-1e6, 1e6, (2,), np.float64),
            "semantic12": Box(0, 12, (d.HEIGHT, d.WIDTH, 1), np.uint8),
            "progress": Box(0.0, 1.0, (1,), np.float64),
            "shortest_path_sensor": Box(0.0, 100.0, (1,), np.float64),
        }
        if env_kw["with_rgb"]:
            r = config.TASK_CONFIG.SIMULATOR.RGB_SENSOR
            sp["rgb"] = Box(0, 255, (r.HEIGHT, r.WIDTH, 3), np.uint8)
        self.observation_spaces = [Dict(dict(sp)) for _ in self._envs]
        self.action_spaces = [Discrete(4) for _ in self._envs]

    @property
    def num_envs(self):
        return len(self._envs)

    @property
    def number_of_episodes(self):
        return [len(e.episodes) for e in self._envs]

    def reset(self):
        return [e._obs() for e in self._envs]

    def step(self, actions):
        return [e.step(a) for e, a in zip(self._envs, actions)]

    def reset_at(self, i):
        return [self._envs[i].reset()]

    def pause_at(self, i):
        self._paused.append(self._envs.pop(i))
        self.observation_spaces.pop(i)
        self.action_spaces.pop(i)

    def current_episodes(self):
        return [e.current_episode for e in self._envs]

    def dtw_data(self):
        out = {}
        for e in self._envs + self._paused:
            for p in e.dtw_data:
                tour = next(ep.tour_id for ep in e.episodes if ep.episode_id == p["episode_id"])
                out.setdefault(tour, []).append(p)
        return out

    def gt_paths(self):
        out = {}
        for e in self._envs + self._paused:
            out.update(e.expert_path())
        return out

    def close(self):
        self._envs, self._paused = [], []


def construct_envs(config, env_class=None, auto_reset_done=True, rank=0, world=1, **kw):
    """The vector env the trainers of this package drive.

    Only the synthetic env is supported: the rollout / eval loops in trainers.py read trajectories and ground
    truth through `dtw_data()` / `gt_paths()` and rely on its auto-advancing episodes, while a real Habitat
    `VectorEnv` (env_utils.py:25-108) reports them through `infos["dtw_data"]` and needs `reset_at` after every
    done (base_il_trainer.py:495, 805) - that simulator-side protocol is out of scope (SURVEY.md section 2 rows
    8-9).  The choice is explicit: `config.ENV_BACKEND` ("synthetic", the default of this package's config) or the
    IVLN_ENV_BACKEND environment variable; anything else fails loudly instead of silently training on synthetic
    observations.  To run on Habitat, use the reference's own trainers with this package's policy / mapper
    plugins."""
    import os

    backend = os.environ.get("IVLN_ENV_BACKEND") or str(getattr(config, "ENV_BACKEND", "synthetic"))
    if backend != "synthetic":
        raise NotImplementedError(
            f"ENV_BACKEND={backend!r}: ivln_ce_amd.trainers only drives the synthetic vector env; real Habitat envs "
            "run through the reference's trainers with the MapCMAPolicy / *Mapper plugins of this package")
    return SyntheticVectorEnv(config, rank=rank, world=world, **kw)
