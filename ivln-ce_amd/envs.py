"""Synthetic vector environment speaking both step protocols of the reference's `ExtendedVectorEnv`
(ivlnce_baselines/common/env_utils.py:117-254; SURVEY.md Appendix D: num_envs, reset(), step(list[int]), reset_at,
pause_at, current_episodes(), number_of_episodes, observation_spaces, action_spaces, close()):

  episodic   (`VLNCEDaggerEnv`)     reset() -> [obs]; step(a) -> [(obs, reward, done, info)]; reset_at(i) -> [obs]
  iterative  (`VLNCEIterativeEnv`)  reset() -> [(obs, tour_done, produce_action)];
                                    step(a) -> [(obs, reward, agent_episode_done, sim_episode_done, tour_done,
                                                 produce_action, info)];  reset_at(i) -> [(obs, tour_done, produce_action)]

and, like the reference's workers, resets a finished env inside `step` only when `auto_reset_done` (the collection
loops; evaluation resets through `reset_at`).  The iterative env walks the reference's phases
(ivlnce_baselines/common/environments.py:36-356): after the agent's episode an oracle conveys the agent to the goal
("oracle_goal"), and - when the next episode belongs to the same tour - from there to the next start pose
("oracle_start"); `agent_episode_done` is True in every oracle step, `produce_action` False while the oracle drives,
`tour_done` is only ever reported by a reset, `info["dtw_data"]` carries the positions logged since the last reset
whenever an agent or sim episode ends.  ENVIRONMENT.ITERATIVE.{ORACLE_PHASES, ORACLE_GOAL_PHASE,
PRECISE_EPISODE_START, ORACLE_STEP_ERROR_LIMIT} are honoured.

Habitat-Sim, MP3D and R2R-CE do not exist on the GPU box (simulator side is out of scope, SURVEY.md section 2 rows
8-9): this stand-in feeds the hot path observations of exactly the sensor dtypes/shapes of section 8a row A0, a
scripted expert (`shortest_path_sensor`) and a kinematic agent in obstacle-free space (0.25 m forward / 15 degree
turns; the oracle is a turn-then-advance follower), so positions, tours and t-nDTW are meaningful.
"""
import math
from types import SimpleNamespace
from typing import List

import numpy as np

from .measures import SDTW, ndtw
from .spaces import Box, Dict, Discrete

STOP, FORWARD, LEFT, RIGHT = 0, 1, 2, 3
STEP_M, TURN_RAD = 0.25, math.radians(15.0)


def _advance(pose, heading, a):
    """One kinematic action; returns the new heading (pose is updated in place)."""
    if a == FORWARD:
        pose[0] += np.float32(-STEP_M * math.sin(heading))
        pose[2] += np.float32(-STEP_M * math.cos(heading))
    elif a == LEFT:
        heading += TURN_RAD
    elif a == RIGHT:
        heading -= TURN_RAD
    return heading


def _wrap(a):
    return (a + math.pi) % (2.0 * math.pi) - math.pi


class _SynthEnv:
    def __init__(self, idx, seed, cfg, episodes_per_tour=3, n_episodes=8, min_len=6, max_len=14, with_rgb=False,
                 with_semantic=True, iterative=False):
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
        it = cfg.TASK_CONFIG.ENVIRONMENT.ITERATIVE
        self.iterative = iterative
        self.oracle_phases, self.oracle_goal_phase = bool(it.ORACLE_PHASES), bool(it.ORACLE_GOAL_PHASE)
        self.precise_start, self.oracle_limit = bool(it.PRECISE_EPISODE_START), int(it.ORACLE_STEP_ERROR_LIMIT)
        self.episodes = []
        starts = np.random.RandomState(seed + 7)  # own stream: scripts / tokens / frames keep their round-1 values
        for e in range(n_episodes):
            n = self.rng.randint(min_len, max_len + 1)
            script = list(self.rng.choice([FORWARD, FORWARD, LEFT, RIGHT], size=n - 1)) + [STOP]
            tokens = np.zeros(200, np.int64)
            L = self.rng.randint(10, 81)
            tokens[:L] = self.rng.randint(2, 2504, size=L)
            off = starts.uniform(-1.0, 1.0, size=2) if iterative else (0.0, 0.0)
            start = np.array([float(idx) * 3.0 + off[0], 1.25, off[1]], np.float32)
            ep = SimpleNamespace(episode_id=f"{idx}_{e}", tour_id=f"tour{idx}_{e // episodes_per_tour}",
                                 scene_id=f"scene{idx}", script=script, tokens=tokens, start=start, start_heading=0.0)
            ep.goal = np.array(self._gt_positions(ep)[-1], np.float32)
            self.episodes.append(ep)
        self.ep_i = -1          # nothing loaded: the first reset has no previous episode
        self.phase = "agent"
        self.dtw_data = []      # iterative protocol: positions since the last reset (environments.py:79-86)
        self._tour_log = []     # episodic protocol: every first-pass position (t-nDTW extension, see dtw_data())
        self.exhausted = False  # every episode played once: stop logging (eval pauses the env)
        self.pose = np.zeros(3, np.float32)
        self.heading = 0.0
        self.t = 0
        self._oracle_steps = 0

    @property
    def current_episode(self):
        return self.episodes[self.ep_i % len(self.episodes)]

    # -- observations / metrics ------------------------------------------------------------------
    def _obs(self):
        ep = self.current_episode
        col = self.rng.rand(1, self.W, 1).astype(np.float32)
        depth = np.clip(0.2 + 0.6 * col + 0.02 * self.rng.rand(self.H, self.W, 1).astype(np.float32), 0, 1)
        expert = ep.script[self.t] if (self.phase == "agent" and self.t < len(ep.script)) else STOP
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

    @staticmethod
    def _gt_positions(ep):
        pose, heading = ep.start.copy(), ep.start_heading
        pts = [[float(x) for x in pose]]
        for a in ep.script:
            heading = _advance(pose, heading, a)
            if a == FORWARD:
                pts.append([float(x) for x in pose])
        return pts

    def _metrics(self, action, done):
        success = float(done and int(action) == STOP)
        ndtw_v = 0.0
        if done:  # per-episode nDTW / SDTW against the scripted expert's path (measures.py:152-230)
            agent = [p for i, p in enumerate(self._positions) if i == 0 or p != self._positions[i - 1]]
            ndtw_v = ndtw(agent, self._gt_positions(self.current_episode), self.ndtw_success_distance, self.ndtw_fdtw)
        return {"distance_to_goal": float(np.linalg.norm((self.pose - self.current_episode.goal)[[0, 2]])),
                "success": success, "spl": 0.0, "ndtw": ndtw_v, "sdtw": SDTW.get_metric(success, ndtw_v),
                "path_length": STEP_M * self.t, "oracle_success": 0.0, "steps_taken": float(self.t)}

    def _load_next_episode(self):
        if self.ep_i + 1 >= len(self.episodes):
            self.exhausted = True
        self.ep_i += 1
        self.t = 0

    def _place_at_start(self):
        ep = self.current_episode
        self.pose, self.heading = ep.start.copy(), ep.start_heading
        self._positions = [[float(x) for x in self.pose]]

    def _agent_step(self, action):
        self.heading = _advance(self.pose, self.heading, int(action))
        self.t += 1
        self._positions.append([float(x) for x in self.pose])
        return int(action) == STOP or self.t >= min(self.max_steps, 4 * len(self.current_episode.script))

    # -- episodic protocol --------------------------------------------------------------------------
    def _log_episodic(self):
        if not self.exhausted:
            self._tour_log.append({"position": [float(x) for x in self.pose], "phase": "agent",
                                   "episode_id": self.current_episode.episode_id})

    def reset_episodic(self):
        self._load_next_episode()
        self.phase = "agent"
        self._place_at_start()
        self._log_episodic()
        return self._obs()

    def step_episodic(self, action, auto_reset):
        done = self._agent_step(action)
        self._log_episodic()
        info = self._metrics(action, done)
        obs = self.reset_episodic() if (done and auto_reset) else self._obs()
        return obs, 0.0, done, info

    # -- iterative protocol -------------------------------------------------------------------------
    def _oracle_action(self, position_to, heading_to):
        """Turn-then-advance follower with the stopping rules of `_get_next_action` (environments.py:195-229): STOP
        within one forward step of the target and, when a heading is asked for, within half a turn of it."""
        dx, dz = float(position_to[0] - self.pose[0]), float(position_to[2] - self.pose[2])
        if math.hypot(dx, dz) >= STEP_M:
            delta = _wrap(math.atan2(-dx, -dz) - self.heading)
            if abs(delta) >= TURN_RAD / 2:
                return LEFT if delta > 0 else RIGHT
            return FORWARD
        if heading_to is not None:
            delta = _wrap(heading_to - self.heading)
            if abs(delta) >= TURN_RAD / 2:
                return LEFT if delta > 0 else RIGHT
        return STOP

    def _oracle_action_safe(self, position_to, heading_to, teleport_on_failure):
        """`_get_next_action_safe` (:149-193): past ORACLE_STEP_ERROR_LIMIT the oracle gives up (teleporting to the
        target if asked to) and calls STOP."""
        if 0 <= self.oracle_limit <= self._oracle_steps:
            if teleport_on_failure:
                self.pose = np.array(position_to, np.float32)
                if heading_to is not None:
                    self.heading = heading_to
            return STOP
        return self._oracle_action(position_to, heading_to)

    def _next_phase(self):
        self.phase = {"agent": "oracle_goal", "oracle_goal": "oracle_start", "oracle_start": "agent"}[self.phase]
        self._oracle_steps = 0

    def reset(self):
        """(:91-147) -> (observations, tour_done, produce_action)."""
        self.dtw_data = []
        self.phase = "agent"
        self._oracle_steps = 0
        first = self.ep_i < 0
        prev_tour = None if first else self.current_episode.tour_id
        prev_pose, prev_heading = self.pose.copy(), self.heading
        self._load_next_episode()
        self._place_at_start()
        observations = self._obs()
        if first:
            return observations, True, True
        tour_done = prev_tour != self.current_episode.tour_id
        produce_action = True
        if tour_done or not self.oracle_phases:
            return observations, tour_done, produce_action
        # same tour: back to where the previous episode ended; the oracle walks to the new start from there
        self.phase = "oracle_start"
        self.pose, self.heading = prev_pose, prev_heading
        ep = self.current_episode
        if self._oracle_action_safe(ep.start, ep.start_heading, True) == STOP:
            self._next_phase()
            self._positions = [[float(x) for x in self.pose]]
        else:
            produce_action = False
        return observations, tour_done, produce_action

    def _step_oracle(self):
        ep = self.current_episode
        to, heading_to = (ep.goal, None) if self.phase == "oracle_goal" else (ep.start, ep.start_heading)
        self.heading = _advance(self.pose, self.heading, self._oracle_action(to, heading_to))
        nxt = self._oracle_action_safe(to, heading_to, self.phase == "oracle_start")
        if nxt == STOP:
            if self.phase == "oracle_start":
                if self.precise_start:
                    self.pose, self.heading = ep.start.copy(), ep.start_heading
                self._positions = [[float(x) for x in self.pose]]
            self._next_phase()
        else:
            self._oracle_steps += 1
        return self._obs()

    def step(self, action, auto_reset):
        """(:287-356) -> (observations, reward, agent_episode_done, sim_episode_done, tour_done, produce_action, info)."""
        agent_done, sim_done, tour_done, produce_action, info = True, False, False, False, {}
        if not self.exhausted:
            self.dtw_data.append({"position": [float(x) for x in self.pose], "phase": self.phase,
                                  "episode_id": self.current_episode.episode_id})
        if self.phase == "agent":
            agent_done = self._agent_step(action)
            produce_action = True
            info = self._metrics(action, agent_done)
            if agent_done:
                self._next_phase()
                produce_action = False
                if not self.oracle_phases:
                    self.phase = "agent"
                    sim_done = True
                elif (self._oracle_action_safe(self.current_episode.goal, None, False) == STOP
                      or not self.oracle_goal_phase):
                    self._next_phase()
                    sim_done = True
            observations = self._obs()
        elif self.phase == "oracle_goal":
            observations = self._step_oracle()
            sim_done = self.phase == "oracle_start"
        else:
            observations = self._step_oracle()
            produce_action = self.phase == "agent"
        if agent_done or sim_done:
            info["dtw_data"] = [dict(p) for p in self.dtw_data]  # the worker pipe pickles it: a snapshot
        if auto_reset and sim_done:
            observations, tour_done, produce_action = self.reset()
        return observations, 0.0, agent_done, sim_done, tour_done, produce_action, info

    def expert_path(self):
        """Positions of the scripted expert for every episode, tour-grouped (ground truth for t-nDTW: what
        `EVAL.ITERATIVE_GT_PATHS[split]` holds for the real dataset)."""
        out = {}
        for ep in self.episodes:
            pose, heading = ep.start.copy(), ep.start_heading
            pts = [{"position": [float(x) for x in pose], "phase": "agent", "episode_id": ep.episode_id}]
            for a in ep.script:
                heading = _advance(pose, heading, a)
                pts.append({"position": [float(x) for x in pose], "phase": "agent", "episode_id": ep.episode_id})
            out.setdefault(ep.tour_id, []).extend(pts)
        return out


class SyntheticVectorEnv:
    def __init__(self, config, num_envs=None, seed=None, rank=0, world=1, iterative=None, auto_reset_done=True,
                 **env_kw):
        n = num_envs if num_envs is not None else config.NUM_ENVIRONMENTS
        seed = config.TASK_CONFIG.SEED if seed is None else seed
        self.iterative = ("Iterative" in str(config.ENV_NAME)) if iterative is None else bool(iterative)
        self.auto_reset_done = bool(auto_reset_done)
        # envs sharded round-robin over ranks like construct_envs splits scenes (env_utils.py:77-99)
        ids = [i for i in range(n * world) if i % world == rank]
        # RGB frames only for policies that read them (RedNet-predicted semantics, Latent-CMA): they are 150 KB
        # of host RNG per step and env otherwise
        needs_rgb = (config.MODEL.policy_name == "LatentCMAPolicy"
                     or any("Predicted" in t for t in config.RL.POLICY.OBS_TRANSFORMS.ENABLED_TRANSFORMS))
        env_kw.setdefault("with_rgb", needs_rgb)
        self._envs: List[_SynthEnv] = [_SynthEnv(i, seed + 1000 * i, config, iterative=self.iterative, **env_kw)
                                       for i in ids]
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
        return [e.reset() if self.iterative else e.reset_episodic() for e in self._envs]

    def step(self, actions):
        if self.iterative:
            return [e.step(a, self.auto_reset_done) for e, a in zip(self._envs, actions)]
        return [e.step_episodic(a, self.auto_reset_done) for e, a in zip(self._envs, actions)]

    def reset_at(self, i):
        return [self._envs[i].reset() if self.iterative else self._envs[i].reset_episodic()]

    def pause_at(self, i):
        self._paused.append(self._envs.pop(i))
        self.observation_spaces.pop(i)
        self.action_spaces.pop(i)

    def current_episodes(self):
        return [e.current_episode for e in self._envs]

    def dtw_data(self):
        """Episodic protocol only: the first-pass positions of every env, tour-grouped.  (The reference's episodic
        evaluation reports no t-nDTW; the iterative one collects `infos["dtw_data"]`.)"""
        out = {}
        for e in self._envs + self._paused:
            for p in e._tour_log:
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
    """The vector env the trainers of this package drive (the reference: env_utils.py:25-114; `auto_reset_done`
    False = `construct_envs_auto_reset_false`).  The step protocol follows `config.ENV_NAME` ("Iterative" in the
    name -> the 7-tuple protocol) unless `iterative=` says otherwise.

    Only the synthetic backend exists in this scope (simulator side: SURVEY.md section 2 rows 8-9).  The choice is
    explicit: `config.ENV_BACKEND` ("synthetic", the default of this package's config) or the IVLN_ENV_BACKEND
    environment variable; anything else fails loudly instead of silently training on synthetic observations.  The
    trainers themselves speak the reference's env protocol, so a Habitat `ExtendedVectorEnv` can be handed to
    them where Habitat exists."""
    import os

    backend = os.environ.get("IVLN_ENV_BACKEND") or str(getattr(config, "ENV_BACKEND", "synthetic"))
    if backend != "synthetic":
        raise NotImplementedError(
            f"ENV_BACKEND={backend!r}: ivln_ce_amd.envs only builds the synthetic vector env; real Habitat envs "
            "come from the reference's construct_envs (same step protocol)")
    return SyntheticVectorEnv(config, rank=rank, world=world, auto_reset_done=auto_reset_done, **kw)
