"""Scripted vector env speaking BOTH step protocols of the reference's `ExtendedVectorEnv`
(ivlnce_baselines/common/env_utils.py:117-254), plus arithmetic-free stand-ins for the policy and the mapper
obs-transform.  Shared by `gen_iterative_golden.py` (which drives the REFERENCE's own
`IterativeCollectionDaggerTrainer._update_dataset`, `BaseVLNCETrainer._eval_checkpoint` and
`_eval_checkpoint_iterative` with them, build container only) and `tests/test_host_logic.py` (which drives this
package's trainers with the same objects and compares with the golden).

  episodic protocol   reset() -> [obs]; step(a) -> [(obs, reward, done, info)]; reset_at(i) -> [obs]
  iterative protocol  reset() -> [(obs, tour_done, produce_action)];
                      step(a) -> [(obs, reward, agent_episode_done, sim_episode_done, tour_done, produce_action, info)];
                      reset_at(i) -> [(obs, tour_done, produce_action)]

The per-env state machine restates `VLNCEIterativeEnv` (ivlnce_baselines/common/environments.py:36-356) with the
simulator replaced by a script: phases agent -> oracle_goal -> oracle_start -> agent, `agent_episode_done` True in
every non-agent step, `produce_action` False while an oracle phase follows, `tour_done` only ever reported by
reset, `info["dtw_data"]` = the positions logged since the last reset whenever an agent or sim episode ends,
metrics only in agent-phase infos.  What is under test is the trainers' bookkeeping around these signals, not the
env: masks, which steps are stored, pausing, tour tables, statistics, report files."""
import copy
from types import SimpleNamespace

import numpy as np
import torch
import torch.nn as nn

EXPERT_UUID = "shortest_path_sensor"


def episode(eid, tour, expert, goal_steps=0, start_steps=0):
    """expert: the expert's action per agent step, ending with 0 (STOP) or -1 (goal unreachable); goal_steps: oracle
    steps from where the agent stopped to the goal; start_steps: oracle steps from the previous episode's end to
    this episode's start (spent when this episode follows another one of the same tour)."""
    return SimpleNamespace(episode_id=eid, tour_id=tour, expert=list(expert), goal_steps=goal_steps,
                           start_steps=start_steps, instruction=SimpleNamespace(instruction_text="unused"))


class _ScriptedEnv:
    def __init__(self, idx, episodes, iterative, oracle_phases, oracle_goal_phase=True):
        self.idx, self.episodes = idx, episodes
        self.iterative, self.oracle_phases, self.oracle_goal_phase = iterative, oracle_phases, oracle_goal_phase
        self.ep_i = -1           # nothing loaded yet: the first reset has no previous episode
        self.phase = ""
        self.t = 0               # agent steps taken in this episode
        self.k = 0               # oracle steps left in the current oracle phase
        self.n_steps = 0         # every step of this episode (positions)
        self.dtw_data = []

    @property
    def current_episode(self):
        return self.episodes[self.ep_i % len(self.episodes)]

    # -- observations ---------------------------------------------------------------------------
    def _obs(self):
        ep = self.current_episode
        k = self.ep_i % len(self.episodes)
        base = float(100 * self.idx + 10 * k + self.n_steps)
        expert = ep.expert[min(self.t, len(ep.expert) - 1)] if self.phase == "agent" else 0
        return {
            "instruction": {"tokens": np.array([self.idx + 1, k + 1, 7, 0, 0], dtype=np.int64), "text": "unused"},
            "depth": np.full((2, 2, 1), base / 1000.0, dtype=np.float32),
            "rgb": np.full((2, 2, 3), int(base) % 256, dtype=np.uint8),
            "semantic12": np.full((2, 2, 1), int(base) % 13, dtype=np.uint8),
            "world_robot_pose": np.array([base, 1.25, -base], dtype=np.float32),
            "world_robot_orientation": np.array([0.0, base / 100.0], dtype=np.float64),
            "env_name": f"scene{self.idx}",
            "progress": np.array([self.t / max(1, len(ep.expert) - 1)], dtype=np.float64),
            EXPERT_UUID: np.array([expert], dtype=np.float64),
            "feat": np.arange(3, dtype=np.float32) + base,
        }

    def _position(self):
        return [float(self.idx), float(self.ep_i % len(self.episodes)), float(self.n_steps)]

    def _metrics(self, done):
        s = float(self.t)
        return {"distance_to_goal": 0.5 * self.idx + 0.1 * s, "success": float(done and self.t % 2 == 0),
                "spl": 0.25 * float(done), "ndtw": 1.0 / (1.0 + s), "path_length": 0.25 * s,
                "oracle_success": float(self.t > 2), "steps_taken": s,
                "collisions": {"count": int(self.t), "is_collision": False}}  # not a number: filtered by the loops

    # -- episodic protocol ------------------------------------------------------------------------
    def reset_episodic(self):
        self.ep_i += 1
        self.phase, self.t, self.n_steps = "agent", 0, 0
        return self._obs()

    def step_episodic(self, action, auto_reset):
        ep = self.current_episode
        done = ep.expert[min(self.t, len(ep.expert) - 1)] in (0, -1)  # the script ends the episode
        self.t += 1
        self.n_steps += 1
        info = self._metrics(done)
        obs = self._obs()
        if done and auto_reset:
            obs = self.reset_episodic()
        return obs, 0.0, done, info

    # -- iterative protocol (environments.py:91-147, 287-356) ----------------------------------------------
    def reset(self):
        self.dtw_data = []
        first = self.ep_i < 0
        prev_tour = None if first else self.current_episode.tour_id
        self.ep_i += 1
        self.phase, self.t, self.k, self.n_steps = "agent", 0, 0, 0
        if first:
            return self._obs(), True, True
        tour_done = prev_tour != self.current_episode.tour_id
        produce_action = True
        if not tour_done and self.oracle_phases:
            self.phase = "oracle_start"
            self.k = self.current_episode.start_steps
            if self.k == 0:
                self.phase = "agent"
            else:
                produce_action = False
        return self._obs(), tour_done, produce_action

    def step(self, action, auto_reset):
        agent_done, sim_done, tour_done, produce_action, info = True, False, False, False, {}
        self.dtw_data.append({"position": self._position(), "phase": self.phase,
                              "episode_id": self.current_episode.episode_id})
        self.n_steps += 1
        if self.phase == "agent":
            ep = self.current_episode
            agent_done = ep.expert[min(self.t, len(ep.expert) - 1)] in (0, -1)  # the script ends the episode
            self.t += 1
            produce_action = True
            info = self._metrics(agent_done)
            if agent_done:
                produce_action = False
                self.phase = "oracle_goal"
                if not self.oracle_phases:
                    self.phase = "agent"
                    sim_done = True
                else:
                    self.k = ep.goal_steps
                    if self.k == 0 or not self.oracle_goal_phase:
                        self.phase = "oracle_start"
                        sim_done = True
        elif self.phase == "oracle_goal":
            self.k -= 1
            if self.k == 0:
                self.phase = "oracle_start"
                sim_done = True
        elif self.phase == "oracle_start":
            self.k -= 1
            if self.k == 0:
                self.phase = "agent"
                produce_action = True
        obs = self._obs()
        if agent_done or sim_done:
            info["dtw_data"] = copy.deepcopy(self.dtw_data)  # the worker pipe pickles it: a snapshot
        if auto_reset and sim_done:
            obs, tour_done, produce_action = self.reset()
        return obs, 0.0, agent_done, sim_done, tour_done, produce_action, info


class ScriptedVectorEnv:
    def __init__(self, scripts, iterative, auto_reset, oracle_phases=True):
        self.envs = [_ScriptedEnv(i, eps, iterative, oracle_phases) for i, eps in enumerate(scripts)]
        self.iterative, self.auto_reset = iterative, auto_reset
        self.action_log = []
        self.reset_at_log = []

    @property
    def num_envs(self):
        return len(self.envs)

    @property
    def number_of_episodes(self):
        return [len(e.episodes) for e in self.envs]

    def current_episodes(self):
        return [e.current_episode for e in self.envs]

    def reset(self):
        return [e.reset() if self.iterative else e.reset_episodic() for e in self.envs]

    def reset_at(self, i):
        self.reset_at_log.append((len(self.action_log), self.envs[i].idx))
        return [self.envs[i].reset() if self.iterative else self.envs[i].reset_episodic()]

    def step(self, actions):
        assert len(actions) == len(self.envs)
        self.action_log.append([int(a) for a in actions])
        if self.iterative:
            return [e.step(a, self.auto_reset) for e, a in zip(self.envs, actions)]
        return [e.step_episodic(a, self.auto_reset) for e, a in zip(self.envs, actions)]

    def pause_at(self, i):
        self.envs.pop(i)

    def close(self):
        pass


class _Visual(nn.Module):
    def forward(self, batch):
        d = batch["depth"].float()
        return torch.stack([d.reshape(d.shape[0], -1).sum(1), d.reshape(d.shape[0], -1).mean(1) * 2.0], 1)


class _Rgb(nn.Module):
    def forward(self, batch):
        return batch["feat"][:, :2].float() * 0.5


class ScriptedIterativePolicy:
    """Pure function of its inputs, logs every call: the four masks, previous actions, compacted state rows and
    the map the step saw.  `rgb_encoder.cnn` exists so that the rgb feature hook of the collection loop runs."""

    def __init__(self, with_rgb=True):
        net = SimpleNamespace(num_recurrent_layers=2, depth_encoder=SimpleNamespace(visual_encoder=_Visual()))
        if with_rgb:
            net.rgb_encoder = SimpleNamespace(cnn=_Rgb())
        self.net = net
        self.calls = []
        self.deleted = []
        net.delete_batch_idx = self.deleted.append

    def eval(self):
        pass

    def _act(self, batch, rnn_states, prev_actions, masks, log):
        f = self.net.depth_encoder.visual_encoder(batch)
        if hasattr(self.net, "rgb_encoder"):
            self.net.rgb_encoder.cnn(batch)
        a = ((f[:, 0] * 1000.0).round().long() + prev_actions.view(-1).long() + masks.view(-1).long()) % 4
        log.update({"rows": int(a.shape[0]), "prev": prev_actions.view(-1).tolist(),
                    "rnn_mean": [round(float(x), 6) for x in rnn_states.reshape(rnn_states.shape[0], -1).mean(1)]})
        if "occupancy_map" in batch:
            log["map"] = batch["occupancy_map"].reshape(a.shape[0], -1)[:, 0].tolist()
        self.calls.append(log)
        new_rnn = rnn_states * masks.view(-1, 1, 1).to(rnn_states.dtype) + batch["feat"][:, :1].view(-1, 1, 1).float() * 1e-3
        return a.view(-1, 1), new_rnn

    def act(self, batch, rnn_states, prev_actions, masks, deterministic=False):
        return self._act(batch, rnn_states, prev_actions, masks,
                         {"masks": masks.view(-1).tolist(), "deterministic": bool(deterministic)})

    def act_iterative(self, batch, rnn_states, prev_actions, agent_episode_not_done_masks, sim_episode_not_done_masks,
                      tour_not_done_masks, action_masks, deterministic=False):
        log = {"agent": agent_episode_not_done_masks.view(-1).tolist(), "sim": sim_episode_not_done_masks.view(-1).tolist(),
               "tour": tour_not_done_masks.view(-1).tolist(), "action": action_masks.view(-1).tolist(),
               "deterministic": bool(deterministic)}
        return self._act(batch, rnn_states, prev_actions, agent_episode_not_done_masks, log)


class ScriptedMapper:
    """Stands in for the `*IterativeMapper` obs-transformers (ivlnce_baselines/common/obs_transforms.py:30-134): the
    maps count the steps since the last zero of `not_done_masks` per BATCH ROW (the mapper's state is keyed by
    row and never re-indexed when envs pause: quirk Q5), and the keys the real mapper deletes are deleted."""

    def __init__(self):
        self.count = None

    def __call__(self, batch):
        m = batch["not_done_masks"].view(-1).long().cpu()
        B = m.shape[0]
        if self.count is None:
            self.count = torch.zeros(B, dtype=torch.long)
        self.count[:B] = self.count[:B] * m + 1
        c = self.count[:B].to(torch.uint8)
        batch["occupancy_map"] = c.view(B, 1, 1).expand(B, 2, 2).clone().to(batch["depth"].device)
        batch["semantic_map"] = (c * 2).view(B, 1, 1).expand(B, 2, 2).clone().to(batch["depth"].device)
        for k in ["world_robot_orientation", "world_robot_pose", "semantic", "semantic12", "env_name"]:
            batch.pop(k, None)
        return batch


# two tours per env; env 1's first episode cannot be solved by the expert (-1: stepped with action 0, dropped)
def scripts():
    return [
        [episode("e0a", "T0", [1, 2, 0], goal_steps=2, start_steps=0),
         episode("e0b", "T0", [3, 1, 1, 0], goal_steps=0, start_steps=2),
         episode("e0c", "T1", [2, 0], goal_steps=1, start_steps=1)],
        [episode("e1a", "T2", [-1], goal_steps=0, start_steps=0),
         episode("e1b", "T2", [1, 1, 0], goal_steps=1, start_steps=1),
         episode("e1c", "T3", [2, 3, 0], goal_steps=0, start_steps=0),
         episode("e1d", "T3", [1, 0], goal_steps=1, start_steps=0)],
        [episode("e2a", "T4", [1, 0], goal_steps=0, start_steps=0),
         episode("e2b", "T4", [2, 2, 1, 0], goal_steps=2, start_steps=2),
         # unreachable goal noticed on the last step, oracle steps follow: the skip flag is recomputed on every step
         # from the CURRENT expert value (:343-347), so by the time the sim episode ends it is gone and the episode
         # is stored - pinned as the reference behaves
         episode("e2c", "T5", [1, 1, 2, 1, 1, -1], goal_steps=1, start_steps=0)],
    ]


COLLECT_CASES = {
    # name: (p, data_it, update_size, torch seed, oracle phases)
    "tf_unique_oracle": (1.0, 0, 10, 21, True),     # beta = 1: expert actions, envs pause on repeated episodes
    "beta_half_oracle": (0.5, 1, 8, 22, True),     # beta = 0.5 with oracle phases
    "policy_no_oracle": (0.0, 2, 6, 23, False),    # ORACLE_PHASES off: the agent is teleported between episodes
}
EVAL_CASES = {
    # name: (iterative, ITERATIVE_MAP_RESET, oracle phases)
    "episodic": (False, "iterative", False),
    "iterative_tour_maps": (True, "iterative", True),
    "iterative_episode_maps": (True, "episodic", True),
    "iterative_no_oracle": (True, "iterative", False),
}
