"""Scripted vector env + arithmetic-free stand-in policy shared by `gen_rollout_golden.py` (which drives the
REFERENCE's own `DaggerTrainer._update_dataset` with them, build container only) and `tests/test_host_logic.py`
(which drives this package's `_update_dataset` with the same objects and compares the stored trajectories with
the golden).  Everything here is deterministic given the seed: the only randomness of the rollout is the beta-mix
draw (`torch.rand` on the default CPU generator, dagger_trainer.py:423-427), seeded by the caller.

The env scripts exercise: an expert action of -1 on the first step of an episode (the episode is stepped with
action 0 and dropped: dagger_trainer.py:469-472, 352), unequal episode lengths, envs that run out of new
episodes (paused when beta == 1.0, :312-316, 392-412)."""
from types import SimpleNamespace

import numpy as np
import torch
import torch.nn as nn

EXPERT_UUID = "shortest_path_sensor"


class ScriptedEnvs:
    """The slice of habitat's VectorEnv the DAgger rollout uses (num_envs, reset, step, pause_at,
    current_episodes, close).  Env e cycles through its episodes; an episode is done when its script says STOP
    (expert action 0) and the agent steps (whatever the agent did - like the reference's expert-driven stop,
    the script ends the episode), after which the env auto-resets to its next episode."""

    def __init__(self, scripts, obs_dim=3):
        # scripts[e] = list of episodes; an episode = list of expert actions ending with 0 (STOP)
        self.scripts = scripts
        self.obs_dim = obs_dim
        self.live = list(range(len(scripts)))
        self.ep = [0] * len(scripts)
        self.t = [0] * len(scripts)
        self.steps = 0
        self.action_log = []

    @property
    def num_envs(self):
        return len(self.live)

    def _episode(self, e):
        return self.ep[e] % len(self.scripts[e])

    def current_episodes(self):
        return [SimpleNamespace(episode_id=f"env{e}_ep{self._episode(e)}") for e in self.live]

    def _obs(self, e):
        k, t = self._episode(e), self.t[e]
        base = float(100 * e + 10 * k + t)
        return {
            "instruction": {"tokens": np.array([e + 1, k + 1, 7, 0, 0], dtype=np.int64), "text": "unused"},
            "depth": np.full((2, 2, 1), base / 1000.0, dtype=np.float32),
            "progress": np.array([t / max(1, len(self.scripts[e][k]) - 1)], dtype=np.float64),
            EXPERT_UUID: np.array([self.scripts[e][k][t]], dtype=np.float64),
            "feat": np.arange(self.obs_dim, dtype=np.float32) + base,
        }

    def reset(self):
        return [self._obs(e) for e in self.live]

    def step(self, actions):
        assert len(actions) == len(self.live)
        self.action_log.append([int(a) for a in actions])
        out = []
        for e in self.live:
            k = self._episode(e)
            done = self.scripts[e][k][self.t[e]] in (0, -1)  # STOP (or an unreachable goal) ends the episode
            if done:
                self.ep[e] += 1
                self.t[e] = 0
            else:
                self.t[e] += 1
            out.append((self._obs(e), 0.0, done, {}))
        self.steps += 1
        return out

    def pause_at(self, i):
        self.live.pop(i)

    def close(self):
        pass


class _Visual(nn.Module):
    """Stands in for `net.depth_encoder.visual_encoder`: the rollout caches its OUTPUT through a forward hook
    (dagger_trainer.py:317-323), so it must be a module whose forward is called inside `act`."""

    def forward(self, batch):
        d = batch["depth"].float()
        return torch.stack([d.reshape(d.shape[0], -1).sum(1), d.reshape(d.shape[0], -1).mean(1) * 2.0], 1)


class ScriptedPolicy:
    """`act` is a pure function of the batch (no sampling), so reference and port see identical actions; what is
    under test is everything AROUND it: beta-mixing with the expert, the -1 skip, previous actions, masks, state
    compaction when envs pause, feature caching, what lands in a stored trajectory."""

    def __init__(self):
        self.net = SimpleNamespace(num_recurrent_layers=2,
                                   depth_encoder=SimpleNamespace(visual_encoder=_Visual()))
        self.calls = []

    def act(self, batch, rnn_states, prev_actions, masks, deterministic=False):
        f = self.net.depth_encoder.visual_encoder(batch)
        a = ((f[:, 0] * 1000.0).round().long() + prev_actions.view(-1).long() + masks.view(-1).long()) % 4
        self.calls.append({"rows": int(a.shape[0]), "masks": masks.view(-1).tolist(), "prev": prev_actions.view(-1).tolist(),
                           "rnn_mean": [round(float(x), 6) for x in rnn_states.reshape(rnn_states.shape[0], -1).mean(1)]})
        # the state carries the env identity so that pause compaction is visible in the next call's log
        new_rnn = rnn_states * masks.view(-1, 1, 1).to(rnn_states.dtype) + batch["feat"][:, :1].view(-1, 1, 1).float() * 1e-3
        return a.view(-1, 1), new_rnn


SCRIPTS = [
    [[1, 0], [1, 3, 0]],                      # runs out of new episodes first, while the others are mid-episode: its
    #                                           pause shifts their rows (quirk Q12: the reference compacts the device
    #                                           rows but not its host-side `observations` / `episodes` lists)
    [[1, 2, 1, 0], [3, 0], [1, 1, 0]],
    [[-1], [2, 2, 3, 1, 0], [1, 0]],          # first episode: expert cannot reach the goal -> skipped
]

CASES = {
    # name: (p, data_it, update_size, torch seed)
    "teacher_forcing_unique": (1.0, 0, 7, 11),   # beta = 1: expert actions only, envs pause on repeated episodes
    "beta_quarter": (0.5, 2, 7, 12),             # beta = 0.25: mostly the policy's own actions
    "policy_only": (0.0, 3, 5, 13),              # beta = 0
}
