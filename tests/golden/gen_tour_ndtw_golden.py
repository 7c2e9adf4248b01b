"""Golden t-nDTW values from the REFERENCE's own `compute_tour_ndtw` (habitat_extensions/tour_ndtw.py:8-130) on seeded
random tours (build container only): novel-only filtering, agent-phase selection, episode alignment pairs, window
construction, exp(-d / (len(gt) * 3)) and the episode-change weighting all run in the reference's code.  Only
`dtw.dtw` (dtw-python 1.3.0, absent from the image) is a stand-in: a plain numpy symmetric1 DTW over the boolean
window the reference's own `window_align_func` returns - that one call stays "parity
unpinned", everything around it is pinned by this fixture.  -> tests/golden/tour_ndtw.json"""
import importlib.util
import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
np.bool = bool  # the reference uses the alias numpy removed (tour_ndtw.py:20)


def _dtw(x, y, step_pattern="symmetric1", window_type=None, window_args=None):
    x, y = np.asarray(x, float), np.asarray(y, float)
    n, m = len(x), len(y)
    win = window_type(None, None, query_size=n, reference_size=m, **(window_args or {}))
    D = np.full((n, m), np.inf)
    for i in range(n):
        for j in range(m):
            if not win[i, j]:
                continue
            d = float(np.linalg.norm(x[i] - y[j]))
            if i == 0 and j == 0:
                D[i, j] = d
                continue
            best = min(D[i - 1, j - 1] if i and j else np.inf, D[i - 1, j] if i else np.inf, D[i, j - 1] if j else np.inf)
            D[i, j] = d + best
    return types.SimpleNamespace(distance=D[-1, -1])


sys.modules["dtw"] = types.SimpleNamespace(dtw=_dtw)
spec = importlib.util.spec_from_file_location("ref_tour_ndtw", "/root/reference/habitat_extensions/tour_ndtw.py")
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)


def tour(rs, n_eps, with_oracle, one_step_episode=False):
    """A ground-truth tour and an agent tour over the same episodes: the agent wanders (repeated positions where it
    turns), and - with oracle phases - is conveyed between episodes by steps that must not be scored."""
    gt, ag = [], []
    pos = rs.rand(3) * 4
    for e in range(n_eps):
        eid = f"ep{e}"
        n = 1 if (one_step_episode and e == 1) else int(rs.randint(2, 7))
        pts = pos + np.cumsum(rs.randn(n, 3) * 0.25, axis=0)
        for p in pts:
            gt.append({"position": p.tolist(), "phase": "agent", "episode_id": eid})
        a = pts[0] + rs.randn(3) * 0.05
        for _ in range(int(rs.randint(2, 9))):
            if rs.rand() < 0.3 and ag and ag[-1]["episode_id"] == eid:
                ag.append(dict(ag[-1]))  # turned in place: identical record, dropped by novel_only
            else:
                a = a + rs.randn(3) * 0.25
                ag.append({"position": a.tolist(), "phase": "agent", "episode_id": eid})
        if with_oracle:
            for ph in ("oracle_goal", "oracle_start"):
                for _ in range(int(rs.randint(0, 3))):
                    a = a + rs.randn(3) * 0.25
                    ag.append({"position": a.tolist(), "phase": ph, "episode_id": eid})
        pos = pts[-1]
    return ag, gt


if __name__ == "__main__":
    rs = np.random.RandomState(17)
    cases = []
    for k in range(6):
        agent, gt = {}, {}
        for t in range(int(rs.randint(1, 4))):
            agent[f"t{t}"], gt[f"t{t}"] = tour(rs, int(rs.randint(2, 5)), with_oracle=k % 2 == 1, one_step_episode=k == 4)
        cases.append({"agent": agent, "gt": gt, "success_distance": 3.0 if k < 4 else 1.5,
                      "score": float(ref.compute_tour_ndtw(agent, gt, 3.0 if k < 4 else 1.5))})
        print(k, len(agent), cases[-1]["score"])
    json.dump(cases, open(os.path.join(HERE, "tour_ndtw.json"), "w"))
