"""t-nDTW (the metric of habitat_extensions/tour_ndtw.py:94-130): one constrained DTW per
tour between the agent's and the ground-truth positions, mapped through exp(-d / (len(gt) * success_distance)) and
averaged over tours with weights proportional to the number of episode changes of the ground-truth tour.

Written from the definition rather than helper by helper.  One pass over a tour's step records (`_track`) yields
everything the metric needs: the positions of the steps the agent itself took (oracle phases are not scored), the
indices where a new episode begins, and the order of the episodes.  The DTW constraint says: the last step of
episode k of the agent's tour aligns with the last step of episode k of the ground truth and with nothing else, and
the same for the first steps - i.e. episodes may not bleed into each other - which is a boolean window over the
cost matrix with those columns closed except for one cell.  The DTW itself is this package's C++
(`csrc/dtw.cpp`, recurrence pinned to the reference's exact DTW by tests/golden/dtw.npz); dtw-python's windowed
call is absent from the image, so the window semantics are restated, with known-answer tests
(tests/test_host_logic.py::test_tour_ndtw_known_answers).  CPU-side metric, not a kernel."""
import ctypes as C
from typing import Dict, List, Sequence, Tuple

import numpy as np

from ._lib import check, lib


def dtw_symmetric1(a, b, window=None) -> float:
    """D[i,j] = |a_i - b_j| + min(D[i-1,j-1], D[i-1,j], D[i,j-1]) over the cells `window` admits; D[-1,-1]."""
    a = np.ascontiguousarray(a, np.float64).reshape(len(a), -1)
    b = np.ascontiguousarray(b, np.float64).reshape(len(b), -1)
    L = lib()
    L.ivln_dtw_symmetric1.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p,
                                      C.POINTER(C.c_double)]
    w = None
    if window is not None:
        window = np.ascontiguousarray(window, np.uint8)
        w = window.ctypes.data_as(C.c_void_p)
    out = C.c_double(0.0)
    check(L.ivln_dtw_symmetric1(a.ctypes.data_as(C.c_void_p), a.shape[0], b.ctypes.data_as(C.c_void_p), b.shape[0],
                                a.shape[1], w, C.byref(out)), "ivln_dtw_symmetric1")
    return out.value


def novel_only(path: Sequence) -> List:
    """Drop every record equal to the one before it (the agent turned or waited in place)."""
    return [p for i, p in enumerate(path) if i == 0 or p != path[i - 1]]


def window_align(query_size: int, reference_size: int, alignments) -> np.ndarray:
    """(query, reference) uint8 mask: all open, except that reference column j of a pair (i, j) admits row i only
    (pairs apply in order: a later pair on the same column - a one-step ground-truth episode - replaces the earlier)."""
    window = np.ones((query_size, reference_size), dtype=np.uint8)
    for i, j in alignments:
        window[:, j] = 0
        window[i, j] = 1
    return window


def _track(path: Sequence[dict]) -> Tuple[np.ndarray, List[int], List[str]]:
    """Agent-phase positions of a tour, the index (into them) of the first step of every episode after the first,
    and the episode ids in the order they were played."""
    positions, starts, order = [], [], []
    for rec in path:
        if rec["phase"] != "agent":
            continue
        ep = rec["episode_id"]
        if not order or ep != order[-1]:
            if order:
                starts.append(len(positions))
            order.append(ep)
        positions.append(rec["position"])
    return np.asarray(positions, np.float64).reshape(len(positions), -1), starts, order


def _episode_changes(path: Sequence[dict]) -> int:
    """The tour's weight: how often consecutive records (of any phase) belong to different episodes - one less than
    the number of episodes for a tour that plays each episode once (so single-episode tours weigh nothing; kept)."""
    return sum(1 for a, b in zip(path, path[1:]) if a["episode_id"] != b["episode_id"])


def tour_score(agent_path: Sequence[dict], gt_path: Sequence[dict], success_distance: float) -> float:
    """exp(-constrained DTW / (ground-truth length * success_distance)) of one tour."""
    ap, a_starts, a_order = _track(novel_only(agent_path))
    gp, g_starts, g_order = _track(gt_path)  # the ground truth is scored as recorded, repeats included
    assert g_order == a_order, "agent and GT episode orders do not match."
    assert len(a_starts) == len(g_starts), "mismatch in number of alignment points."
    pairs = []
    for sa, sg in zip(a_starts, g_starts):
        pairs += [(sa - 1, sg - 1), (sa, sg)]  # where one episode stops and where the next one starts
    d = dtw_symmetric1(ap, gp, window_align(len(ap), len(gp), pairs))
    return float(np.exp(-d / (len(gp) * success_distance)))


def compute_tour_ndtw(agent_paths: Dict[str, List], gt_paths: Dict[str, List], success_distance: float = 3.0,
                      verbose: bool = False) -> float:
    """Episode-change-weighted mean of the per-tour scores; both arguments map tour id -> step records
    {"position": [x, y, z], "phase": "agent" | "oracle_goal" | "oracle_start", "episode_id": ...}."""
    if set(gt_paths) != set(agent_paths):
        raise ValueError("tours are different")
    weights = {tour: _episode_changes(path) for tour, path in gt_paths.items()}
    total = sum(weights.values())
    score = 0.0
    for tour, agent_path in agent_paths.items():
        s = tour_score(agent_path, gt_paths[tour], success_distance)
        if verbose:
            print(round(s, 4), "\t", sum(1 for p in gt_paths[tour] if p["phase"] == "agent"))
        score += s * (weights[tour] / total)
    return score
