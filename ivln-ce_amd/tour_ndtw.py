"""t-nDTW: constrained DTW over whole tours, episode-count weighted (same functions and results as
habitat_extensions/tour_ndtw.py:8-130); the DTW itself is our own C++ (csrc/dtw.cpp) because
dtw-python is not vendored.  CPU-side metric, not a kernel."""
import ctypes as C
from collections import defaultdict
from typing import Dict, List

import numpy as np

from ._lib import check, lib


def dtw_symmetric1(a, b, window=None) -> float:
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


def compute_episodes_per_tour(tours: Dict[str, List]) -> Dict[str, int]:
    eps_per_tour = defaultdict(int)
    for tour_id, path in tours.items():
        for i in range(1, len(path)):
            if path[i]["episode_id"] != path[i - 1]["episode_id"]:
                eps_per_tour[tour_id] += 1
    return eps_per_tour


def window_align(query_size, reference_size, alignments):
    """tour_ndtw.py:19-27: every alignment column j only admits its own row i."""
    window = np.ones((query_size, reference_size), dtype=np.uint8)
    for (i, j) in alignments:
        window[:, j] = 0
        window[i, j] = 1
    return window


def extract_ep_order(path):
    eps = [p["episode_id"] for p in path]
    single = [eps[i - 1] for i in range(1, len(eps)) if eps[i - 1] != eps[i]]
    single.append(eps[-1])
    return single


def alignments_from_paths(agent_path, gt_path):
    gt_path = [p for p in gt_path if p["phase"] == "agent"]
    agent_path = [p for p in agent_path if p["phase"] == "agent"]
    assert extract_ep_order(gt_path) == extract_ep_order(agent_path), "agent and GT episode orders do not match."

    def points(path):
        out = []
        for i in range(1, len(path)):
            if path[i]["episode_id"] != path[i - 1]["episode_id"]:
                out += [i - 1, i]  # stopping point, starting point
        return out

    a, g = points(agent_path), points(gt_path)
    assert len(a) == len(g), "mismatch in number of alignment points."
    return list(zip(a, g))


def novel_only(path):
    if len(path) <= 1:
        return path
    new_path = [path[0]]
    for i in range(1, len(path)):
        if path[i - 1] != path[i]:
            new_path.append(path[i])
    return new_path


def aggregate_scores(t_ndtws, episodes_per_tour):
    total_eps = sum(episodes_per_tour.values())
    return sum(t * (episodes_per_tour[tid] / total_eps) for tid, t in t_ndtws.items())


def compute_tour_ndtw(agent_paths: Dict[str, List], gt_paths: Dict[str, List], success_distance: float = 3.0,
                      verbose: bool = False) -> float:
    if not set(gt_paths.keys()) == set(agent_paths.keys()):
        raise ValueError("tours are different")
    t_ndtws = {}
    for tour_id, agent_path in agent_paths.items():
        agent_path = novel_only(agent_path)
        gt_path = gt_paths[tour_id]  # the reference overwrites its novel_only(gt) result (tour_ndtw.py:112-113)
        alignments = alignments_from_paths(agent_path, gt_path)
        ap = [p["position"] for p in agent_path if p["phase"] == "agent"]
        gtp = [p["position"] for p in gt_path if p["phase"] == "agent"]
        d = dtw_symmetric1(ap, gtp, window_align(len(ap), len(gtp), alignments))
        t_ndtws[tour_id] = float(np.exp(-d / (len(gtp) * success_distance)))
        if verbose:
            print(round(t_ndtws[tour_id], 4), "\t", len(gtp))
    return aggregate_scores(t_ndtws, compute_episodes_per_tour(gt_paths))
