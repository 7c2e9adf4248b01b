"""Per-episode nDTW / SDTW (habitat_extensions/measures.py:152-230) for the evaluators of this package.

    nDTW = exp(-DTW(agent path, gt path) / (len(gt path) * SUCCESS_DISTANCE)),  SDTW = success * nDTW

`TASK.NDTW.FDTW False` -> exact DTW: the reference's own `habitat_extensions.utils.dtw` recurrence
(utils.py:155-221), here the host C++ `ivln_dtw_symmetric1` (csrc/dtw.cpp), pinned against goldens produced
by that reference function (tests/golden/dtw.npz).
`TASK.NDTW.FDTW True` (the reference default, habitat_extensions/config/default.py:111) -> FastDTW with
radius 1, i.e. the third-party `fastdtw` package (requirements.txt) which is not in this image: its published
algorithm (Salvador & Chan 2007: halve, recurse, project the path, band of +-radius, constrained DTW) is
restated below; parity UNPINNED (no reference fixture exercises it).
CPU-side metrics, not kernels.
"""
from collections import defaultdict
from typing import List, Sequence

import numpy as np

from .tour_ndtw import dtw_symmetric1


def euclidean_distance(a, b) -> float:
    return float(np.linalg.norm(np.asarray(b, dtype=np.float64) - np.asarray(a, dtype=np.float64), ord=2))


def _dtw_window(x, y, window, dist):
    """DTW restricted to `window` (list of (i, j), row-major) - the inner routine of fastdtw."""
    len_x, len_y = len(x), len(y)
    if window is None:
        window = [(i, j) for i in range(len_x) for j in range(len_y)]
    D = defaultdict(lambda: (float("inf"),))
    D[0, 0] = (0.0, 0, 0)
    for i, j in ((i + 1, j + 1) for i, j in window):
        dt = dist(x[i - 1], y[j - 1])
        D[i, j] = min((D[i - 1, j][0] + dt, i - 1, j), (D[i, j - 1][0] + dt, i, j - 1),
                      (D[i - 1, j - 1][0] + dt, i - 1, j - 1), key=lambda a: a[0])
    path = []
    i, j = len_x, len_y
    while not (i == j == 0):
        path.append((i - 1, j - 1))
        i, j = D[i, j][1], D[i, j][2]
    path.reverse()
    return D[len_x, len_y][0], path


def _reduce_by_half(x):
    return [(x[i] + x[i + 1]) / 2 for i in range(0, len(x) - len(x) % 2, 2)]


def _expand_window(path, len_x, len_y, radius):
    path_ = set(path)
    for i, j in path:
        for a in range(-radius, radius + 1):
            for b in range(-radius, radius + 1):
                path_.add((i + a, j + b))
    window_ = set()
    for i, j in path_:
        window_.update(((i * 2, j * 2), (i * 2, j * 2 + 1), (i * 2 + 1, j * 2), (i * 2 + 1, j * 2 + 1)))
    window = []
    start_j = 0
    for i in range(len_x):
        new_start_j = None
        for j in range(start_j, len_y):
            if (i, j) in window_:
                window.append((i, j))
                if new_start_j is None:
                    new_start_j = j
            elif new_start_j is not None:
                break
        start_j = new_start_j
    return window


def fastdtw(x: Sequence, y: Sequence, radius: int = 1, dist=euclidean_distance):
    """(distance, path) of FastDTW; exact DTW when either sequence is shorter than radius + 2."""
    x = [np.asarray(p, dtype=np.float64) for p in x]
    y = [np.asarray(p, dtype=np.float64) for p in y]

    def rec(a, b):
        if len(a) < radius + 2 or len(b) < radius + 2:
            return _dtw_window(a, b, None, dist)
        _, path = rec(_reduce_by_half(a), _reduce_by_half(b))
        return _dtw_window(a, b, _expand_window(path, len(a), len(b), radius), dist)

    return rec(x, y)


def dtw_distance(locations, gt_locations, fdtw: bool = False) -> float:
    if fdtw:
        return float(fastdtw(locations, gt_locations, radius=1)[0])
    return float(dtw_symmetric1(locations, gt_locations))


def ndtw(locations, gt_locations, success_distance: float = 3.0, fdtw: bool = False) -> float:
    """measures.py:204-212."""
    return float(np.exp(-dtw_distance(locations, gt_locations, fdtw) / (len(gt_locations) * success_distance)))


class NDTW:
    """Stateful mirror of the habitat Measure (measures.py:152-212): positions are appended only when they
    differ from the last one, the metric is refreshed on every update."""

    cls_uuid = "ndtw"

    def __init__(self, success_distance: float = 3.0, fdtw: bool = True):
        self.success_distance, self.fdtw = success_distance, fdtw
        self.locations: List = []
        self.gt_locations: List = []
        self._metric = 0.0

    def reset_metric(self, gt_locations, position):
        self.locations = []
        self.gt_locations = [list(map(float, p)) for p in gt_locations]
        self.update_metric(position)

    def update_metric(self, position):
        position = [float(v) for v in position]
        if self.locations and position == self.locations[-1]:
            return
        self.locations.append(position)
        self._metric = ndtw(self.locations, self.gt_locations, self.success_distance, self.fdtw)

    def get_metric(self) -> float:
        return self._metric


class SDTW:
    """measures.py:215-238: success-weighted nDTW."""

    cls_uuid = "sdtw"

    @staticmethod
    def get_metric(success: float, ndtw_value: float) -> float:
        return float(success) * float(ndtw_value)
