"""Golden vectors for the DTW recurrence behind nDTW / SDTW / t-nDTW, produced by the REFERENCE's own exact
DTW (`habitat_extensions/utils.py:155-221`, the function `NDTW` uses when `TASK.NDTW.FDTW` is False,
`habitat_extensions/measures.py:163,204-212`): D[i][j] = d(x_i, y_j) + min(D[i-1][j-1], D[i-1][j], D[i][j-1]).
Run in the build container only:  python tests/golden/gen_dtw_golden.py   -> tests/golden/dtw.npz
(data only: path pairs, their DTW distance and nDTW = exp(-d / (len(gt) * 3.0))).
"""
import importlib.util
import os
import sys
import types

import numpy as np

REF = "/root/reference/habitat_extensions/utils.py"
OUT = os.path.dirname(os.path.abspath(__file__))


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def load_reference_utils():
    """The module imports habitat / quaternion at the top; only `dtw` (pure numpy) is used here."""
    _stub("quaternion", quaternion=object)
    _stub("habitat")
    _stub("habitat.core")
    _stub("habitat.core.utils", try_cv2_import=lambda: None)
    _stub("habitat.tasks")
    _stub("habitat.tasks.utils", cartesian_to_polar=None)
    _stub("habitat.utils")
    _stub("habitat.utils.geometry_utils", quaternion_rotate_vector=None)
    _stub("habitat.utils.visualizations", maps=None)
    _stub("habitat.utils.visualizations.maps")
    _stub("habitat.utils.visualizations.utils", draw_collision=None, images_to_video=None)
    _stub("habitat_baselines")
    _stub("habitat_baselines.common")
    _stub("habitat_baselines.common.tensorboard_utils", TensorboardWriter=None)
    sys.modules["habitat.utils.visualizations"].maps = sys.modules["habitat.utils.visualizations.maps"]
    hext = _stub("habitat_extensions")
    hext.maps = _stub("habitat_extensions.maps")
    spec = importlib.util.spec_from_file_location("ref_hab_ext_utils", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    ref = load_reference_utils()
    rng = np.random.RandomState(20260)
    euclid = lambda a, b: float(np.linalg.norm(np.array(b) - np.array(a)))  # noqa: E731  (measures.py:36-41)
    out = {}
    cases = [(1, 1), (1, 7), (9, 1), (5, 5), (12, 30), (40, 17), (64, 64), (120, 95)]
    for k, (n, m) in enumerate(cases):
        # random walks with 0.25 m steps, like agent / gt trajectories
        x = np.cumsum(rng.randn(n, 3) * 0.25, axis=0)
        y = x[np.linspace(0, n - 1, m).astype(int)] + rng.randn(m, 3) * 0.1 if k % 2 else np.cumsum(rng.randn(m, 3) * 0.25, axis=0)
        d = ref.dtw(x.tolist(), y.tolist(), dist=euclid)[0]
        out[f"x_{k}"], out[f"y_{k}"] = x, y
        out[f"d_{k}"] = np.float64(d)
        out[f"ndtw_{k}"] = np.float64(np.exp(-d / (len(y) * 3.0)))
    out["n_cases"] = np.int64(len(cases))
    np.savez(os.path.join(OUT, "dtw.npz"), **out)
    print("wrote dtw.npz:", [(c, round(float(out[f"d_{k}"]), 4)) for k, c in enumerate(cases)])


if __name__ == "__main__":
    main()
