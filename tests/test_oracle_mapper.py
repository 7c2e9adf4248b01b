"""Pins the C oracle (oracle/mapper_ref.c) to the goldens produced by the reference's own
MappingModule (tests/golden/gen_mapper_golden.py): maps, world cloud contents AND order must be
bit-identical at every step."""
import glob
import os

import numpy as np
import pytest

from oracle.mapper_ref import MapperRef

CASES = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "mapper_*.npz")))


@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(p)[7:-4] for p in CASES])
@pytest.mark.parametrize("own_frames", [False, True])
def test_oracle_matches_reference_golden(path, own_frames):
    g = np.load(path)
    H, W, steps = int(g["H"]), int(g["W"]), int(g["steps"])
    m = MapperRef(H, W)
    for t in range(steps):
        kw = {} if own_frames else dict(T=g[f"T_{t}"], rot=g[f"rot_{t}"])
        occ, sem = m.step(
            g[f"depth_{t}"], g[f"semantic12_{t}"], g[f"pose_{t}"], g[f"orientation_{t}"], g[f"not_done_{t}"], **kw
        )
        xyz, b, s = m.world()
        assert xyz.shape[0] == int(g[f"world_n_{t}"]), f"world size step {t}"
        assert np.array_equal(occ, g[f"occ_{t}"]), f"occupancy step {t}"
        assert np.array_equal(sem, g[f"sem_{t}"]), f"semantic step {t}"
        if f"world_xyz_{t}" in g:
            assert np.array_equal(xyz.view(np.uint32), g[f"world_xyz_{t}"].view(np.uint32))
            assert np.array_equal(b, g[f"world_b_{t}"])
            assert np.array_equal(s, g[f"world_sem_{t}"])


def test_frames_match_reference_trig():
    g = np.load(CASES[0])
    for t in range(int(g["steps"])):
        T, rot = MapperRef.frames(g[f"pose_{t}"], g[f"orientation_{t}"])
        assert np.array_equal(T.view(np.uint32), g[f"T_{t}"].view(np.uint32))
        assert np.array_equal(rot.view(np.uint32), g[f"rot_{t}"].view(np.uint32))


def test_oracle_known_map_mode_matches_reference_golden():
    """Known-map mode of the C oracle against the reference's own `create_known_mapper` run
    (tests/golden/gen_known_map_golden.py): maps bit-exact every step, world-cloud size, two envs sharing a scene,
    reloads on reset."""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "known_map.npz"))
    names = [str(x) for x in g["env_names"]]
    clouds = {n: (g[f"scene_{n}_xyz"], g[f"scene_{n}_semantics"]) for n in set(names)}
    ref = MapperRef(8, 8)
    for t in range(int(g["steps"])):
        occ, sem = ref.known_step(clouds, names, g[f"pose_{t}"], g[f"orientation_{t}"], g[f"not_done_{t}"])
        assert np.array_equal(occ, g[f"occ_{t}"]), f"occupancy step {t}"
        assert np.array_equal(sem, g[f"sem_{t}"]), f"semantic step {t}"
        assert ref.world()[0].shape[0] == int(g[f"world_n_{t}"])
        assert np.array_equal(ref.world()[1], g[f"world_b_{t}"])
