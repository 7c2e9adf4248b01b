"""Golden vectors for the known-map mapper mode by running the REFERENCE's own `create_known_mapper`
(ivlnce_baselines/common/mapping_module/mapper.py:968-985: `GetGTWorldSemanticPointcloud` :851-881 +
`SemanticPointcloud.from_npz_file` :283-294 + the shared height filter / raster) on seeded scene clouds written to
temporary `{env_name}.npz` files.

Build container only:  python tests/golden/gen_known_map_golden.py  ->  tests/golden/known_map.npz
(scene clouds, per-step poses / masks / env names, expected occupancy + semantic maps and world-cloud sizes)."""
import os
import sys
import tempfile

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _ref_shim  # noqa: E402

_ref_shim.install()
torch.set_num_threads(1)  # the duplicate-index map store is only deterministic single-threaded (quirk Q3)

from ivlnce_baselines.common.mapping_module import mapper as M  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def scene(seed, n):
    rs = np.random.RandomState(seed)
    xyz = np.stack([rs.uniform(-4, 4, n), rs.uniform(0.0, 2.6, n), rs.uniform(-4, 4, n)], 1).astype(np.float32)
    # snap x/z to a 5 cm lattice: many points per 10 cm map cell, so the raster's last-writer rule (file order)
    # and the label-0 drop decide most cells
    xyz[:, [0, 2]] = (np.round(xyz[:, [0, 2]] / 0.05) * 0.05).astype(np.float32)
    sem = rs.randint(0, 13, n).astype(np.int64)
    return xyz, sem


if __name__ == "__main__":
    scenes = {"sceneA": scene(1, 1800), "sceneB": scene(2, 1300)}
    B, steps = 3, 5
    env_names = ["sceneA", "sceneB", "sceneA"]  # two envs in the same scene: its cloud is loaded twice
    resets = {0: [0, 1, 2], 2: [1], 3: [0, 2]}
    d = dict(B=B, steps=steps, env_names=np.array(env_names))
    for k, (xyz, sem) in scenes.items():
        d[f"scene_{k}_xyz"], d[f"scene_{k}_semantics"] = xyz, sem
    with tempfile.TemporaryDirectory() as tmp:
        for k, (xyz, sem) in scenes.items():
            np.savez(os.path.join(tmp, f"{k}.npz"), xyz=xyz, semantics=sem)
        dims = M.MapDimensions(height_meters=6.4, width_meters=6.4, resolution_meters=0.1)
        mm = M.create_known_mapper(torch.device("cpu"), dims, maps_location=tmp)
        g = torch.Generator().manual_seed(3)
        pose = torch.zeros(B, 3)
        pose[:, 1] = torch.tensor([1.25, 1.4, 0.9])
        pose[:, 0] = torch.tensor([0.0, -1.0, 1.5])
        heading = torch.rand(B, generator=g, dtype=torch.float64) * 6.28 - 3.14
        for t in range(steps):
            nd = torch.ones(B, 1, dtype=torch.uint8)
            for b in resets.get(t, []):
                nd[b, 0] = 0
            elev = (torch.rand(B, generator=g, dtype=torch.float64) - 0.5) * 0.02
            orient = torch.stack([elev, heading.clone()], 1)
            ep = M.EpisodesInfo(nd.clone(), list(env_names))
            obs = M.Observations(semantics=None, depth_normalized=torch.zeros(B, 1, 8, 8), rgb=None)
            st = M.RobotCurrentState(pose=pose.clone(), elevation=orient[:, 0].clone(), heading=orient[:, 1].clone())
            mem = mm(ep, obs, st)
            w = mm.get_world_semantic_pointcloud()
            d[f"pose_{t}"], d[f"orientation_{t}"], d[f"not_done_{t}"] = pose.numpy().copy(), orient.numpy().copy(), nd.numpy()
            d[f"occ_{t}"], d[f"sem_{t}"] = mem.occupancy.clone().numpy(), mem.semantic.clone().numpy()
            d[f"world_n_{t}"] = np.int64(w.xyz.shape[0])
            d[f"world_b_{t}"] = w.batch_indices.clone().numpy().astype(np.int32)
            pose[:, 0] += (-0.25 * torch.sin(heading)).float()
            pose[:, 2] += (-0.25 * torch.cos(heading)).float()
            heading += np.deg2rad(15.0) * (1 if t % 2 == 0 else -2)
    path = os.path.join(OUT, "known_map.npz")
    np.savez_compressed(path, **d)
    print("world sizes", [int(d[f"world_n_{t}"]) for t in range(steps)], "occ cells",
          [int(d[f"occ_{t}"].sum()) for t in range(steps)], "sem cells", [int((d[f"sem_{t}"] > 0).sum()) for t in range(steps)],
          os.path.getsize(path) // 1024, "KiB")
