"""Generate golden vectors for the egocentric mapper by running the REFERENCE's own
`MappingModule` (ivlnce_baselines/common/mapping_module/mapper.py:904-944, built by
`create_gt_semantics_iterative_mapper` :991-998) on seeded synthetic observations.

Run in the build container only:  python tests/golden/gen_mapper_golden.py
Writes tests/golden/mapper_*.npz (inputs + expected outputs; data only, no reference source).
torch.set_num_threads(1): the reference's duplicate-index map store (mapper.py:571) is only
deterministic single-threaded.
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _ref_shim  # noqa: E402

_ref_shim.install()
torch.set_num_threads(1)

from ivlnce_baselines.common.mapping_module import mapper as M  # noqa: E402
from ivlnce_baselines.common.mapping_module.projector import _transform3D  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def make_sequence(seed, B, H, W, steps, resets, planar=False, far_apart=False):
    """Seeded synthetic rollout: wall-like depth, random labels, 0.25 m / 15 deg random walk."""
    g = torch.Generator().manual_seed(seed)
    pose = torch.zeros(B, 3)
    pose[:, 1] = 1.25
    if far_apart:
        pose[:, 0] = torch.arange(B).float() * 7.5
        pose[:, 2] = -torch.arange(B).float() * 3.25
    heading = torch.rand(B, generator=g, dtype=torch.float64) * 6.28 - 3.14
    frames = []
    for t in range(steps):
        if planar:
            depth = torch.full((B, H, W, 1), 0.3) + 0.05 * torch.rand(B, 1, 1, 1, generator=g)
        else:
            col = torch.rand(B, 1, W, 1, generator=g)
            depth = 0.2 + 0.6 * col + 0.02 * torch.rand(B, H, W, 1, generator=g)
            depth = depth.clamp(0, 1)
            # sprinkle invalid readings (0 and 1) like a real depth sensor
            inval = torch.rand(B, H, W, 1, generator=g)
            depth = torch.where(inval < 0.02, torch.zeros_like(depth), depth)
            depth = torch.where(inval > 0.985, torch.ones_like(depth), depth)
        sem = torch.randint(0, 13, (B, H, W, 1), generator=g, dtype=torch.uint8)
        elev = (torch.rand(B, generator=g, dtype=torch.float64) - 0.5) * 0.02
        not_done = torch.ones(B, 1, dtype=torch.uint8)
        for (tt, b) in resets:
            if tt == t:
                not_done[b, 0] = 0
        if t == 0:
            not_done[:] = 0
        frames.append(
            dict(
                depth=depth.clone(),
                semantic12=sem,
                world_robot_pose=pose.clone(),
                world_robot_orientation=torch.stack([elev, heading.clone()], 1),
                not_done_masks=not_done,
            )
        )
        # random walk: forward 0.25 m along heading, or turn +-15 deg
        act = torch.randint(0, 3, (B,), generator=g)
        for b in range(B):
            if act[b] == 0:
                pose[b, 0] += float(-0.25 * np.sin(heading[b].item()))
                pose[b, 2] += float(-0.25 * np.cos(heading[b].item()))
            elif act[b] == 1:
                heading[b] += np.deg2rad(15.0)
            else:
                heading[b] -= np.deg2rad(15.0)
    return frames


def run_reference(frames, H, W, hfov_deg=90.0):
    cam = M.CameraParameters(
        vertical_fov_radians=float(np.deg2rad(hfov_deg * (H / W))),
        features_spatial_dimensions=(H, W),
        height_clip=0.1,
    )
    dims = M.MapDimensions(height_meters=6.4, width_meters=6.4, resolution_meters=0.1)
    mm = M.create_gt_semantics_iterative_mapper(torch.device("cpu"), cam, dims)
    out = []
    for f in frames:
        B = f["depth"].shape[0]
        ep = M.EpisodesInfo(f["not_done_masks"].clone(), ["s"] * B)
        obs = M.Observations(
            semantics=f["semantic12"].permute(0, 3, 1, 2),
            depth_normalized=f["depth"].permute(0, 3, 1, 2),
            rgb=None,
        )
        st = M.RobotCurrentState(
            pose=f["world_robot_pose"].clone(),
            elevation=f["world_robot_orientation"][:, 0].clone(),
            heading=f["world_robot_orientation"][:, 1].clone(),
        )
        T = _transform3D(st.pose, st.elevation + torch.pi, st.heading)
        rot = M.rotate_around_y_matrix(-st.heading)
        mem = mm(ep, obs, st)
        w = mm.get_world_semantic_pointcloud()
        out.append(
            dict(
                occ=mem.occupancy.clone().numpy(),
                sem=mem.semantic.clone().numpy(),
                T=T.numpy().copy(),
                rot=rot.numpy().copy(),
                world_xyz=w.xyz.clone().numpy(),
                world_b=w.batch_indices.clone().numpy().astype(np.int32),
                world_sem=w.semantics.clone().numpy(),
            )
        )
    return out


def save_case(name, seed, B, H, W, steps, resets, keep_world_steps, **kw):
    frames = make_sequence(seed, B, H, W, steps, resets, **kw)
    outs = run_reference(frames, H, W)
    d = dict(B=B, H=H, W=W, steps=steps, seed=seed)
    for t, (f, o) in enumerate(zip(frames, outs)):
        d[f"depth_{t}"] = f["depth"].numpy()
        d[f"semantic12_{t}"] = f["semantic12"].numpy()
        d[f"pose_{t}"] = f["world_robot_pose"].numpy()
        d[f"orientation_{t}"] = f["world_robot_orientation"].numpy()
        d[f"not_done_{t}"] = f["not_done_masks"].numpy()
        d[f"occ_{t}"] = o["occ"]
        d[f"sem_{t}"] = o["sem"]
        d[f"T_{t}"] = o["T"]
        d[f"rot_{t}"] = o["rot"]
        d[f"world_n_{t}"] = np.int64(o["world_xyz"].shape[0])
        if t in keep_world_steps:
            d[f"world_xyz_{t}"] = o["world_xyz"]
            d[f"world_b_{t}"] = o["world_b"]
            d[f"world_sem_{t}"] = o["world_sem"]
    path = os.path.join(OUT, f"mapper_{name}.npz")
    np.savez_compressed(path, **d)
    print(name, "world sizes", [int(d[f"world_n_{t}"]) for t in range(steps)],
          "occ cells", [int(d[f"occ_{t}"].sum()) for t in range(steps)],
          os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    save_case("b2_64", seed=11, B=2, H=64, W=64, steps=6, resets=[(3, 1)], keep_world_steps=[0, 2, 5])
    save_case("b4_32", seed=12, B=4, H=32, W=32, steps=5, resets=[(2, 0), (2, 3), (4, 2)], keep_world_steps=[4])
    save_case("b1_256", seed=13, B=1, H=256, W=256, steps=2, resets=[], keep_world_steps=[1])
    save_case("planar_ties", seed=14, B=2, H=48, W=48, steps=3, resets=[], keep_world_steps=[2], planar=True)
    save_case("far_apart", seed=15, B=3, H=32, W=32, steps=3, resets=[(1, 2)], keep_world_steps=[2], far_apart=True)
