"""Host side of the egocentric mapper: owns the device state handle and mirrors the reference's
`MappingModule` objects (ivlnce_baselines/common/mapping_module/mapper.py:904-1028) and
`setup_mapping_module.py:13-89`.  All arithmetic is in csrc/mapper.hip.
"""
import ctypes as C
import math
import os
from dataclasses import dataclass
from typing import Dict, List, Optional

import numpy as np
import torch

from . import _lib
from ._lib import check, dptr, lib, stream_ptr

# ivln_mapper_step_posed: camera transforms derived inside the step's first kernel (6 launches per step instead of
# frames + 6); IVLN_MAPPER_POSED=0 keeps the separate ivln_mapper_frames launch (A/B, tests of the older entry points)
STEP_POSED = os.environ.get("IVLN_MAPPER_POSED", "1") != "0"


@dataclass
class CameraParameters:  # mapper.py:336-340
    vertical_fov_radians: float
    features_spatial_dimensions: tuple
    height_clip: float


@dataclass
class MapDimensions:  # mapper.py:89-99
    height_meters: float
    width_meters: float
    resolution_meters: float

    @property
    def num_rows(self):
        return math.ceil(self.height_meters / self.resolution_meters)

    @property
    def num_cols(self):
        return math.ceil(self.width_meters / self.resolution_meters)


def extract_camera_parameters(depth_sensor_params, map_sensor_params) -> CameraParameters:
    """setup_mapping_module.py:13-42: vfov = HFOV * H / W (degrees -> radians)."""
    vfov_deg = depth_sensor_params.HFOV * (depth_sensor_params.HEIGHT / depth_sensor_params.WIDTH)
    return CameraParameters(
        vertical_fov_radians=float(np.deg2rad(vfov_deg)),
        features_spatial_dimensions=(depth_sensor_params.HEIGHT, depth_sensor_params.WIDTH),
        height_clip=map_sensor_params.height_clip,
    )


def extract_egocentric_map_parameters(map_sensor_params) -> MapDimensions:
    """setup_mapping_module.py:45-54."""
    return MapDimensions(
        height_meters=map_sensor_params.height_meters,
        width_meters=map_sensor_params.width_meters,
        resolution_meters=map_sensor_params.resolution_meters,
    )


class OccupancySemanticMapMemory:
    """mapper.py:620-648: persistent (B,rows,cols) uint8 buffers returned by reference."""

    def __init__(self, b_max, rows, cols, device):
        self._occ = torch.zeros((b_max, rows, cols), dtype=torch.uint8, device=device)
        self._sem = torch.zeros((b_max, rows, cols), dtype=torch.uint8, device=device)
        self.batch_size = b_max

    @property
    def occupancy(self):
        return self._occ[: self.batch_size]

    @property
    def semantic(self):
        return self._sem[: self.batch_size]


class MappingModule:
    """Drop-in for mapper.py:904-944 (`create_*_mapper` factories :950-1028).

    mode "iterative": labels come from `semantic12` (gt) or from `semantics_module(observations)`
    (predicted, e.g. RedNet) and the world cloud is built online.
    mode "known": the world cloud of each env is loaded from `{maps_location}/{env_name}.npz`
    when its episode resets (mapper.py:851-881).
    """

    def __init__(
        self,
        device: torch.device,
        camera_parameters: Optional[CameraParameters],
        map_dimensions: MapDimensions,
        mode: str = "iterative",
        semantics_module=None,
        maps_location: Optional[str] = None,
        b_max: int = 64,
        world_capacity: int = 0,
        table_cells: int = 0,
    ):
        if device.type != "cuda":
            raise _lib.IvlnError("the HIP mapper needs a GPU device (no CPU fallback)")
        self.device = device
        self.mode = mode
        self.camera_parameters = camera_parameters
        self.map_dimensions = map_dimensions
        self.semantics_module = semantics_module
        self.maps_location = maps_location
        self.b_max = b_max
        self._world_capacity = world_capacity
        self._table_cells = table_cells
        self._h = None
        self._hw = None
        self.map_memory = OccupancySemanticMapMemory(
            b_max, map_dimensions.num_rows, map_dimensions.num_cols, device
        )
        self._T = torch.empty((b_max, 4, 4), dtype=torch.float32, device=device)
        self._rot = torch.empty((b_max, 3, 3), dtype=torch.float32, device=device)
        self._known_cache: Dict[str, tuple] = {}

    # -- handle management ---------------------------------------------------------------
    def _ensure_handle(self, H, W):
        if self._h is not None and self._hw == (H, W):
            return
        if self._h is not None:
            lib().ivln_mapper_destroy(self._h)
        h = C.c_void_p()
        vfov = self.camera_parameters.vertical_fov_radians if self.camera_parameters else math.pi / 2
        md = self.map_dimensions
        with torch.cuda.device(self.device):
            check(
                lib().ivln_mapper_create(
                    self.b_max, H, W, vfov, md.height_meters, md.width_meters, md.resolution_meters,
                    self._world_capacity, self._table_cells, C.byref(h),
                ),
                "ivln_mapper_create",
            )
        self._h, self._hw = h, (H, W)
        if getattr(self, "_width", (0, 0)) != (0, 0):
            check(lib().ivln_mapper_set_launch_width(self._h, *self._width), "ivln_mapper_set_launch_width")

    def __del__(self):
        try:
            if self._h is not None:
                lib().ivln_mapper_destroy(self._h)
                self._h = None
        except Exception:  # noqa: BLE001
            pass

    def reset(self):
        if self._h is not None:
            check(lib().ivln_mapper_reset(self._h, stream_ptr()), "ivln_mapper_reset")

    # -- frames -----------------------------------------------------------------------------
    def frames(self, pose: torch.Tensor, orientation: torch.Tensor):
        """(B,3) f32 pose + (B,2) f64 [elevation, heading] -> T (B,4,4), rot (B,3,3) on device
        (core.py:6-37, mapper.py:38-48, 132-138)."""
        B = pose.shape[0]
        pose = pose.to(self.device, torch.float32).contiguous()
        orientation = orientation.to(self.device, torch.float64).contiguous()
        check(
            lib().ivln_mapper_frames(dptr(pose), dptr(orientation), B, dptr(self._T), dptr(self._rot), stream_ptr()),
            "ivln_mapper_frames",
        )
        return self._T[:B], self._rot[:B], pose

    # -- one step -----------------------------------------------------------------------------
    def forward(self, observations: Dict[str, torch.Tensor], T=None, rot=None) -> OccupancySemanticMapMemory:
        """observations: the dict the reference's `Mapper.forward` receives
        (obs_transforms.py:79-103): depth (B,H,W,1) f32, semantic12 (B,H,W,1) u8 (gt) or rgb
        (pred), world_robot_pose (B,3), world_robot_orientation (B,2) f64, not_done_masks (B,1),
        env_name list[str]."""
        depth = observations["depth"]
        B, H, W = depth.shape[0], depth.shape[1], depth.shape[2]
        if B > self.b_max:
            raise _lib.IvlnError(f"batch {B} > b_max {self.b_max}")
        self._ensure_handle(H, W)
        depth = depth.to(torch.float32).contiguous()
        not_done = observations["not_done_masks"].reshape(-1).to(torch.uint8).contiguous()
        posed = T is None and self.mode == "iterative" and STEP_POSED
        if posed:  # the transforms are derived inside the step's first kernel (ivln_mapper_step_posed)
            pose = observations["world_robot_pose"].to(self.device, torch.float32).contiguous()
            orientation = observations["world_robot_orientation"].to(self.device, torch.float64).contiguous()
        elif T is None:
            T, rot, pose = self.frames(observations["world_robot_pose"], observations["world_robot_orientation"])
        else:
            pose = observations["world_robot_pose"].to(self.device, torch.float32).contiguous()
            T = T.to(self.device, torch.float32).contiguous()
            rot = rot.to(self.device, torch.float32).contiguous()
        mem = self.map_memory
        mem.batch_size = B
        s = stream_ptr()
        if self.mode == "iterative":
            if self.semantics_module is not None:
                labels = self.semantics_module(observations)  # (B,1,H,W) or (B,H,W) u8
            else:
                if "semantic12" not in observations or observations["semantic12"] is None:
                    raise Exception("Semantic Sensor not in use")  # mapper.py:660-661
                labels = observations["semantic12"]
            labels = labels.reshape(B, H, W).to(torch.uint8).contiguous()
            if getattr(self, "dry_run", False):
                # a capture's warm-up after the first one (graphed.GraphedRollout(warmup_mapper=False)): everything in front of
                # the mapper has run on this stream - RedNet's workspaces exist now - but the world cloud belongs to the
                # rollout in progress and is not stepped; the maps are whatever the last real step left
                return mem
            if posed:
                check(
                    lib().ivln_mapper_step_posed(
                        self._h, dptr(depth), dptr(labels), dptr(pose), dptr(orientation), dptr(not_done), B,
                        dptr(mem._occ), dptr(mem._sem), dptr(self._T), dptr(self._rot), s,
                    ),
                    "ivln_mapper_step_posed",
                )
            else:
                check(
                    lib().ivln_mapper_step(
                        self._h, dptr(depth), dptr(labels), dptr(T), dptr(pose), dptr(rot), dptr(not_done), B,
                        dptr(mem._occ), dptr(mem._sem), s,
                    ),
                    "ivln_mapper_step",
                )
        else:
            check(lib().ivln_mapper_known_begin(self._h, dptr(not_done), B, s), "ivln_mapper_known_begin")
            finished = (not_done == 0).nonzero().reshape(-1).tolist()
            for b in finished:  # mapper.py:871-879
                xyz, sem = self._load_known(observations["env_name"][b])
                check(
                    lib().ivln_mapper_load_known(self._h, int(b), dptr(xyz), dptr(sem), xyz.shape[0], s),
                    "ivln_mapper_load_known",
                )
            check(
                lib().ivln_mapper_known_raster(self._h, dptr(pose), dptr(rot), B, dptr(mem._occ), dptr(mem._sem), s),
                "ivln_mapper_known_raster",
            )
        return mem

    __call__ = forward

    # -- the same step in two halves (ivln_mapper_step_begin / _finish): `begin` needs depth and pose only, so with predicted
    #    semantics it can be enqueued on another stream beside the network whose labels `finish` waits for ------------------
    def begin(self, observations: Dict[str, torch.Tensor]):
        """Camera transforms, local min / max and the keep-highest arg-max of this step's depth frames (2 launches on the
        current stream).  Iterative mode with the posed entry only; `finish` must follow, ordered behind it."""
        if self.mode != "iterative" or not STEP_POSED:
            raise _lib.IvlnError("MappingModule.begin / finish: iterative mode with the posed step entry only")
        depth = observations["depth"]
        B, H, W = depth.shape[0], depth.shape[1], depth.shape[2]
        if B > self.b_max:
            raise _lib.IvlnError(f"batch {B} > b_max {self.b_max}")
        self._ensure_handle(H, W)
        self.map_memory.batch_size = B
        if getattr(self, "dry_run", False):
            return
        depth = depth.to(torch.float32).contiguous()
        not_done = observations["not_done_masks"].reshape(-1).to(torch.uint8).contiguous()
        pose = observations["world_robot_pose"].to(self.device, torch.float32).contiguous()
        orientation = observations["world_robot_orientation"].to(self.device, torch.float64).contiguous()
        # (finish reads pose / T / rot through the pointers begin was given: they stay alive on the module)
        self._open = (depth, not_done, pose, orientation)
        check(lib().ivln_mapper_step_begin(self._h, dptr(depth), dptr(pose), dptr(orientation), dptr(not_done), B,
                                           dptr(self.map_memory._occ), dptr(self._T), dptr(self._rot), stream_ptr()),
              "ivln_mapper_step_begin")

    def finish(self, observations: Dict[str, torch.Tensor]) -> OccupancySemanticMapMemory:
        """Labels (gt, or the semantics module's) -> world-cloud merge -> maps (4 launches behind the labels)."""
        mem = self.map_memory
        depth = observations["depth"]
        B, H, W = depth.shape[0], depth.shape[1], depth.shape[2]
        if self.semantics_module is not None:
            labels = self.semantics_module(observations)
        else:
            if "semantic12" not in observations or observations["semantic12"] is None:
                raise Exception("Semantic Sensor not in use")  # mapper.py:660-661
            labels = observations["semantic12"]
        labels = labels.reshape(B, H, W).to(torch.uint8).contiguous()
        from . import rednet as _rednet

        _rednet._stage_done("labels")  # (a capturing GraphedRollout cuts its graph here and waits for `begin`'s event)
        if getattr(self, "dry_run", False):
            return mem
        depth_c, not_done, pose, _ = self._open
        check(lib().ivln_mapper_step_finish(self._h, dptr(depth_c), dptr(labels), dptr(self._T), dptr(pose), dptr(self._rot),
                                            dptr(not_done), B, dptr(mem._occ), dptr(mem._sem), stream_ptr()),
              "ivln_mapper_step_finish")
        return mem

    def _load_known(self, env_name):
        if env_name not in self._known_cache:
            with np.load(os.path.join(self.maps_location, f"{env_name}.npz")) as f:  # mapper.py:283-294
                xyz = torch.from_numpy(np.ascontiguousarray(f["xyz"], dtype=np.float32)).to(self.device)
                sem = torch.from_numpy(np.ascontiguousarray(f["semantics"]).astype(np.int64).astype(np.uint8)).to(self.device)
            self._known_cache[env_name] = (xyz.contiguous(), sem.contiguous())
        return self._known_cache[env_name]

    # -- introspection (tests) -----------------------------------------------------------------
    def status(self):
        n = C.c_int64(0)
        code = lib().ivln_mapper_status(self._h, C.byref(n), stream_ptr())
        return code, n.value

    def set_launch_width(self, local_blocks: int = 0, world_blocks: int = 0):
        """Workgroups of the local- / world-cloud kernels (0 = full width).  Narrow when the mapper runs beside a
        latency-bound chain on another stream (graphed.py, split replay); results do not depend on it."""
        self._width = (int(local_blocks), int(world_blocks))
        if self._h is not None:
            check(lib().ivln_mapper_set_launch_width(self._h, *self._width), "ivln_mapper_set_launch_width")

    def check_status(self):
        code, n = self.status()
        check(code, "mapper device status")
        return n

    def world_cloud(self):
        """World cloud in the reference's order (ascending key of the last keep-highest):
        xyz (n,3) f32, batch (n,) i32, semantics (n,) u8 - numpy, for parity tests."""
        code, n = self.status()
        xyz = torch.empty((max(n, 1), 3), dtype=torch.float32, device=self.device)
        meta = torch.empty((max(n, 1),), dtype=torch.int32, device=self.device)
        rank = torch.empty((max(n, 1),), dtype=torch.int64, device=self.device)
        nn = C.c_int64(0)
        check(
            lib().ivln_mapper_world_export(self._h, dptr(xyz), dptr(meta), dptr(rank), n, C.byref(nn), stream_ptr()),
            "ivln_mapper_world_export",
        )
        xyz, meta, rank = xyz[:n].cpu().numpy(), meta[:n].cpu().numpy().view(np.uint32), rank[:n].cpu().numpy()
        order = np.argsort(rank, kind="stable")
        return xyz[order], (meta[order] >> 8).astype(np.int32), (meta[order] & 0xFF).astype(np.uint8)


def create_gt_semantics_iterative_mapper(device, camera_parameters, map_dimensions, **kw) -> MappingModule:
    """mapper.py:991-998."""
    return MappingModule(device, camera_parameters, map_dimensions, mode="iterative", **kw)


def create_predicted_semantics_iterative_mapper(device, camera_parameters, map_dimensions, **kw) -> MappingModule:
    """mapper.py:1001-1008: labels = argmax of RedNet(rgb, depth)."""
    from .rednet import PredictSemantics

    return MappingModule(
        device, camera_parameters, map_dimensions, mode="iterative", semantics_module=PredictSemantics(device), **kw
    )


def create_gt_semantics_known_mapper(device, map_dimensions, **kw) -> MappingModule:
    """mapper.py:1011-1017."""
    return MappingModule(device, None, map_dimensions, mode="known", maps_location="data/known_maps/gt_semantics", **kw)


def create_predicted_semantics_known_mapper(device, map_dimensions, **kw) -> MappingModule:
    """mapper.py:1020-1028."""
    return MappingModule(
        device, None, map_dimensions, mode="known", maps_location="data/known_maps/predicted_semantics", **kw
    )
