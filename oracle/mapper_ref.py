"""ctypes binding of oracle/libmapper_ref.so (the C restatement in mapper_ref.c).
TEST INFRASTRUCTURE ONLY - see the header of mapper_ref.c."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "libmapper_ref.so")
        if not os.path.exists(so):
            subprocess.check_call(["make", "-C", _HERE, "libmapper_ref.so"])
        L = C.CDLL(so)
        L.mapper_ref_create.restype = C.c_void_p
        L.mapper_ref_create.argtypes = [C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double]
        L.mapper_ref_destroy.argtypes = [C.c_void_p]
        L.mapper_ref_reset.argtypes = [C.c_void_p]
        L.mapper_ref_frames.argtypes = [C.c_int] + [C.c_void_p] * 4
        L.mapper_ref_step.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 8
        L.mapper_ref_clear_done.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.mapper_ref_raster.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 4
        L.mapper_ref_load_known.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int64]
        L.mapper_ref_world_size.restype = C.c_int64
        L.mapper_ref_world_size.argtypes = [C.c_void_p]
        L.mapper_ref_world_get.argtypes = [C.c_void_p] * 4
        _LIB = L
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class MapperRef:
    """CPU oracle of MappingModule (mapper.py:904-944) with gt/replayed labels."""

    def __init__(self, H=256, W=256, hfov_deg=90.0, height_m=6.4, width_m=6.4, res_m=0.1):
        self.H, self.W = H, W
        vfov = float(np.deg2rad(hfov_deg * (H / W)))
        self.rows = int(np.ceil(height_m / res_m))
        self.cols = int(np.ceil(width_m / res_m))
        self.h = lib().mapper_ref_create(H, W, vfov, height_m, width_m, res_m)

    def __del__(self):
        if getattr(self, "h", None):
            lib().mapper_ref_destroy(self.h)
            self.h = None

    @staticmethod
    def frames(pose, orientation):
        pose = np.ascontiguousarray(pose, np.float32)
        orientation = np.ascontiguousarray(orientation, np.float64)
        B = pose.shape[0]
        T = np.zeros((B, 4, 4), np.float32)
        rot = np.zeros((B, 3, 3), np.float32)
        lib().mapper_ref_frames(B, _p(pose), _p(orientation), _p(T), _p(rot))
        return T, rot

    def step(self, depth, labels, pose, orientation, not_done, T=None, rot=None):
        depth = np.ascontiguousarray(depth, np.float32).reshape(-1, self.H, self.W)
        labels = np.ascontiguousarray(labels, np.uint8).reshape(-1, self.H, self.W)
        B = depth.shape[0]
        pose = np.ascontiguousarray(pose, np.float32)
        if T is None:
            T, rot = self.frames(pose, orientation)
        T = np.ascontiguousarray(T, np.float32)
        rot = np.ascontiguousarray(rot, np.float32)
        nd = np.ascontiguousarray(not_done, np.uint8).reshape(-1)
        occ = np.zeros((B, self.rows, self.cols), np.uint8)
        sem = np.zeros((B, self.rows, self.cols), np.uint8)
        lib().mapper_ref_step(self.h, B, _p(depth), _p(labels), _p(T), _p(pose), _p(rot), _p(nd), _p(occ), _p(sem))
        return occ, sem

    def world(self):
        n = lib().mapper_ref_world_size(self.h)
        xyz = np.zeros((n, 3), np.float32)
        b = np.zeros((n,), np.int32)
        s = np.zeros((n,), np.uint8)
        lib().mapper_ref_world_get(self.h, _p(xyz), _p(b), _p(s))
        return xyz, b, s

    # -- known-map mode (mapper.py:851-881) --------------------------------------------------------------
    def known_step(self, clouds, env_names, pose, orientation, not_done):
        """One step of the known-map mapper: drop the clouds of finished envs, load `clouds[env_name]` =
        (xyz f32 (n,3), semantics u8 (n,)) for each of them in batch order, raster."""
        pose = np.ascontiguousarray(pose, np.float32)
        nd = np.ascontiguousarray(not_done, np.uint8).reshape(-1)
        B = pose.shape[0]
        _, rot = self.frames(pose, orientation)
        L = lib()
        L.mapper_ref_clear_done(self.h, B, _p(nd))
        for b in range(B):
            if nd[b] == 0:
                xyz, sem = clouds[env_names[b]]
                xyz = np.ascontiguousarray(xyz, np.float32)
                sem = np.ascontiguousarray(sem).astype(np.uint8)
                L.mapper_ref_load_known(self.h, b, _p(xyz), _p(sem), len(sem))
        occ = np.zeros((B, self.rows, self.cols), np.uint8)
        sem_o = np.zeros((B, self.rows, self.cols), np.uint8)
        L.mapper_ref_raster(self.h, B, _p(pose), _p(np.ascontiguousarray(rot)), _p(occ), _p(sem_o))
        return occ, sem_o
