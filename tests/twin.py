"""Test helper: call the SAME `ivln_*` C-ABI entry point in libivln_hip.so (device pointers) and in its CPU twin
oracle/libivln_ref.so (host pointers) - SURVEY.md section 8b, "a CPU twin exports the same symbols for parity tests".
Test infrastructure only."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_TWIN = None

vp, i32, i64, f32, f64 = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_double


def desc_type():
    from ivln_ce_amd.ops import GemmDesc

    return GemmDesc


def _sigs(L):
    D = desc_type()
    L.ivln_gemm_f32.argtypes = [C.POINTER(D), vp]
    L.ivln_groupnorm_f32.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, i32, f32, i32, i64, i64, i32, i64, i64, i64, vp, vp, vp]
    L.ivln_mapper_create.argtypes = [i32, i32, i32, f64, f64, f64, f64, i64, i64, C.POINTER(vp)]
    L.ivln_mapper_destroy.argtypes = [vp]
    L.ivln_mapper_frames.argtypes = [vp, vp, i32, vp, vp, vp]
    L.ivln_mapper_step.argtypes = [vp, vp, vp, vp, vp, vp, vp, i32, vp, vp, vp]
    L.ivln_mapper_known_begin.argtypes = [vp, vp, i32, vp]
    L.ivln_mapper_load_known.argtypes = [vp, i32, vp, vp, i64, vp]
    L.ivln_mapper_known_raster.argtypes = [vp, vp, vp, i32, vp, vp, vp]
    L.ivln_mapper_status.argtypes = [vp, C.POINTER(i64), vp]
    L.ivln_strerror.restype = C.c_char_p
    L.ivln_strerror.argtypes = [i32]
    return L


def twin():
    """oracle/libivln_ref.so (built on demand by oracle/Makefile)."""
    global _TWIN
    if _TWIN is None:
        so = os.path.join(ROOT, "oracle", "libivln_ref.so")
        if not os.path.exists(so):
            subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "libivln_ref.so"])
        _TWIN = _sigs(C.CDLL(so))
    return _TWIN


def hp(a):
    """host pointer of a contiguous numpy array (or None)"""
    return None if a is None else a.ctypes.data_as(vp)


def conv_desc(ptr, x, w, out, stride=1, pad=0, scale=None, shift=None, residual=None, relu=False, grouped=False):
    """ivln_gemm_desc of a forward NCHW convolution exactly as ivln_ce_amd.ops.conv2d fills it (non-deferred, split
    heuristic off: splits = 1); `ptr(array)` turns an operand into the pointer the library expects."""
    from ivln_ce_amd import ops

    N, Cin, H, W = x.shape
    Cout, _, KH, KW = w.shape[-4:]
    Ho, Wo = (H + 2 * pad - KH) // stride + 1, (W + 2 * pad - KW) // stride + 1
    d = desc_type()()
    d.A, d.B, d.D = ptr(w), ptr(x), ptr(out)
    d.M, d.N, d.K = Cout, N * Ho * Wo, Cin * KH * KW
    d.amode, d.dmode = ops.A_MK, ops.D_NCHW
    d.lda = d.K
    d.Cin, d.Hin, d.Win, d.Hout, d.Wout = Cin, H, W, Ho, Wo
    d.stride, d.pad, d.dil = stride, pad, 1
    d.HoWo = Ho * Wo
    d.bmode = ops.B_CONV1X1 if (KH == 1 and pad == 0) else (ops.B_CONV_K3 if KH == 3 else ops.B_CONV_K7)
    d.scale, d.shift, d.residual = ptr(scale), ptr(shift), ptr(residual)
    d.relu = int(relu)
    d.splits = 1
    if grouped:
        d.grp_imgs, d.a_grp_stride = N // w.shape[0], Cout * Cin * KH * KW
    return d


def check(L, code, what):
    assert code == 0, f"{what}: {L.ivln_strerror(code).decode()} ({code})"


def np32(a):
    return np.ascontiguousarray(a, dtype=np.float32)
