"""ctypes loader for libivln_hip.so.  Fails loudly: there is no CPU or eager-PyTorch fallback."""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# (IVLN_HIP_LIB: another build of the same library - A/B measurements of a kernel variant in one gpurun call; never set in production)
_SO = os.environ.get("IVLN_HIP_LIB") or os.path.join(_HERE, "libivln_hip.so")
_LIB = None

vp = C.c_void_p
i32 = C.c_int
i64 = C.c_int64
f32 = C.c_float
f64 = C.c_double


IVLN_E_UNSUPPORTED = -5  # include/ivln_hip.h: the entry point declines this shape / device (callers take their other path)


class IvlnError(RuntimeError):
    pass


def _sig(L):
    L.ivln_strerror.restype = C.c_char_p
    L.ivln_strerror.argtypes = [i32]
    L.ivln_version.restype = i32
    L.ivln_mapper_create.argtypes = [i32, i32, i32, f64, f64, f64, f64, i64, i64, C.POINTER(vp)]
    L.ivln_mapper_destroy.argtypes = [vp]
    L.ivln_mapper_reset.argtypes = [vp, vp]
    L.ivln_mapper_frames.argtypes = [vp, vp, i32, vp, vp, vp]
    L.ivln_mapper_step.argtypes = [vp, vp, vp, vp, vp, vp, vp, i32, vp, vp, vp]
    L.ivln_mapper_step_posed.argtypes = [vp, vp, vp, vp, vp, vp, i32, vp, vp, vp, vp, vp]
    L.ivln_mapper_step_begin.argtypes = [vp, vp, vp, vp, vp, i32, vp, vp, vp, vp]
    L.ivln_mapper_step_finish.argtypes = [vp, vp, vp, vp, vp, vp, vp, i32, vp, vp, vp]
    L.ivln_mapper_known_begin.argtypes = [vp, vp, i32, vp]
    L.ivln_mapper_load_known.argtypes = [vp, i32, vp, vp, i64, vp]
    L.ivln_mapper_known_raster.argtypes = [vp, vp, vp, i32, vp, vp, vp]
    L.ivln_mapper_status.argtypes = [vp, C.POINTER(i64), vp]
    L.ivln_mapper_world_export.argtypes = [vp, vp, vp, vp, i64, C.POINTER(i64), vp]


def lib():
    """The loaded HIP library.  Raises if it has not been built (run __graft_entry__.build())."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(_SO):
            raise IvlnError(
                f"{_SO} not found: build it with `python ivln-ce_amd/build.py` "
                "(there is no CPU fallback for the HIP hot path)"
            )
        L = C.CDLL(_SO)
        _sig(L)
        _LIB = L
    return _LIB


def check(code: int, what: str = ""):
    if code != 0:
        raise IvlnError(f"{what}: {lib().ivln_strerror(code).decode()} ({code})")


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream_ptr() -> int:
    """hipStream_t of torch's current stream (so torch.cuda.graph capture sees our launches).  Through torch's raw
    accessor when it exists: `torch.cuda.current_stream()` builds a Stream object and resolves the device index on every
    call (~10 us - as much as the dispatch of the launch it precedes in an eager rollout step)."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def dptr(t: torch.Tensor) -> int:
    """Device pointer of a contiguous GPU tensor; anything else is a usage error."""
    if not t.is_cuda:
        raise IvlnError("HIP hot path needs GPU tensors (no CPU fallback); got device " + str(t.device))
    if not t.is_contiguous():
        raise IvlnError("tensor must be contiguous")
    return t.data_ptr()
