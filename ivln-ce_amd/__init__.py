"""MI355X-native MapCMA hot path for IVLN-CE: egocentric semantic mapper, MapCMA policy
forward/backward and the DAgger rollout/update, behind the reference's registry names.

Compute runs in hand-written HIP kernels (libivln_hip.so, C ABI in include/ivln_hip.h) bound with
ctypes; PyTorch only owns device memory, streams and torch.distributed.  There is NO CPU fallback:
ops raise if the HIP library is missing or a tensor is not on a GPU.
"""
from . import config  # noqa: F401
from .registry import baseline_registry  # noqa: F401

__all__ = ["config", "baseline_registry"]
