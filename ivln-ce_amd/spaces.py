"""Minimal gym.spaces stand-ins (Box / Dict / Discrete) used when gym is not installed; only the
attributes the hot path reads (`.shape`, `.spaces`, `.n`, `.low`, `.high`, `.dtype`)."""
try:  # pragma: no cover
    from gym.spaces import Box, Dict, Discrete  # type: ignore
    from gym import Space  # type: ignore
except Exception:  # noqa: BLE001
    import numpy as np

    class Space:
        pass

    class Box(Space):
        def __init__(self, low, high, shape=None, dtype=np.float32):
            self.shape = tuple(shape) if shape is not None else np.asarray(low).shape
            self.low = np.full(self.shape, low) if np.isscalar(low) else np.asarray(low)
            self.high = np.full(self.shape, high) if np.isscalar(high) else np.asarray(high)
            self.dtype = np.dtype(dtype)

    class Dict(Space):
        def __init__(self, spaces):
            self.spaces = dict(spaces)

    class Discrete(Space):
        def __init__(self, n):
            self.n = n
