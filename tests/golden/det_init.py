"""Deterministic, data-free initialiser shared by the golden generator (applied to the reference's
own modules) and the tests (applied to the oracle / HIP modules): every tensor of a state_dict is
filled from a torch.Generator seeded with crc32(key), so identical keys+shapes give identical
weights without shipping 54 MB of parameters.  Exercises state_dict key compatibility as a side
effect."""
import zlib

import torch


def det_value(key: str, like: torch.Tensor, seed: int = 0, conv_gain: float = 2.0 ** 0.5):
    """The tensor det_fill gives the state_dict entry `key` (shape / dtype of `like`)."""
    v = like
    g = torch.Generator().manual_seed((zlib.crc32(key.encode()) + seed) % (2 ** 31))
    if not v.dtype.is_floating_point:
        return v.clone()  # num_batches_tracked
    name = key.rsplit(".", 1)[-1]
    if v.dim() == 0:
        return v.clone()  # _scale buffer
    if name == "running_var":
        return 0.5 + torch.rand(v.shape, generator=g)
    if name == "running_mean":
        return 0.1 * torch.randn(v.shape, generator=g)
    if v.dim() == 1 and name == "weight":  # norm gammas
        return 1.0 + 0.1 * torch.randn(v.shape, generator=g)
    if v.dim() == 1:  # biases / norm betas
        return 0.05 * torch.randn(v.shape, generator=g)
    fan_in = v[0].numel()
    std = conv_gain * (1.0 / fan_in) ** 0.5 if v.dim() == 4 else (1.0 / fan_in) ** 0.5
    if "embedding" in key:
        std = 0.5
    out = std * torch.randn(v.shape, generator=g)
    if "embedding_layer" in key:
        out[0].zero_()  # PAD row (padding_idx = 0)
    return out


def det_fill(module: torch.nn.Module, seed: int = 0, prefix: str = "", conv_gain: float = 2.0 ** 0.5):
    sd = module.state_dict()
    module.load_state_dict({k: det_value(prefix + k, v, seed, conv_gain) for k, v in sd.items()})
    return module
