"""Deterministic, data-free initialiser shared by the golden generator (applied to the reference's
own modules) and the tests (applied to the oracle / HIP modules): every tensor of a state_dict is
filled from a torch.Generator seeded with crc32(key), so identical keys+shapes give identical
weights without shipping 54 MB of parameters.  Exercises state_dict key compatibility as a side
effect."""
import zlib

import torch


def det_fill(module: torch.nn.Module, seed: int = 0, prefix: str = "", conv_gain: float = 2.0 ** 0.5):
    sd = module.state_dict()
    new = {}
    for k, v in sd.items():
        g = torch.Generator().manual_seed((zlib.crc32((prefix + k).encode()) + seed) % (2 ** 31))
        if not v.dtype.is_floating_point:
            new[k] = v.clone()  # num_batches_tracked
            continue
        name = k.rsplit(".", 1)[-1]
        if v.dim() == 0:
            new[k] = v.clone()  # _scale buffer
        elif name == "running_var":
            new[k] = 0.5 + torch.rand(v.shape, generator=g)
        elif name == "running_mean":
            new[k] = 0.1 * torch.randn(v.shape, generator=g)
        elif v.dim() == 1 and name == "weight":  # norm gammas
            new[k] = 1.0 + 0.1 * torch.randn(v.shape, generator=g)
        elif v.dim() == 1:  # biases / norm betas
            new[k] = 0.05 * torch.randn(v.shape, generator=g)
        else:
            fan_in = v[0].numel()
            std = conv_gain * (1.0 / fan_in) ** 0.5 if v.dim() == 4 else (1.0 / fan_in) ** 0.5
            if "embedding" in k:
                std = 0.5
            new[k] = std * torch.randn(v.shape, generator=g)
            if "embedding_layer" in k:
                new[k][0].zero_()  # PAD row (padding_idx = 0)
    module.load_state_dict(new)
    return module
