"""Golden vectors for RedNet / PredictSemantics from the REFERENCE's own modules
(ivlnce_baselines/common/mapping_module/rednet.py, mapper.py:703-800) with det_init weights.
Build container only:  python tests/golden/gen_rednet_golden.py  -> tests/golden/rednet.npz"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _ref_shim  # noqa: E402

_ref_shim.install()
torch.set_num_threads(8)
from det_init import det_fill  # noqa: E402

from ivlnce_baselines.common.mapping_module import mapper as M  # noqa: E402
from ivlnce_baselines.common.mapping_module.rednet import RedNet  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))

if __name__ == "__main__":
    cfg = {"arch": "rednet", "resnet_pretrained": False, "finetune": True, "SUNRGBD_pretrained_weights": "",
           "n_classes": 13, "upsample_prediction": True, "load_model": ""}
    net = det_fill(RedNet(cfg), seed=1, conv_gain=0.6).eval()
    ps = M.PredictSemantics()
    ps.model = net
    g = torch.Generator().manual_seed(5)
    B, H, W = 2, 64, 64
    rgb = torch.randint(0, 256, (B, 56, 56, 3), generator=g, dtype=torch.uint8)
    depth = torch.rand(B, H, W, 1, generator=g)
    obs = M.Observations(semantics=None, depth_normalized=depth.permute(0, 3, 1, 2), rgb=rgb.permute(0, 3, 1, 2))
    ps.setup_normalization(obs)
    with torch.no_grad():
        rgb_n = ps.rgb_normalization(obs.rgb.float() / 255.0)
        dep_n = ps.depth_normalization(obs.depth_normalized)
        scores = net(rgb_n, dep_n)
    labels = scores.argmax(1, keepdims=True).to(torch.uint8)
    np.savez_compressed(os.path.join(OUT, "rednet.npz"), rgb=rgb.numpy(), depth=depth.numpy(),
                        rgb_n=rgb_n.numpy(), scores=scores.numpy(), labels=labels.numpy())
    print("scores", scores.shape, float(scores.abs().max()), "labels hist", np.bincount(labels.flatten().numpy(), minlength=13))
    print(os.path.getsize(os.path.join(OUT, "rednet.npz")) // 1024, "KiB")
