"""Golden vectors for the Latent-CMA policy: the REFERENCE's own `LatentCMAPolicy`
(ivlnce_baselines/models/latent_cma_policy.py:28-497) and `TorchVisionResNet50` wrapper
(models/encoders/resnet_encoders.py:118-229) run on seeded inputs with det_init weights; torchvision's
ResNet-50 body itself comes from oracle/torchvision_ref.py (not in the image: unpinned).
Build container only:  python tests/golden/gen_latent_golden.py  -> tests/golden/latent_act_{plain,tourmem}.npz
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _ref_shim  # noqa: E402

_ref_shim.install()
torch.set_num_threads(8)
from det_init import det_fill  # noqa: E402

from ivlnce_baselines.models.latent_cma_policy import LatentCMAPolicy  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def make_policy(variant):
    cfg = _ref_shim.default_model_config()
    cfg.MODEL.policy_name = "LatentCMAPolicy"
    cfg.MODEL.tour_memory_variant = variant
    cfg.MODEL.tour_memory = variant
    sp = sys.modules["gym.spaces"]
    space = sp.Dict({"depth": sp.Box(0.0, 1.0, (256, 256, 1), np.float32), "rgb": sp.Box(0, 255, (224, 224, 3), np.uint8),
                     "instruction": sp.Box(0, 2504, (200,), np.int64)})
    pol = LatentCMAPolicy.from_config(cfg, space, sp.Discrete(4))
    det_fill(pol, seed=0, conv_gain=1.0)
    return pol.eval()


def gen(variant, name):
    B = 2
    g = torch.Generator().manual_seed(77 + int(variant))
    pol = make_policy(variant)
    L = pol.net.num_recurrent_layers
    rnn = torch.zeros(B, L, 512)
    prev = torch.zeros(B, 1, dtype=torch.long)
    instr = torch.zeros(B, 200, dtype=torch.long)
    for b, n in enumerate([80, 31]):
        instr[b, :n] = torch.randint(2, 2504, (n,), generator=g)
    d = {"instruction": instr.numpy(), "B": B, "L": L}
    feats = {}
    pol.net.rgb_encoder.register_forward_hook(lambda m, i, o: feats.__setitem__("rgb", o.detach().clone()))
    for t in range(3):
        col = torch.rand(B, 1, 256, 1, generator=g)
        depth = (0.2 + 0.6 * col + 0.02 * torch.rand(B, 256, 256, 1, generator=g)).clamp(0, 1)
        rgb = torch.randint(0, 256, (B, 224, 224, 3), generator=g, dtype=torch.uint8)
        ep = torch.ones(B, 1, dtype=torch.uint8)
        tour = torch.ones(B, 1, dtype=torch.uint8)
        if t == 0:
            ep[:] = 0
            tour[:] = 0
        if t == 2:
            ep[1] = 0  # a new episode inside the same tour for env 1
        obs = {"depth": depth, "rgb": rgb, "instruction": instr}
        with torch.no_grad():
            feats.clear()
            features, rnn_out = pol.net(obs, rnn.clone(), prev, action_masks=ep,
                                        episode_masks=(ep if variant else None), tour_masks=(tour if variant else None))
            logits = pol.action_distribution(features).logits
            act, rnn_out2 = pol.act_iterative(obs, rnn.clone(), prev, ep, ep, tour, ep, deterministic=True)
        assert torch.equal(rnn_out, rnn_out2)
        d.update({f"depth_{t}": depth.numpy(), f"rgb_{t}": rgb.numpy(), f"ep_{t}": ep.numpy(), f"tour_{t}": tour.numpy(),
                  f"prev_{t}": prev.numpy(), f"rnn_in_{t}": rnn.numpy(), f"features_{t}": features.numpy(),
                  f"rnn_out_{t}": rnn_out.numpy(), f"logits_{t}": logits.numpy(), f"action_{t}": act.numpy()})
        if t == 0:
            d["rgb_feats_0"] = feats["rgb"].numpy()
        rnn, prev = rnn_out, act
    np.savez_compressed(os.path.join(OUT, name), **d)
    print(name, "actions", [d[f"action_{t}"].ravel().tolist() for t in range(3)], "rgb feat", d["rgb_feats_0"].shape,
          "|features|", float(np.abs(d["features_2"]).max()))


if __name__ == "__main__":
    gen(False, "latent_act_plain.npz")
    gen(True, "latent_act_tourmem.npz")
