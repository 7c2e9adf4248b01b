"""Golden vectors for TRAINING the Latent-CMA policy: the REFERENCE's own
`LatentCMAPolicy.build_distribution` (ivlnce_baselines/models/latent_cma_policy.py:124-179, sequence mode and
the unrolled tour-memory mode) + the loss of `IterativeDaggerTrainer._update_agent`
(trainers/iterative_dagger_trainer.py:33-94) + autograd, on seeded inputs with det_init weights.
Three memory configurations, as the reference's latent_baselines configs use them:
  plain   - episodic memory, progress monitor on, h0 = 0                 (1_cma)
  tour    - `tour_memory`: GRUs reset with the TOUR mask, carried h0     (2_tour_cma)
  variant - `tour_memory_variant` + `memory_at_end`: unrolled, third tour-long memory slot (4_pool_end_cma)
The big feature tensors are regenerated in the test from numpy RandomState seeds (stable across numpy versions).
Build container only:  python tests/golden/gen_latent_update_golden.py -> tests/golden/latent_update_{plain,tour,variant}.npz
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _ref_shim  # noqa: E402

_ref_shim.install()
torch.set_num_threads(8)
from det_init import det_fill  # noqa: E402
from gen_latent_update_features import features  # noqa: E402

from ivlnce_baselines.common.aux_losses import AuxLosses  # noqa: E402
from ivlnce_baselines.models.latent_cma_policy import LatentCMAPolicy  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
T, N = 4, 3


def make_policy(mode):
    cfg = _ref_shim.default_model_config()
    cfg.MODEL.policy_name = "LatentCMAPolicy"
    cfg.MODEL.tour_memory = mode == "tour"
    cfg.MODEL.tour_memory_variant = mode == "variant"
    cfg.MODEL.memory_at_end = mode == "variant"
    cfg.MODEL.PROGRESS_MONITOR.use = mode == "plain"
    sp = sys.modules["gym.spaces"]
    space = sp.Dict({"depth": sp.Box(0.0, 1.0, (256, 256, 1), np.float32), "rgb": sp.Box(0, 255, (224, 224, 3), np.uint8),
                     "instruction": sp.Box(0, 2504, (200,), np.int64)})
    pol = LatentCMAPolicy.from_config(cfg, space, sp.Discrete(4))
    det_fill(pol, seed=0, conv_gain=1.0)
    return pol.train()


def gen(mode, seed):
    g = torch.Generator().manual_seed(seed)
    pol = make_policy(mode)
    L = pol.net.num_recurrent_layers
    TN = T * N
    rgb_np, dep_np = features(seed, T, N)
    instr1 = torch.zeros(N, 200, dtype=torch.long)
    for b, n in enumerate([57, 200, 9]):
        instr1[b, :n] = torch.randint(2, 2504, (n,), generator=g)
    instr = instr1.unsqueeze(0).expand(T, N, 200).reshape(TN, 200).float()  # batch_to casts observations to f32
    progress = torch.rand(TN, 1, generator=g)
    prev = torch.randint(0, 4, (TN, 1), generator=g)
    tgt = torch.randint(0, 4, (T, N), generator=g)
    w = torch.where(torch.rand(T, N, generator=g) < 0.3, torch.tensor(3.2), torch.tensor(1.0))
    w[3:, 1] = 0.0  # a padded (shorter) trajectory
    ep = torch.ones(T, N, dtype=torch.uint8)
    ep[0] = 0  # every batch row starts an episode (tour_dataset.py:90-93)
    tour = torch.ones(T, N, dtype=torch.uint8)
    tour[0, 0] = 0  # only trajectory 0 starts a new tour; the others continue theirs (tour_dataset.py:280-281)
    ep, tour = ep.view(-1, 1), tour.view(-1, 1)
    if mode == "plain":
        h0 = torch.zeros(N, L, 512)
    else:
        h0 = 0.3 * torch.randn(N, L, 512, generator=g)  # state carried over from the previous batch of the tour
        if mode == "variant":
            h0[:, : L - 1] = 0  # iterative_dagger_trainer.py:58-60: episodic slots cleared, tour slot kept
    obs = {"rgb_features": torch.from_numpy(rgb_np), "depth_features": torch.from_numpy(dep_np), "instruction": instr,
           "progress": progress}
    AuxLosses.clear()
    if mode == "plain":
        AuxLosses.activate()
    else:
        AuxLosses.deactivate()
    dist, rnn_out = pol.build_distribution(obs, h0.clone().detach(), prev, ep, tour)
    logits = dist.logits.view(T, N, -1)
    ce = F.cross_entropy(logits.permute(0, 2, 1), tgt, reduction="none")
    action_loss = ((w * ce).sum(0) / w.sum(0)).mean()
    aux = AuxLosses.reduce((w > 0).view(-1)) if mode == "plain" else 0.0
    loss = action_loss + aux
    loss.backward()
    AuxLosses.deactivate()
    d = dict(T=T, N=N, seed=seed, instruction=instr.numpy(), progress=progress.numpy(), prev=prev.numpy(),
             targets=tgt.numpy(), weights=w.numpy(), ep=ep.numpy(), tour=tour.numpy(), h0=h0.numpy(),
             logits=logits.detach().numpy(), rnn_out=rnn_out.detach().numpy(), loss=float(loss),
             action_loss=float(action_loss), aux_loss=float(aux))
    named = dict(pol.named_parameters())
    n_grad = 0
    for k, p in named.items():
        if p.grad is not None:
            d["gradnorm/" + k] = float(p.grad.norm())
            n_grad += 1
    full = ["action_distribution.linear.weight", "net.state_q.weight", "net.rgb_kv.bias", "net.depth_kv.weight",
            "net.rgb_linear.2.bias", "net.prev_action_embedding.weight", "net.text_q.bias",
            "net.state_encoder.rnn.bias_hh_l0", "net.second_state_encoder.rnn.bias_ih_l0",
            "net.instruction_encoder.encoder_rnn.bias_ih_l0_reverse", "net.rgb_encoder.spatial_embeddings.weight",
            "net.depth_encoder.spatial_embeddings.weight"]
    if mode == "variant":
        full.append("net.out_layer.0.bias")
    if mode == "plain":
        full.append("net.progress_monitor.weight")
    for k in full:
        d["grad/" + k] = named[k].grad.numpy()
    name = f"latent_update_{mode}.npz"
    np.savez_compressed(os.path.join(OUT, name), **d)
    print(name, "loss", float(loss), float(action_loss), float(aux), "params with grad", n_grad,
          os.path.getsize(os.path.join(OUT, name)) // 1024, "KiB")


if __name__ == "__main__":
    gen("plain", 101)
    gen("tour", 102)
    gen("variant", 103)
