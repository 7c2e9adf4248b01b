"""Generate golden vectors for the MapCMA policy by running the REFERENCE's own
`MapCMAPolicy` (ivlnce_baselines/models/map_cma_policy.py:28-100) on seeded inputs, with weights
from tests/golden/det_init.py.  Build container only:  python tests/golden/gen_policy_golden.py
Writes tests/golden/policy_act.npz (two consecutive act() steps with a mask reset, plus the
intermediate encoder outputs) and tests/golden/policy_update.npz (build_distribution on a T x N
trajectory batch, loss of base_il_trainer.py:201-211, and gradients)."""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _ref_shim  # noqa: E402

_ref_shim.install()
torch.set_num_threads(4)
from det_init import det_fill  # noqa: E402

from ivlnce_baselines.common.aux_losses import AuxLosses  # noqa: E402
from ivlnce_baselines.models.map_cma_policy import MapCMAPolicy  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def make_policy(use_pm):
    cfg = _ref_shim.default_model_config()
    cfg.MODEL.PROGRESS_MONITOR.use = use_pm
    sp = sys.modules["gym.spaces"]
    pol = MapCMAPolicy.from_config(cfg, _ref_shim.observation_space(), sp.Discrete(4))
    det_fill(pol, seed=0)
    return pol


def gen_act():
    B = 3
    g = torch.Generator().manual_seed(2024)
    pol = make_policy(False)
    pol.eval()
    d = {}
    feats = {}
    pol.net.depth_encoder.visual_encoder.register_forward_hook(lambda m, i, o: feats.__setitem__("depth", o.detach().clone()))
    pol.net.map_encoder.register_forward_hook(lambda m, i, o: feats.__setitem__("map", o.detach().clone()))
    pol.net.instruction_encoder.register_forward_hook(lambda m, i, o: feats.__setitem__("txt", o.detach().clone()))
    rnn = torch.zeros(B, 2, 512)
    prev = torch.zeros(B, 1, dtype=torch.long)
    lens = [80, 23, 200]
    instr = torch.zeros(B, 200, dtype=torch.long)
    for b, L in enumerate(lens):
        instr[b, :L] = torch.randint(2, 2504, (L,), generator=g)
    d["instruction"] = instr.numpy()
    for t in range(2):
        col = torch.rand(B, 1, 256, 1, generator=g)
        depth = (0.2 + 0.6 * col + 0.02 * torch.rand(B, 256, 256, 1, generator=g)).clamp(0, 1)
        occ = (torch.rand(B, 64, 64, generator=g) < 0.3).to(torch.uint8)
        sem = (torch.randint(0, 13, (B, 64, 64), generator=g) * occ).to(torch.uint8)
        masks = torch.ones(B, 1, dtype=torch.uint8)
        if t == 0:
            masks[:] = 0
        else:
            masks[1] = 0  # env 1 starts a new episode at step 1
        obs = {"depth": depth, "occupancy_map": occ, "semantic_map": sem, "instruction": instr}
        with torch.no_grad():
            f, rnn_out = pol.net(obs, rnn, prev, masks)
            logits = pol.action_distribution(f).logits
            action = logits.argmax(-1, keepdim=True)
        d[f"depth_{t}"] = depth.numpy().astype(np.float32)
        d[f"occ_{t}"] = occ.numpy()
        d[f"sem_{t}"] = sem.numpy()
        d[f"masks_{t}"] = masks.numpy()
        d[f"prev_{t}"] = prev.numpy()
        d[f"rnn_in_{t}"] = rnn.numpy().copy()
        d[f"rnn_out_{t}"] = rnn_out.numpy().copy()
        d[f"features_{t}"] = f.numpy()
        d[f"logits_{t}"] = logits.numpy()
        d[f"depth_feat_{t}"] = feats["depth"].numpy()
        d[f"map_feat_{t}"] = feats["map"].numpy()
        if t == 0:
            d["txt_feat"] = feats["txt"].numpy()
        rnn, prev = rnn_out, action
    np.savez_compressed(os.path.join(OUT, "policy_act.npz"), **d)
    print("act logits", d["logits_1"])


def gen_update():
    T, N = 6, 5
    g = torch.Generator().manual_seed(77)
    pol = make_policy(True)
    pol.train()  # base trainer: policy in train mode during updates (BN batch stats, quirk Q6)
    TN = T * N
    lens = [40, 80, 12, 200, 66]
    instr1 = torch.zeros(N, 200, dtype=torch.long)
    for b, L in enumerate(lens):
        instr1[b, :L] = torch.randint(2, 2504, (L,), generator=g)
    instr = instr1.unsqueeze(0).expand(T, N, 200).reshape(TN, 200).float()  # batch_to casts obs to f32
    occ = (torch.rand(TN, 64, 64, generator=g) < 0.3).float()
    sem = (torch.randint(0, 13, (TN, 64, 64), generator=g).float() * occ)
    depth_features = torch.randn(TN, 128, 4, 4, generator=g)
    progress = torch.rand(TN, 1, generator=g)
    prev = torch.randint(0, 4, (TN, 1), generator=g)
    tgt = torch.randint(0, 4, (T, N), generator=g)
    w = torch.where(torch.rand(T, N, generator=g) < 0.3, torch.tensor(3.2), torch.tensor(1.0))
    w[4:, 2] = 0.0  # a padded (shorter) trajectory
    nd = torch.ones(T, N, dtype=torch.uint8)
    nd[0] = 0
    nd = nd.view(-1, 1)
    obs = {"depth_features": depth_features, "occupancy_map": occ, "semantic_map": sem, "instruction": instr,
           "progress": progress}
    AuxLosses.activate()
    AuxLosses.clear()
    h0 = torch.zeros(N, 2, 512)
    dist, _ = pol.build_distribution(obs, h0, prev, nd)
    logits = dist.logits.view(T, N, -1)
    ce = F.cross_entropy(logits.permute(0, 2, 1), tgt, reduction="none")
    action_loss = ((w * ce).sum(0) / w.sum(0)).mean()
    aux = AuxLosses.reduce((w > 0).view(-1))
    loss = action_loss + aux
    loss.backward()
    d = dict(T=T, N=N, instruction=instr.numpy(), occ=occ.numpy(), sem=sem.numpy(),
             depth_features=depth_features.numpy(), progress=progress.numpy(), prev=prev.numpy(),
             targets=tgt.numpy(), weights=w.numpy(), not_done=nd.numpy(),
             logits=logits.detach().numpy(), loss=float(loss), action_loss=float(action_loss), aux_loss=float(aux))
    for k, p in pol.named_parameters():
        if p.grad is not None:
            d["gradnorm/" + k] = float(p.grad.norm())
    for k in ["action_distribution.linear.weight", "net.state_q.weight", "net.map_encoder.cnn.0.conv.0.bias",
              "net.map_encoder.cnn.3.conv.1.weight", "net.prev_action_embedding.weight", "net.text_q.bias",
              "net.state_encoder.rnn.bias_hh_l0", "net.instruction_encoder.encoder_rnn.bias_ih_l0_reverse",
              "net.depth_encoder.spatial_embeddings.weight", "net.progress_monitor.weight"]:
        d["grad/" + k] = dict(pol.named_parameters())[k].grad.numpy()
    sd = pol.state_dict()
    for k in ["net.map_encoder.cnn.0.conv.1.running_mean", "net.map_encoder.cnn.3.conv.1.running_var"]:
        d["post/" + k] = sd[k].numpy()
    np.savez_compressed(os.path.join(OUT, "policy_update.npz"), **d)
    print("update loss", float(loss), float(action_loss), float(aux))


if __name__ == "__main__":
    gen_act()
    gen_update()
    for f in ["policy_act.npz", "policy_update.npz"]:
        print(f, os.path.getsize(os.path.join(OUT, f)) // 1024, "KiB")
