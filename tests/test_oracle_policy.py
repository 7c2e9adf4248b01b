"""Pins the torch-CPU policy oracle (oracle/policy_ref.py + oracle/habitat_ext_ref.py) to goldens
produced by the reference's own MapCMAPolicy (tests/golden/gen_policy_golden.py)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
from det_init import det_fill  # noqa: E402

from oracle.policy_ref import MapCMAPolicyRef  # noqa: E402

G = os.path.join(os.path.dirname(__file__), "golden")


def test_state_dict_keys_match_reference_contract():
    pol = MapCMAPolicyRef()
    keys = set(pol.state_dict().keys())
    for k in [
        "net.map_encoder.cnn.0.conv.0.weight", "net.map_encoder.cnn.3.conv.1.running_var",
        "net.instruction_encoder.encoder_rnn.weight_hh_l0_reverse", "net.instruction_encoder.embedding_layer.weight",
        "net.depth_encoder.visual_encoder.backbone.conv1.0.weight",
        "net.depth_encoder.visual_encoder.backbone.layer3.5.convs.7.bias",
        "net.depth_encoder.visual_encoder.backbone.layer2.0.downsample.1.weight",
        "net.depth_encoder.visual_encoder.compression.1.weight", "net.depth_encoder.spatial_embeddings.weight",
        "net.prev_action_embedding.weight", "net.depth_linear.1.weight", "net.map_linear.1.bias",
        "net.state_encoder.rnn.weight_ih_l0", "net.dep_kv.weight", "net.map_kv.bias", "net.state_q.weight",
        "net.text_k.weight", "net.text_q.bias", "net._scale", "net.second_state_compress.0.weight",
        "net.second_state_encoder.rnn.bias_hh_l0", "net.progress_monitor.weight", "action_distribution.linear.bias",
    ]:
        assert k in keys, k
    assert pol.state_dict()["net.state_encoder.rnn.weight_ih_l0"].shape == (1536, 416)
    n = sum(p.numel() for p in pol.parameters())
    assert n == 13642357  # SURVEY.md section 8 [probe]


def test_act_matches_reference_golden():
    g = np.load(os.path.join(G, "policy_act.npz"))
    torch.set_num_threads(4)
    pol = det_fill(MapCMAPolicyRef(), seed=0).eval()
    instr = torch.from_numpy(g["instruction"])
    for t in range(2):
        obs = {
            "depth": torch.from_numpy(g[f"depth_{t}"]), "occupancy_map": torch.from_numpy(g[f"occ_{t}"]),
            "semantic_map": torch.from_numpy(g[f"sem_{t}"]), "instruction": instr,
        }
        with torch.no_grad():
            logits, states, feats = pol.logits(
                obs, torch.from_numpy(g[f"rnn_in_{t}"]), torch.from_numpy(g[f"prev_{t}"]), torch.from_numpy(g[f"masks_{t}"])
            )
        assert np.allclose(feats.numpy(), g[f"features_{t}"], atol=2e-5, rtol=1e-4)
        assert np.allclose(states.numpy(), g[f"rnn_out_{t}"], atol=2e-5, rtol=1e-4)
        assert np.allclose(torch.log_softmax(logits, -1).numpy(), g[f"logits_{t}"], atol=1e-5)


def test_update_loss_and_grads_match_reference_golden():
    g = np.load(os.path.join(G, "policy_update.npz"))
    torch.set_num_threads(4)
    pol = det_fill(MapCMAPolicyRef(use_pm=True), seed=0).train()
    obs = {k: torch.from_numpy(g[k2]) for k, k2 in [
        ("depth_features", "depth_features"), ("occupancy_map", "occ"), ("semantic_map", "sem"),
        ("instruction", "instruction"), ("progress", "progress")]}
    loss, action_loss, aux, logits = pol.update_loss(
        obs, torch.from_numpy(g["prev"]), torch.from_numpy(g["not_done"]), torch.from_numpy(g["targets"]),
        torch.from_numpy(g["weights"]),
    )
    loss.backward()
    assert abs(float(loss) - float(g["loss"])) < 1e-5
    assert abs(float(action_loss) - float(g["action_loss"])) < 1e-5
    assert abs(float(aux) - float(g["aux_loss"])) < 1e-5
    params = dict(pol.named_parameters())
    for k in g.files:
        if k.startswith("gradnorm/"):
            ref = float(g[k])
            got = float(params[k[9:]].grad.norm())
            assert abs(got - ref) <= 1e-4 * max(1.0, abs(ref)), k
        elif k.startswith("grad/"):
            assert np.allclose(params[k[5:]].grad.numpy(), g[k], atol=1e-6, rtol=1e-3), k
        elif k.startswith("post/"):
            assert np.allclose(pol.state_dict()[k[5:]].numpy(), g[k], atol=1e-6, rtol=1e-5), k
