"""Drop-in loaders, GPU part: after loading the files of tests/loader_files.py through this package's loaders, one
forward on the HIP kernels equals what the REFERENCE computed after loading byte-identical files through its own
loaders (tests/golden/gen_loader_golden.py -> loader_golden.npz).  Tolerances are the per-component bars of the
other parity tests (depth encoder 2e-4, RedNet scores 3e-4, instruction / map encoders 2e-4)."""
import json
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import loader_files as LF  # noqa: E402
from test_loaders import MAN, _policy  # noqa: E402

DEV = "cuda:0"
G = np.load(os.path.join(ROOT, "tests", "golden", "loader_golden.npz"))


def _close(got, ref, atol):
    got, ref = got.detach().cpu().float(), torch.as_tensor(ref).float()
    err = (got - ref).abs().max().item()
    assert err <= atol, f"max err {err:.3e} (bar {atol:.1e}), ref max {ref.abs().max().item():.3e}"


def test_depth_encoder_from_ddppo_checkpoint_matches_reference_forward(tmp_path):
    path = str(tmp_path / "ddppo.pth")
    LF.write_ddppo_checkpoint(path, MAN["ddppo"])
    pol = _policy(["MODEL.DEPTH_ENCODER.ddppo_checkpoint", path]).to(DEV).eval()
    depth = torch.rand(2, 256, 256, 1, generator=torch.Generator().manual_seed(78)).to(DEV)
    with torch.no_grad():
        feats = pol.net.depth_encoder.visual_encoder({"depth": depth})
    _close(feats, G["ddppo_features"], 2e-4)


def test_rednet_from_module_prefixed_pickle_matches_reference_forward(tmp_path):
    from ivln_ce_amd.rednet import PredictSemantics

    path = str(tmp_path / "rednet_mp3d_best_model.pkl")
    LF.write_rednet_pickle(path, MAN["rednet"])
    ps = PredictSemantics(torch.device(DEV))
    ps.CFG = dict(PredictSemantics.CFG, load_model=path)
    ps.setup()
    with torch.no_grad():
        scores = ps.model(torch.from_numpy(G["rednet_rgb"]).to(DEV), torch.from_numpy(G["rednet_depth"]).to(DEV))
    _close(scores, G["rednet_scores"], 3e-4)
    assert (scores.argmax(1).cpu() == torch.from_numpy(G["rednet_scores"]).argmax(1)).float().mean() >= 0.999


@pytest.mark.parametrize("tune", [False, True])
def test_instruction_encoder_with_pretrained_embeddings_matches_reference_forward(tmp_path, tune):
    emb = str(tmp_path / "embeddings.json.gz")
    LF.write_embeddings_file(emb, 2504, 50)
    pol = _policy(["MODEL.INSTRUCTION_ENCODER.use_pretrained_embeddings", True, "MODEL.INSTRUCTION_ENCODER.embedding_file",
                   emb, "MODEL.INSTRUCTION_ENCODER.fine_tune_embeddings", tune])
    ie = pol.net.instruction_encoder
    sd = ie.state_dict()
    from det_init import det_value

    for k, v in sd.items():
        if not k.startswith("embedding_layer"):
            sd[k] = det_value("net.instruction_encoder." + k, v)
    ie.load_state_dict(sd)
    ie.to(DEV)
    tokens = torch.from_numpy(G["emb_tokens"]).to(DEV)
    with torch.no_grad():
        out = ie({"instruction": tokens})
    out = out[0] if isinstance(out, tuple) else out
    ref = G[f"emb_out_tune{int(tune)}"]           # (B, 256, Lmax = 80): the reference pads to the batch's longest
    _close(out[:, :, : ref.shape[2]], ref, 2e-4)
    assert float(out[:, :, ref.shape[2]:].abs().max()) == 0.0 if out.shape[2] > ref.shape[2] else True


def test_map_encoder_from_pretrained_checkpoint_matches_reference_forward(tmp_path):
    mp = str(tmp_path / "map_encoder.pth")
    LF.write_map_encoder_checkpoint(mp, MAN["map_ckpt"])
    pol = _policy(["MODEL.SEMANTIC_MAP_ENCODER.from_pretrained", True, "MODEL.SEMANTIC_MAP_ENCODER.checkpoint", mp,
                   "MODEL.SEMANTIC_MAP_ENCODER.trainable", False]).to(DEV)
    enc = pol.net.map_encoder
    assert not enc.training   # frozen encoders run in eval mode (map_encoder.py:72-76): running statistics
    obs = {"occupancy_map": torch.from_numpy(G["map_occ"]).to(DEV), "semantic_map": torch.from_numpy(G["map_sem"]).to(DEV)}
    with torch.no_grad():
        out = enc(obs)
    out = out[0] if isinstance(out, tuple) else out
    _close(out, G["map_features"], 2e-4)
