"""Pins oracle/rednet_ref.py to the golden produced by the reference's own RedNet."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
from det_init import det_fill  # noqa: E402

from oracle.rednet_ref import RedNetRef, predict_semantics_ref  # noqa: E402


def test_rednet_oracle_matches_reference_golden():
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "rednet.npz"))
    torch.set_num_threads(8)
    net = det_fill(RedNetRef(), seed=1, conv_gain=0.6).eval()
    scores, labels, rgb_n = predict_semantics_ref(net, torch.from_numpy(g["rgb"]), torch.from_numpy(g["depth"]))
    assert np.allclose(rgb_n.numpy(), g["rgb_n"], atol=1e-6)
    assert np.allclose(scores.numpy(), g["scores"], atol=1e-5)
    assert (labels.numpy() == g["labels"]).mean() == 1.0
