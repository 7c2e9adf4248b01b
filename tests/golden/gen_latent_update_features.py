"""Seeded feature tensors shared by gen_latent_update_golden.py and tests/test_gpu_latent.py (kept out of the
fixtures: 6 MB of incompressible noise per case)."""
import numpy as np


def features(seed, T, N):
    """(rgb_features (TN,2048,4,4) >= 0 like post-ReLU ResNet activations, depth_features (TN,128,4,4))."""
    rs = np.random.RandomState(seed)
    rgb = np.abs(rs.standard_normal((T * N, 2048, 4, 4))).astype(np.float32) * 0.5
    dep = rs.standard_normal((T * N, 128, 4, 4)).astype(np.float32)
    return rgb, dep
