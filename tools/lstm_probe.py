"""Time k_lstm_bidir for different sequence lengths (fixed part vs per-timestep part)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import ivln_ce_amd  # noqa: E402,F401
from ivln_ce_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
B, L, H = 4, 200, 128
gx_f = torch.randn(B * L, 4 * H, device=dev) * 0.1
gx_r = torch.randn(B * L, 4 * H, device=dev) * 0.1
whh_f, whh_r = torch.randn(4 * H, H, device=dev) * 0.05, torch.randn(4 * H, H, device=dev) * 0.05
bf, br = torch.zeros(4 * H, device=dev), torch.zeros(4 * H, device=dev)
for ln in (1, 10, 40, 80, 160, 200):
    lengths = torch.full((B,), ln, dtype=torch.int32, device=dev)
    for _ in range(5):
        ops.lstm_bidir(gx_f, gx_r, whh_f, whh_r, bf, br, lengths, B, L, H)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(50):
        ops.lstm_bidir(gx_f, gx_r, whh_f, whh_r, bf, br, lengths, B, L, H)
    b.record()
    torch.cuda.synchronize()
    print(f"len {ln:4d}: {a.elapsed_time(b) / 50 * 1e3:8.1f} us per call (incl. ~launch)")
