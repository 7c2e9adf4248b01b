"""Times the sequence GRU (ivln_cma_seq_fwd/bwd) as one persistent launch against the launch-per-timestep path:
HIP events around 20 repetitions each, at the update step's shape (T = 64, N = 8, H = 512) and a few others."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import ivln_ce_amd  # noqa: F401,E402
from ivln_ce_amd import ops  # noqa: E402
from test_gpu_kernels import _gru_seq_case  # noqa: E402

DEV = "cuda:0"


def time_it(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1000.0


for T, N in [(64, 8), (64, 5), (64, 16), (16, 8), (200, 8)]:
    H, gi, h0, masks, w_hh, b_hh, d_out = _gru_seq_case(T, N, seed=1)
    out = torch.empty((T * N, H), device=DEV)
    state = torch.empty((N, H), device=DEV)
    saves = tuple(torch.empty((T * N, H), device=DEV) for _ in range(4))
    whh_t = w_hh.t().contiguous()
    dgi = torch.empty((T * N, 3 * H), device=DEV)
    dgh = torch.empty((T * N, 3 * H), device=DEV)
    hp = torch.empty((T * N, H), device=DEV)
    dhz = torch.empty((N, H), device=DEV)
    row = {}
    for persistent in (False, True):
        ops.SEQ_PERSISTENT = persistent
        f = time_it(lambda: ops.gru_seq(gi, h0, masks, w_hh, b_hh, out, state, T, N, saves))
        b = time_it(lambda: ops.gru_seq_bwd(d_out, *saves, out, h0, masks, whh_t, T, N, dgi, dgh, hp, dhz))
        row[persistent] = (f, b)
    ops.check_seq_sync()
    print(f"T={T:3d} N={N:2d}  fwd: launches {row[False][0]:7.1f} us ({row[False][0] / T:5.2f}/step)  persistent "
          f"{row[True][0]:7.1f} us ({row[True][0] / T:5.2f}/step)   bwd: launches {row[False][1]:7.1f} us "
          f"({row[False][1] / T:5.2f}/step)  persistent {row[True][1]:7.1f} us ({row[True][1] / T:5.2f}/step)", flush=True)
