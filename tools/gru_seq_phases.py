"""Where does a timestep of the persistent sequence GRU go?  Builds csrc/gru_seq.hip with -DGRU_SEQ_TIMING into a scratch
library (per-workgroup wall-clock stamps at the phase boundaries of the forward kernel), runs it at T = 64 and prints,
per N: staging (sc1 loads of h_{t-1} -> LDS), matvec + reduction, element part + stores issued, store drain + arrival,
deferred stores + counter wait - medians over workgroups and steps.   usage: python tools/gru_seq_phases.py"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ivln_ce_amd  # noqa: E402,F401
from test_gpu_kernels import _gru_seq_case  # noqa: E402

so = "/tmp/libgruseq_timing.so"
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC",
                       "-DGRU_SEQ_TIMING", os.path.join(ROOT, "ivln-ce_amd", "csrc", "gru_seq.hip"), "-o", so])
L = C.CDLL(so)
vp, i64, i32 = C.c_void_p, C.c_int64, C.c_int
L.ivln_gru_seq_fwd_persistent.argtypes = [vp, vp, i64, vp, vp, vp, vp, i64, vp, i64, i32, i32, vp, vp, vp, vp, vp, vp]
L.ivln_gru_seq_stamps.argtypes = [vp, i32]
DEV = "cuda:0"
NWG = 32 if os.environ.get("(removed switch) IVLN_SEQ_UPB") == "16" else 64
T = 64
print(f"T = {T}; us per phase, median over {NWG} workgroups x {T - 2} steps (100 MHz stamps)")
print(f"{'N':>3s} {'stage':>7s} {'matvec':>7s} {'element':>8s} {'drain':>7s} {'exchange':>9s} {'step':>7s}")
for N in (1, 4, 8, 16, 32):
    H, gi, h0, masks, w_hh, b_hh, d_out = _gru_seq_case(T, N, seed=1)
    out = torch.empty((T * N, H), device=DEV)
    state = torch.empty((N, H), device=DEV)
    saves = [torch.empty((T * N, H), device=DEV) for _ in range(4)]
    ws = torch.zeros(64, dtype=torch.int32, device=DEV)
    for _ in range(3):
        rc = L.ivln_gru_seq_fwd_persistent(gi.data_ptr(), h0.data_ptr(), h0.stride(0), masks.data_ptr(), w_hh.data_ptr(),
                                           b_hh.data_ptr(), out.data_ptr(), out.stride(0), state.data_ptr(), state.stride(0),
                                           T, N, *[s.data_ptr() for s in saves], ws.data_ptr(), None)
        assert rc == 0
        torch.cuda.synchronize()
    st = np.zeros(64 * 256 * 8, np.uint64)
    assert L.ivln_gru_seq_stamps(st.ctypes.data_as(vp), st.nbytes) == 0
    st = st.reshape(64, 256, 8)[:NWG, 1:T - 1].astype(np.int64)  # skip the first / last step
    d = np.diff(st[..., :6], axis=-1) / 100.0
    step = (st[:, 1:, 0] - st[:, :-1, 0]) / 100.0
    med = np.median(d.reshape(-1, 5), axis=0)
    print(f"{N:3d} {med[0]:7.2f} {med[1]:7.2f} {med[2]:8.2f} {med[3]:7.2f} {med[4]:9.2f} {np.median(step):7.2f}")
