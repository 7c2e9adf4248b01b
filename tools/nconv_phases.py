"""Phase anatomy of k_nconv (ivln_nconv_f32) on the layer-1 shapes, like tools/gn_conv_phases.py: builds csrc/gn_conv.hip
with -DGN_CONV_TIMING and prints us per phase (load | table + transform | convs | stores + statistics), median over
workgroups, and the event time per launch.   python tools/nconv_phases.py [N=4]"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ivln_ce_amd  # noqa: E402,F401
from ivln_ce_amd import ops  # noqa: E402

so = "/tmp/libgnconv_timing.so"
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-DGN_CONV_TIMING",
                       os.path.join(ROOT, "ivln-ce_amd", "csrc", "gn_conv.hip"), "-o", so])
L = C.CDLL(so)
L.ivln_nconv_f32.argtypes = [C.POINTER(ops.NconvDesc), C.c_void_p]
L.ivln_gn_conv_stamps.argtypes = [C.c_void_p, C.c_int]
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device("cuda:0")
H = W = 32
RS = 2
strips = H // RS
# name, C, stats?, x2?, residual?, act?, (Cout_a, k), (Cout_b)
SHAPES = [("l1.0 conv1+ds", 32, 0, 0, 0, 0, (32, 1), 128), ("conv2 3x3", 32, 1, 0, 0, 0, (32, 3), 0),
          ("conv3 1x1", 32, 1, 0, 0, 0, (128, 1), 0), ("tail ds -> conv1", 128, 1, 1, 0, 1, (32, 1), 0),
          ("tail id -> conv1", 128, 1, 0, 1, 1, (32, 1), 0)]
print(f"N = {N}; us per phase, median over workgroups")
print(f"{'shape':18s} {'load':>6s} {'xform':>6s} {'conv':>6s} {'store':>6s} {'block':>6s} {'event':>7s}")
for name, Cc, st, x2, res, act, ca, cb in SHAPES:
    d = ops.NconvDesc()
    keep = []

    def t(*shape):
        a = torch.randn(*shape, device=dev)
        keep.append(a)
        return a

    d.x = t(Cc, N, H, W).data_ptr()
    if st:
        s_ = torch.rand(strips, N, 16, 3, device=dev) + 0.5
        s_[..., 0] = Cc // 16 * RS * W
        keep.append(s_)
        d.stats, d.parts, d.gamma, d.beta, d.groups, d.eps = s_.data_ptr(), strips, t(Cc).data_ptr(), t(Cc).data_ptr(), 16, 1e-5
    if x2:
        s2 = torch.rand(strips, N, 16, 3, device=dev) + 0.5
        s2[..., 0] = Cc // 16 * RS * W
        keep.append(s2)
        d.x2, d.stats2, d.parts2, d.gamma2, d.beta2 = t(Cc, N, H, W).data_ptr(), s2.data_ptr(), strips, t(Cc).data_ptr(), t(Cc).data_ptr()
    if res:
        d.residual = t(N, Cc, H, W).data_ptr()
    d.N, d.C, d.H, d.W, d.relu = N, Cc, H, W, 1
    if act:
        d.act_out = t(N, Cc, H, W).data_ptr()
    d.wa, d.Cout_a, d.ka, d.groups_a = t(ca[0], Cc, ca[1], ca[1]).data_ptr(), ca[0], ca[1], 16
    d.ya, d.stats_a = t(ca[0], N, H, W).data_ptr(), t(strips, N, 16, 3).data_ptr()
    if cb:
        d.wb, d.Cout_b, d.groups_b = t(cb, Cc, 1, 1).data_ptr(), cb, 16
        d.yb, d.stats_b = t(cb, N, H, W).data_ptr(), t(strips, N, 16, 3).data_ptr()
    d.rows_per_block = RS
    stream = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        assert L.ivln_nconv_f32(C.byref(d), stream) == 0, name
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        L.ivln_nconv_f32(C.byref(d), stream)
    e1.record()
    torch.cuda.synchronize()
    nb = strips * N
    stp = np.zeros(nb * 8, np.uint64)
    assert L.ivln_gn_conv_stamps(stp.ctypes.data, nb * 8) == 0
    tt = stp.reshape(nb, 8).astype(np.int64)[:, :5] / 100.0
    ph = np.median(np.diff(tt, axis=1), axis=0)
    print(f"{name:18s} {ph[0]:6.2f} {ph[1]:6.2f} {ph[2]:6.2f} {ph[3]:6.2f} {np.median(tt[:, 4] - tt[:, 0]):6.2f} {e0.elapsed_time(e1) * 1e3 / 20:7.2f}")
