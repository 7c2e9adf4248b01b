"""Where does a k_gn_conv launch spend its time?  Builds csrc/gn_conv.hip with -DGN_CONV_TIMING into a scratch library
(per-block wall-clock stamps at the phase boundaries), runs the depth ResNet's layer shapes at N envs and prints, per
shape: load (slabs + weights), GroupNorm (stats, normalise, pool), activation store, conv A, conv B - medians over
blocks, plus first-start -> last-end of the whole grid.   usage: python tools/gn_conv_phases.py [N=4]"""
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
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-DGN_CONV_TIMING",
                       os.path.join(ROOT, "ivln-ce_amd", "csrc", "gn_conv.hip"), "-o", so])
L = C.CDLL(so)
L.ivln_gn_conv_f32.argtypes = [C.POINTER(ops.GnConvDesc), C.c_void_p]
L.ivln_gn_conv_stamps.argtypes = [C.c_void_p, C.c_int]
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device("cuda:0")
# name, C, H, W, splits, x2, residual, pool, (Cout_a, k, s, p), (Cout_b, s)
SHAPES = [
    ("stem", 32, 64, 64, 1, 0, 0, 1, (32, 1, 1, 0), (128, 1)),
    ("l1 gn1", 32, 32, 32, 16, 0, 0, 0, (32, 3, 1, 1), None),
    ("l1 gn2", 32, 32, 32, 16, 0, 0, 0, (128, 1, 1, 0), None),
    ("l1 tail", 128, 32, 32, 16, 0, 1, 0, (32, 1, 1, 0), None),
    ("l1 last", 128, 32, 32, 16, 0, 1, 0, (64, 1, 1, 0), (256, 2)),
    ("l2 gn1 s2", 64, 32, 32, 16, 0, 0, 0, (64, 3, 2, 1), None),
    ("l2 gn2", 64, 16, 16, 16, 0, 0, 0, (256, 1, 1, 0), None),
    ("l2 tail", 256, 16, 16, 16, 0, 1, 0, (64, 1, 1, 0), None),
    ("l2 last", 256, 16, 16, 16, 0, 1, 0, (128, 1, 1, 0), (512, 2)),
    ("l3 gn1", 128, 8, 8, 16, 0, 0, 0, (128, 3, 1, 1), None),
    ("l3 gn2", 128, 8, 8, 16, 0, 0, 0, (512, 1, 1, 0), None),
    ("l3 tail", 512, 8, 8, 16, 0, 1, 0, (128, 1, 1, 0), None),
    ("l3 last", 512, 8, 8, 16, 0, 1, 0, (256, 1, 1, 0), (1024, 2)),
    ("l4 gn1", 256, 4, 4, 16, 0, 0, 0, (256, 3, 1, 1), None),
    ("l4 gn2", 256, 4, 4, 16, 0, 0, 0, (1024, 1, 1, 0), None),
    ("l4 tail ds", 1024, 4, 4, 16, 1, 0, 0, (256, 1, 1, 0), None),
    ("l4 last", 1024, 4, 4, 16, 0, 1, 0, (128, 3, 1, 1), None),
    # two conv layers per launch: splits < 0 = channels of the front stage (GN2 -> conv3 -> this GroupNorm)
    ("l2 pair", 256, 16, 16, -64, 0, 1, 0, (64, 1, 1, 0), None),
    ("l3 pair", 512, 8, 8, -128, 0, 1, 0, (128, 1, 1, 0), None),
    ("l4 pair", 1024, 4, 4, -256, 0, 1, 0, (256, 1, 1, 0), None),
    ("l4 pair last", 1024, 4, 4, -256, 0, 1, 0, (128, 3, 1, 1), None),
]
print(f"N = {N}; us per phase, median over blocks")
print(f"{'shape':12s} {'grid':>9s} {'load':>6s} {'gn':>6s} {'act':>6s} {'convA':>6s} {'convB':>6s} {'mfma':>6s} {'front':>6s} {'block':>6s} {'grid span':>9s} {'event':>7s}")
for name, Cc, H, W, splits, x2, res, pool, ca, cb in SHAPES:
    g = torch.Generator(device="cpu").manual_seed(1)
    d = ops.GnConvDesc()
    gam, bet = torch.randn(Cc, device=dev), torch.randn(Cc, device=dev)
    keep = [gam, bet]
    d.gamma, d.beta = gam.data_ptr(), bet.data_ptr()
    if splits > 0:
        x = torch.randn(splits, Cc, N * H * W, device=dev)
        d.x, d.splits, d.slab_stride = x.data_ptr(), splits, Cc * N * H * W
    else:
        C0 = -splits
        x = torch.randn(16, C0, N * H * W, device=dev)
        g0, b0, w0 = torch.randn(C0, device=dev), torch.randn(C0, device=dev), torch.randn(Cc, C0, 1, 1, device=dev) / C0 ** 0.5
        keep += [g0, b0, w0]
        d.x0, d.splits0, d.slab_stride0, d.C0, d.groups0 = x.data_ptr(), 16, C0 * N * H * W, C0, 16
        d.gamma0, d.beta0, d.w0 = g0.data_ptr(), b0.data_ptr(), w0.data_ptr()
    keep.append(x)
    if x2:
        t = torch.randn(16, Cc, N * H * W, device=dev)
        keep.append(t)
        d.x2, d.splits2, d.slab_stride2, d.gamma2, d.beta2 = t.data_ptr(), 16, Cc * N * H * W, gam.data_ptr(), bet.data_ptr()
    if res:
        t = torch.randn(N, Cc, H, W, device=dev)
        keep.append(t)
        d.residual = t.data_ptr()
    d.N, d.C, d.H, d.W, d.groups, d.eps, d.relu, d.pool = N, Cc, H, W, 16, 1e-5, 1, pool
    Hp, Wp = ((H - 1) // 2 + 1, (W - 1) // 2 + 1) if pool else (H, W)
    act = torch.empty(N, Cc, Hp, Wp, device=dev)
    d.act_out = act.data_ptr()
    Co, k, s_, p_ = ca
    wa = torch.randn(Co, Cc, k, k, device=dev)
    Ho, Wo = (Hp + 2 * p_ - k) // s_ + 1, (Wp + 2 * p_ - k) // s_ + 1
    ya = torch.empty(16 * Co * N * Ho * Wo, device=dev)
    d.wa, d.Cout_a, d.ka, d.stride_a, d.pad_a, d.ya = wa.data_ptr(), Co, k, s_, p_, ya.data_ptr()
    if cb:
        wb = torch.randn(cb[0], Cc, 1, 1, device=dev)
        Hb, Wb = (Hp - 1) // cb[1] + 1, (Wp - 1) // cb[1] + 1
        yb = torch.empty(16 * cb[0] * N * Hb * Wb, device=dev)
        d.wb, d.Cout_b, d.stride_b, d.yb = wb.data_ptr(), cb[0], cb[1], yb.data_ptr()
    stream = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        assert L.ivln_gn_conv_f32(C.byref(d), stream) == 0, name
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        L.ivln_gn_conv_f32(C.byref(d), stream)
    e1.record()
    torch.cuda.synchronize()
    ev = e0.elapsed_time(e1) * 1e3 / 20
    nb = N * 16 * 8
    st = np.zeros(nb * 8, np.uint64)
    assert L.ivln_gn_conv_stamps(st.ctypes.data, nb * 8) == 0
    st = st.reshape(nb, 8).astype(np.int64)
    used = st[:, 5] > 0
    # the grid: blocks whose stamps belong to the last launch (S unknown here: take those with the newest start)
    newest = st[:, 0].max()
    live = used & (st[:, 0] > newest - 100_000)
    t = st[live][:, :6] / 100.0  # 100 MHz -> us
    front = 0.0
    if splits < 0:  # stamps: 0 start, 1 loads done, 7 front stage done, 2 GroupNorm done, ...
        t7 = st[live][:, 7] / 100.0
        front = float(np.median(t7 - t[:, 1]))
        t = t.copy()
        t[:, 2:] -= (t7 - t[:, 1])[:, None]  # report the GroupNorm phase from the end of the front stage
    ph = np.median(np.diff(t, axis=1), axis=0)
    print(f"{name:12s} {int(live.sum()):9d} {ph[0]:6.2f} {ph[1]:6.2f} {ph[2]:6.2f} {ph[3]:6.2f} {ph[4]:6.2f} {np.median(st[live][:, 6] / 100.0 - t[:, 3]):6.2f} {front:6.2f} {np.median(t[:, 5] - t[:, 0]):6.2f} "
          f"{t[:, 5].max() - t[:, 0].min():9.2f} {ev:7.2f}")
