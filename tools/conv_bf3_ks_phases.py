"""Where do the wave-split split-bf16 kernels (k_conv1x1_bf3_ks in both forms, k_conv_bf3_ks) spend their time?  Links a scratch copy of the library
with csrc/conv_bf3.hip built -DBF3_TIMING: every WAVE leaves (prologue round trip, K loop, epilogue, start time) on the 100 MHz
wall clock and, for the 1x1 kernels, the K loop's shader cycles by section (waits + split, load issue, MFMA issue).  Printed per shape: the launch's event time, the medians of the three phases over the waves, and the spread of the
waves' start and end times (a launch is as long as its last wave).
usage: python tools/conv_bf3_ks_phases.py"""
import ctypes as C
import glob
import os
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CS = os.path.join(ROOT, "ivln-ce_amd", "csrc")
so = "/tmp/libivln_bf3_timing.so"
# IVLN_PHASES_DEFS="-DBF3_PROBE_NO_SPLIT" | "-DBF3_PROBE_NO_MFMA": the 1x1 K loop without its VALU work / with one product of six
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-DBF3_TIMING", "-I", os.path.join(ROOT, "include")]
                      + os.environ.get("IVLN_PHASES_DEFS", "").split() + ["-c", os.path.join(CS, "conv_bf3.hip"), "-o", "/tmp/conv_bf3_timing.o"])
objs = [o for o in glob.glob(os.path.join(CS, "*.o")) if not o.endswith("conv_bf3.o")]
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so, "/tmp/conv_bf3_timing.o"] + objs)
from ivln_ce_amd import _lib  # noqa: E402

_lib._SO = so
import ivln_ce_amd  # noqa: E402,F401
from ivln_ce_amd import ops  # noqa: E402

L = _lib.lib()
L.ivln_conv_bf3_stamps.argtypes = [C.c_void_p, C.c_int]
dev = torch.device("cuda:0")
# (name, images, Cin, Cout, H = W, residual)
SH = [("256<-64@128", 8, 64, 256, 128, True), ("256<-64@64", 8, 64, 256, 64, True), ("512<-128@64", 8, 128, 512, 64, True),
      ("512<-128@32", 8, 128, 512, 32, True), ("1024<-256@32", 8, 256, 1024, 32, True), ("1024<-256@16", 8, 256, 1024, 16, True),
      ("2048<-512@8", 8, 512, 2048, 8, True), ("256<-1024@32", 8, 1024, 256, 32, False), ("256<-1024@16", 8, 1024, 256, 16, False),
      ("512<-2048@8", 8, 2048, 512, 8, False), ("512<-1024@16", 8, 1024, 512, 16, False), ("128<-512@64", 8, 512, 128, 64, False),
      ("64<-256@128", 8, 256, 64, 128, False)]
print(f"{'shape':<14} {'form':>4} {'waves':>6} {'launch us':>9} | per wave, us (median): prologue | K loop | epilogue | start spread, last end")
for name, n, cin, cout, hw, res in SH:
    x = torch.randn(n, cin, hw, hw, device=dev)
    w = torch.randn(cout, cin, 1, 1, device=dev) / cin ** 0.5
    r = torch.randn(n, cout, hw, hw, device=dev) if res else None
    ops.TILE_OVERRIDE = 11
    try:
        for _ in range(3):
            ops.conv2d(x, w, stride=1, pad=0, residual=r, relu=True, splitk=False)
    except Exception as e:  # noqa: BLE001
        print(f"{name:<14} not taken: {e}")
        ops.TILE_OVERRIDE = 0
        continue
    torch.cuda.synchronize()
    st = np.zeros(8192 * 8, dtype=np.uint64)
    L.ivln_conv_bf3_stamps(st.ctypes.data, st.size)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    ops.conv2d(x, w, stride=1, pad=0, residual=r, relu=True, splitk=False)
    b.record()
    torch.cuda.synchronize()
    ops.TILE_OVERRIDE = 0
    assert L.ivln_conv_bf3_stamps(st.ctypes.data, st.size) == 0
    st = st.reshape(8192, 8).astype(np.float64)
    st[:, :4] /= 100.0
    st = st[st[:, 3] > 0]
    med = np.median(st[:, :3], axis=0)
    cyc = np.median(st[:, 4:7], axis=0)
    t0 = st[:, 3].min()
    end = st[:, 3] + st[:, 0] + st[:, 1] + st[:, 2]
    form = "wt" if cin // 16 <= 16 else "ks"
    print(f"{name:<14} {form:>4} {len(st):6d} {a.elapsed_time(b) * 1e3:9.1f} | {med[0]:6.1f} | {med[1]:6.1f} | {med[2]:6.1f} | "
          f"starts within {st[:, 3].max() - t0:5.1f}, last wave ends at {end.max() - t0:5.1f} | K loop in shader cycles: waits + split "
          f"{cyc[0]:7.0f}, load issue {cyc[1]:6.0f}, MFMA issue {cyc[2]:7.0f} (= {cyc.sum() / med[1] / 1e3:.2f} GHz)")

# ---- 3x3, K split over the waves: 8 words per wave (prologue, K loop, reduction + epilogue, start, staging inside the K loop)
SH3 = [("512@8 x8", 8, 512, 512, 8), ("256@16 x8", 8, 256, 256, 16), ("128@32 x8", 8, 128, 128, 32), ("256@16 x4", 4, 256, 256, 16),
       ("512<-256@8", 8, 256, 512, 8), ("256<-512@16", 8, 512, 256, 16)]
print(f"{'3x3 shape':<14} {'waves':>6} {'launch us':>9} | per wave, us (median): prologue | K loop (of which staging) | reduction + epilogue | start spread, last end")
for name, n, cin, cout, hw in SH3:
    x = torch.randn(n, cin, hw, hw, device=dev)
    w = torch.randn(cout, cin, 3, 3, device=dev) / (9 * cin) ** 0.5
    ops.TILE_OVERRIDE = 10
    try:
        for _ in range(3):
            ops.conv2d(x, w, stride=1, pad=1, relu=True, splitk=False)
    except Exception as e:  # noqa: BLE001
        print(f"{name:<14} not taken: {e}")
        ops.TILE_OVERRIDE = 0
        continue
    torch.cuda.synchronize()
    st = np.zeros(8192 * 8, dtype=np.uint64)
    L.ivln_conv_bf3_stamps(st.ctypes.data, st.size)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    ops.conv2d(x, w, stride=1, pad=1, relu=True, splitk=False)
    b.record()
    torch.cuda.synchronize()
    ops.TILE_OVERRIDE = 0
    assert L.ivln_conv_bf3_stamps(st.ctypes.data, st.size) == 0
    st = st.reshape(8192, 8).astype(np.float64) / 100.0
    st = st[st[:, 3] > 0]
    med = np.median(st[:, [0, 1, 2, 4]], axis=0)
    t0 = st[:, 3].min()
    end = st[:, 3] + st[:, 0] + st[:, 1] + st[:, 2]
    print(f"{name:<14} {len(st):6d} {a.elapsed_time(b) * 1e3:9.1f} | {med[0]:6.1f} | {med[1]:6.1f} ({med[3]:5.1f}) | {med[2]:6.1f} | "
          f"starts within {st[:, 3].max() - t0:5.1f}, last wave ends at {end.max() - t0:5.1f}")
