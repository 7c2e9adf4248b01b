"""Where does the split-bf16 direct conv (csrc/conv_bf3.hip) spend its time?  Links a scratch copy of the library with that
file built -DBF3_TIMING (per-workgroup sums of the phases on the 100 MHz wall clock) and prints, per shape, the medians over
the workgroups: whole K loop, of which staging (wait for the patch loads + split + LDS writes + barrier) and MFMA phase,
and the epilogue.   usage: python tools/conv_bf3_phases.py"""
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
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-DBF3_TIMING", "-c",
                       os.path.join(CS, "conv_bf3.hip"), "-o", "/tmp/conv_bf3_timing.o"])
objs = [o for o in glob.glob(os.path.join(CS, "*.o")) if not o.endswith("conv_bf3.o")]
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so, "/tmp/conv_bf3_timing.o"] + objs)
from ivln_ce_amd import _lib  # noqa: E402

_lib._SO = so
import ivln_ce_amd  # noqa: E402,F401
from ivln_ce_amd import ops  # noqa: E402

L = _lib.lib()
L.ivln_conv_bf3_stamps.argtypes = [C.c_void_p, C.c_int]
dev = torch.device("cuda:0")
SH = [("map L1 fwd", 512, 14, 32, 64, 7), ("map L2 fwd", 512, 32, 64, 32, 7), ("map L3 fwd", 512, 64, 128, 16, 7),
      ("map L4 fwd", 512, 128, 128, 8, 7), ("map L2 dgrad", 512, 64, 32, 32, 7), ("rednet 64@128", 8, 64, 64, 128, 3),
      ("rednet 128@64", 8, 128, 128, 64, 3), ("rednet 256@32", 8, 256, 256, 32, 3), ("rednet 64@64 x8", 8, 64, 64, 64, 3),
      ("rednet 64@32 x8", 8, 64, 64, 32, 3), ("rednet 128@64 x8", 8, 128, 128, 64, 3),
      ("1x1 1024<-256@16", 16, 256, 1024, 16, 1), ("1x1 256<-1024@16", 16, 1024, 256, 16, 1), ("1x1 512<-128@32", 16, 128, 512, 32, 1)]
print(f"{'shape':<16} {'blocks':>6} {'launch us':>9} | per workgroup, us (median): K loop = staging + MFMA phase | epilogue")
for name, n, cin, cout, hw, ks in SH:
    x = torch.randn(n, cin, hw, hw, device=dev)
    w = torch.randn(cout, cin, ks, ks, device=dev) / (cin * ks * ks) ** 0.5
    ops.TILE_OVERRIDE = 9
    for _ in range(3):
        ops.conv2d(x, w, stride=1, pad=ks // 2, splitk=False)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    ops.conv2d(x, w, stride=1, pad=ks // 2, splitk=False)
    b.record()
    torch.cuda.synchronize()
    ops.TILE_OVERRIDE = 0
    st = np.zeros(8192 * 8, dtype=np.uint64)
    assert L.ivln_conv_bf3_stamps(st.ctypes.data, st.size) == 0
    st = st.reshape(8192, 8).astype(np.float64)
    st[:, :3] /= 100.0
    st = st[st[:, 0] > 0]
    med = np.median(st, axis=0)
    ent, k0, end = st[:, 4] / 100.0, st[:, 5] / 100.0, st[:, 6] / 100.0
    t0 = ent.min()
    print(f"{name:<16} {len(st):6d} {a.elapsed_time(b) * 1e3:9.1f} | {med[0]:8.1f} = {med[1]:7.1f} + {med[2]:7.1f} | shader cycles over "
          f"K loop + epilogue {med[3]:9.0f} (= {med[3] / med[0] / 1e3:.2f} GHz if the epilogue were free) | timeline: workgroups enter within "
          f"{ent.max() - t0:5.1f}, entry -> K loop {np.median(k0 - ent):5.1f}, K loop end -> stores landed {np.median(end - k0) - med[0]:5.1f}, last ends at {end.max() - t0:5.1f}")

print(f"{'7x7 weight gradient':<16} {'blocks':>6} {'launch us':>9} | per workgroup, us (median): whole loop = staging + MFMA phase | shader cycles")
for name, n, cin, cout, hw in [("map L1", 512, 14, 32, 64), ("map L2", 512, 32, 64, 32), ("map L3", 512, 64, 128, 16), ("map L4", 512, 128, 128, 8)]:
    x = torch.randn(n, cin, hw, hw, device=dev)
    dy = torch.randn(n, cout, hw, hw, device=dev)
    ops.TILE_OVERRIDE = 9
    for _ in range(3):
        ops.conv2d_bwd_weight(dy, x, 7, 7, pad=3)
    torch.cuda.synchronize()
    st = np.zeros(8192 * 8, dtype=np.uint64)
    L.ivln_conv_bf3_stamps(st.ctypes.data, st.size)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    ops.conv2d_bwd_weight(dy, x, 7, 7, pad=3)
    b.record()
    torch.cuda.synchronize()
    ops.TILE_OVERRIDE = 0
    assert L.ivln_conv_bf3_stamps(st.ctypes.data, st.size) == 0
    st = st.reshape(8192, 8).astype(np.float64)
    st[:, :3] /= 100.0
    st = st[st[:, 0] > 0]
    med = np.median(st, axis=0)
    print(f"{name:<16} {len(st):6d} {a.elapsed_time(b) * 1e3:9.1f} | {med[0]:8.1f} = {med[1]:7.1f} + {med[2]:7.1f} | {med[3]:9.0f} (= {med[3] / med[0] / 1e3:.2f} GHz)")
