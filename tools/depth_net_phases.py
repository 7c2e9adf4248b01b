"""Where does the persistent depth encoder (csrc/depth_net.hip) spend its time?  Builds the file with -DDEPTH_NET_TIMING
into a scratch library (per-op wall-clock stamps of cluster 0's 32 workgroups), runs the encoder at N images and prints
per op, in us: wait (arrival of the slowest workgroup), stats (partials -> scale / shift table), stage (input rows -> LDS),
mma, reduce (K ranges -> output tile), store (+ statistics partials), arrive - medians over the workgroups that had a task -
and the op's span from the first workgroup's start to the last one's arrival.   usage: python tools/depth_net_phases.py [N=4]"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ivln_ce_amd  # noqa: E402,F401
from ivln_ce_amd import depth_net  # noqa: E402
from ivln_ce_amd.encoders import ResNetEncoder  # noqa: E402

def op_next_bar(ops_, oi):
    return oi + 1 < len(ops_) and ops_[oi + 1]["barrier_before"]


so = "/tmp/libdepthnet_timing.so"
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-DDEPTH_NET_TIMING",
                       os.path.join(ROOT, "ivln-ce_amd", "csrc", "depth_net.hip"), "-o", so])
L = C.CDLL(so)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device("cuda:0")
torch.manual_seed(0)
enc = ResNetEncoder((256, 256, 1)).to(dev).eval()
plan = depth_net.DepthNetPlan(enc, dev)
vp, i64, I32 = C.c_void_p, C.c_int64, C.c_int
L.ivln_depth_net_f32.argtypes = [vp, C.POINTER(depth_net.DepthNetOp), I32, vp, vp, vp, i64, vp, i64, vp, i64, I32, C.c_float, vp, vp]
L.ivln_depth_net_status.argtypes = [vp, vp]
L.ivln_depth_net_stamps.argtypes = [vp, I32]
plan._L = L
depth = torch.rand(N, 256, 256, 1, device=dev)
out = torch.empty(N, 128, 4, 4, device=dev)
for _ in range(5):
    assert plan.run(depth, out, 2048)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    plan.run(depth, out, 2048)
e1.record()
torch.cuda.synchronize()
plan.check_status()
print(f"N = {N}: {1e3 * e0.elapsed_time(e1) / 20:.1f} us per launch (event pair over 20 back-to-back launches)")
n_ops = len(plan.prog.ops)
st = np.zeros(32 * 64 * 12, dtype=np.uint64)
assert L.ivln_depth_net_stamps(st.ctypes.data, st.size) == 0
st = st.reshape(32, 64, 12).astype(np.float64) / 100.0  # us
names = ["wait", "stats", "stage", "mma", "reduce", "store", "arrive"]
print(f"{'op':>3s} {'conv':>22s} {'tasks':>5s} " + " ".join(f"{n:>7s}" for n in names) + f" {'span':>7s} {'start..next':>11s}")
tot = np.zeros(len(names))
t_first = st[:, 0, 0].min()
for oi, op in enumerate(plan.prog.ops[:64]):
    if op["kind"]:
        print(f"{oi:3d} final GroupNorm: wait {np.median(st[:, oi, 1] - st[:, oi, 0]):.2f}, total {st[0, oi, 6] - st[0, oi, 0]:.2f}")
        continue
    nt = op["n_ctg"] * op["n_ptg"] * op["kwg"]
    s = st[:nt, oi]
    ph = [np.median(s[:, k + 1] - s[:, k]) for k in range(6)]
    ph.append(np.median(s[:, 7] - s[:, 6]) if s[:, 7].max() > s[:, 6].min() else 0.0)
    tot += np.array(ph)
    span = (st[:, oi, 7].max() if op_next_bar(plan.prog.ops, oi) else st[:, oi, 6].max()) - st[:, oi, 0].min()
    nxt = st[:, oi + 1, 0].min() - st[:, oi, 0].min() if oi + 1 < n_ops else 0.0
    desc = f"{op['Cin']}->{op['Cout']} k{op['ks']}s{op['stride']} @{1 << op['wout_shift']}"
    pre, loop, bar = np.median(s[:, 8] - s[:, 3]), np.median(s[:, 9] - s[:, 8]), np.median(s[:, 4] - s[:, 9])
    print(f"{oi:3d} {desc:>22s} {nt:5d} " + " ".join(f"{v:7.2f}" for v in ph) + f" {span:7.2f} {nxt:11.2f}   mma = setup {pre:.2f} + loop(wave 0) {loop:.2f} + barrier {bar:.2f}")
print("sum of medians: " + " ".join(f"{n} {v:.1f}" for n, v in zip(names, tot)) + f"; first start -> last stamp {st[:, :n_ops].max() - t_first:.1f} us")
