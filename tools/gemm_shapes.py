"""Every ivln_gemm_f32 call of one DAgger update (or one RedNet forward) with its shape, operand modes and GPU
time (event pairs around each call, so launch gaps are included for tiny calls):
python tools/gemm_shapes.py [update|rednet]"""
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402

import ivln_ce_amd  # noqa: E402,F401
from ivln_ce_amd import ops  # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else "update"
records = []
orig = ops.gemm
live = [False]


def traced(desc):
    if not live[0]:
        return orig(desc)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    orig(desc)
    b.record()
    records.append(((desc.M, desc.N, desc.K, desc.amode, desc.bmode, desc.dmode, int(desc.defer_epilogue),
                     desc.stride, desc.Cin and (desc.K // max(1, desc.Cin))), a, b))


ops.gemm = traced
import bench_components as bc  # noqa: E402

fn = {"update": bc.bench_update, "rednet": bc.bench_rednet}[what]
_timeit = bc.timeit


def timeit(f, warm=2, iters=1):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    live[0] = True
    f()
    torch.cuda.synchronize()
    live[0] = False
    return 1.0


bc.timeit = timeit
fn()
agg = defaultdict(lambda: [0, 0.0])
for key, a, b in records:
    e = agg[key]
    e[0] += 1
    e[1] += a.elapsed_time(b) * 1e3
tot = sum(e[1] for e in agg.values())
print(f"{len(records)} GEMM-family calls, {tot / 1e3:.2f} ms (event pairs)")
try:
    import ctypes as _C

    from ivln_ce_amd._lib import lib as _lib

    _k = (_C.c_longlong * 4)()
    _lib().ivln_conv_split_kinds(_k, 0)
    print("split-bf16 launches since start (tiled, 3x3 K-split, 1x1 K-split, 1x1 wave tiles):", list(_k))
except Exception as e:  # noqa: BLE001
    print("kinds:", e)
print(f"{'M':>6} {'N':>8} {'K':>7} am bm dm df s taps  calls   us/call   total us   TFLOP/s")
for key, (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    M, N, K, am, bm, dm, df, st, taps = key
    fl = 2.0 * M * N * K * n
    print(f"{M:6d} {N:8d} {K:7d} {am:2d} {bm:2d} {dm:2d} {df:2d} {st} {taps:4d} {n:6d} {us / n:9.1f} {us:10.1f} {fl / us / 1e6:9.1f}")
