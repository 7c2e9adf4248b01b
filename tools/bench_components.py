"""Component benchmarks on the GPU box (not the driver contract - see bench.py):
RedNet forward, pred-semantics rollout step, DAgger update step, and a conv-shape sweep of the MFMA
implicit-GEMM kernel.  python tools/bench_components.py [rednet|update|convs|all]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import ivln_ce_amd  # noqa: E402,F401
from ivln_ce_amd import ops  # noqa: E402

DEV = torch.device("cuda:0")


def timeit(fn, warm=3, iters=10):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


def bench_convs():
    shapes = [  # (name, N, Cin, H, W, Cout, k, s, p)
        ("rednet 3x3 64->64 @128^2 B8", 8, 64, 128, 128, 64, 3, 1, 1),
        ("rednet 3x3 256->256 @16^2 B8", 8, 256, 16, 16, 256, 3, 1, 1),
        ("rednet 3x3 512->512 @8^2 B8", 8, 512, 8, 8, 512, 3, 1, 1),
        ("rednet 3x3 128->128 @32^2 B8", 8, 128, 32, 32, 128, 3, 1, 1),
        ("rednet 1x1 256->1024 @16^2 B8", 8, 256, 16, 16, 1024, 1, 1, 0),
        ("rednet 1x1 64->256 @64^2 B8", 8, 64, 64, 64, 256, 1, 1, 0),
        ("rednet stem 7x7 3->64 s2 @256^2 B8", 8, 3, 256, 256, 64, 7, 2, 3),
        ("mapcnn 7x7 14->32 @64^2 TN512", 512, 14, 64, 64, 32, 7, 1, 3),
        ("mapcnn 7x7 32->64 @32^2 TN512", 512, 32, 32, 32, 64, 7, 1, 3),
        ("mapcnn 7x7 64->128 @16^2 TN512", 512, 64, 16, 16, 128, 7, 1, 3),
        ("mapcnn 7x7 128->128 @8^2 TN512", 512, 128, 8, 8, 128, 7, 1, 3),
        ("gemm-like 1x1 1024->1024 @64^2 B1", 1, 1024, 64, 64, 1024, 1, 1, 0),
    ]
    out = []
    for name, N, Cin, H, W, Cout, k, s, p in shapes:
        x = torch.randn(N, Cin, H, W, device=DEV)
        w = torch.randn(Cout, Cin, k, k, device=DEV)
        Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
        y = torch.empty(N, Cout, Ho, Wo, device=DEV)
        fl = 2.0 * N * Ho * Wo * Cout * Cin * k * k
        row = []
        for ov in (0, 1, 4, 5):
            ops.TILE_OVERRIDE = ov
            t = timeit(lambda: ops.conv2d(x, w, stride=s, pad=p, out=y, relu=True), 3, 20)
            row.append(round(fl / t / 1e12, 1))
        ops.TILE_OVERRIDE = 0
        out.append((name, row))
        print(f"{name:42s} TFLOP/s auto {row[0]:6.1f} | 64x64 {row[1]:6.1f} | 128x128 {row[2]:6.1f} | 64x128 {row[3]:6.1f}", flush=True)
    return out


def bench_rednet(B=8):
    from ivln_ce_amd.rednet import PredictSemantics, RedNet

    torch.manual_seed(0)
    net = RedNet(PredictSemantics.CFG).to(DEV).eval()
    ps = PredictSemantics(DEV, model=net)
    obs = {"rgb": torch.randint(0, 256, (B, 224, 224, 3), dtype=torch.uint8, device=DEV),
           "depth": torch.rand(B, 256, 256, 1, device=DEV)}
    t = timeit(lambda: ps(obs), 2, 5)
    fl = 38.72e9 * B
    print(f"RedNet fwd B={B}: {t*1e3:.2f} ms  {fl/t/1e12:.2f} TFLOP/s  {B/t:.1f} frames/s", flush=True)
    return {"B": B, "ms": t * 1e3, "tflops": fl / t / 1e12}


def bench_update(T=64, N=8):
    sys.path.insert(0, ROOT)
    from bench import make_policy
    from ivln_ce_amd.trainers import FlatAdam, update_agent

    cfg, pol = make_policy(DEV)
    pol.train()
    opt = FlatAdam(pol, lr=2.5e-4)
    g = torch.Generator().manual_seed(0)
    TN = T * N
    instr = torch.zeros(N, 200)
    instr[:, :80] = torch.randint(2, 2504, (N, 80), generator=g).float()
    obs = {"depth_features": torch.randn(TN, 128, 4, 4, generator=g).to(DEV),
           "occupancy_map": (torch.rand(TN, 64, 64, generator=g) < 0.3).float().to(DEV),
           "semantic_map": torch.randint(0, 13, (TN, 64, 64), generator=g).float().to(DEV),
           "instruction": instr.repeat(T, 1)}
    if not os.environ.get("IVLN_NO_TRIM"):
        from ivln_ce_amd.utils import trim_instruction_padding

        obs = trim_instruction_padding(obs)  # what trainers.PrefetchLoader does on the host
    obs["instruction"] = obs["instruction"].to(DEV)
    prev = torch.randint(0, 4, (TN, 1), generator=g).to(DEV)
    nd = torch.ones(T, N, dtype=torch.uint8)
    nd[0] = 0
    nd = nd.view(-1, 1).to(DEV)
    tgt = torch.randint(0, 4, (T, N), generator=g).to(DEV)
    w = torch.ones(T, N).to(DEV)
    t = timeit(lambda: update_agent(pol, opt, obs, prev, nd, tgt, w), 2, 5)
    print(f"update T={T} N={N}: {t*1e3:.1f} ms  {TN/t:.0f} rows/s  {2.02e9*TN/t/1e12:.2f} TFLOP/s (2.02 GFLOP/row)", flush=True)
    return {"T": T, "N": N, "ms": t * 1e3, "rows_per_s": TN / t}


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    res = {}
    if what in ("convs", "all"):
        res["convs"] = bench_convs()
    if what in ("rednet", "all"):
        res["rednet"] = bench_rednet()
    if what in ("update", "all"):
        res["update"] = bench_update()
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(res, open(os.path.join(ROOT, "gpurun_out", "components.json"), "w"), indent=1)
