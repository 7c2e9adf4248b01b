"""Split-bf16 direct conv (csrc/conv_bf3.hip) against the fp32 MFMA direct conv (csrc/conv_direct.hip): error of both against
a float64 convolution on a few images, time of both at the full batch.
python tools/conv_bf3_probe.py [map|rednet|all]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

import ivln_ce_amd  # noqa: E402,F401
from ivln_ce_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
what = sys.argv[1] if len(sys.argv) > 1 else "all"
# (name, images, Cin, Cout, H, KS)
MAP = [("map L1 fwd", 512, 14, 32, 64, 7), ("map L1 fwd one-hot", 512, 14, 32, 64, 7), ("map L2 fwd", 512, 32, 64, 32, 7), ("map L3 fwd", 512, 64, 128, 16, 7),
       ("map L4 fwd", 512, 128, 128, 8, 7), ("map L2 dgrad", 512, 64, 32, 32, 7), ("map L3 dgrad", 512, 128, 64, 16, 7)]
RED = [("rednet 64@128 x8", 8, 64, 64, 128, 3), ("rednet 64@64 x16", 16, 64, 64, 64, 3), ("rednet 64@64 x8", 8, 64, 64, 64, 3),
       ("rednet 128@32 x16", 16, 128, 128, 32, 3), ("rednet 128@32 x8", 8, 128, 128, 32, 3), ("rednet 256@16 x16", 16, 256, 256, 16, 3),
       ("rednet 256@16 x8", 8, 256, 256, 16, 3), ("rednet 512@8 x16", 16, 512, 512, 8, 3), ("rednet 512@8 x8", 8, 512, 512, 8, 3)]
ONE = [("1x1 1024<-256 @16 x16", 16, 256, 1024, 16, 1), ("1x1 256<-1024 @16 x16", 16, 1024, 256, 16, 1), ("1x1 512<-128 @32 x16", 16, 128, 512, 32, 1),
       ("1x1 2048<-512 @8 x16", 16, 512, 2048, 8, 1), ("1x1 128<-512 @32 x16", 16, 512, 128, 32, 1), ("1x1 512<-2048 @8 x16", 16, 2048, 512, 8, 1),
       ("1x1 128<-256 @64 x16", 16, 256, 128, 64, 1), ("1x1 512<-1024 @16 x16", 16, 1024, 512, 16, 1), ("1x1 256<-512 @32 x16", 16, 512, 256, 32, 1),
       ("1x1 1024<-256 @16 x8", 8, 256, 1024, 16, 1), ("1x1 256<-1024 @16 x8", 8, 1024, 256, 16, 1)]
RED = RED + ONE if what in ("rednet", "all", "one") else RED
if what == "one":
    RED = ONE
shapes = (MAP if what in ("map", "all") else []) + (RED if what in ("rednet", "all", "one") else [])


def run(x, w, b, mode):
    ops.TILE_OVERRIDE = mode
    try:
        return ops.conv2d(x, w, stride=1, pad=w.shape[-1] // 2, shift=b, splitk=True)
    except Exception:  # noqa: BLE001
        if mode == 6:  # (1x1: the fp32 path is the float4-staged GEMM, not the direct kernel)
            ops.TILE_OVERRIDE = 0
            sb = ops.SPLIT_BF16
            ops.SPLIT_BF16 = False
            try:
                return ops.conv2d(x, w, stride=1, pad=w.shape[-1] // 2, shift=b, splitk=True)
            finally:
                ops.SPLIT_BF16 = sb
        raise
    finally:
        ops.TILE_OVERRIDE = 0


def timeit(f, iters=10):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        f()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


if what in ("wgrad",):
    print(f"{'7x7 weight gradient':<22} {'GFLOP':>7} | {'fp32 MFMA us':>12} {'TF/s':>6} | {'split-bf16 us':>13} {'TF/s':>6} {'x':>5} | max err vs f64 / max|dW|: fp32 MFMA, split-bf16")
    for name, n, cin, cout, hw in [("map L1", 512, 14, 32, 64), ("map L1 one-hot x", 512, 14, 32, 64), ("map L2", 512, 32, 64, 32), ("map L3", 512, 64, 128, 16), ("map L4", 512, 128, 128, 8)]:
        torch.manual_seed(2)
        x = torch.randn(n, cin, hw, hw, device=dev)
        if "one-hot" in name:
            x = (torch.rand(n, cin, hw, hw, device=dev) > 0.9).float()
        dy = torch.randn(n, cout, hw, hw, device=dev)

        def wg(mode):
            ops.TILE_OVERRIDE = mode
            try:
                return ops.conv2d_bwd_weight(dy, x, 7, 7, pad=3)
            finally:
                ops.TILE_OVERRIDE = 0

        g9, g6 = wg(9), wg(6)
        k = 16
        w = torch.zeros(cout, cin, 7, 7, dtype=torch.float64, requires_grad=True)
        F.conv2d(x[:k].double().cpu(), w, padding=3).backward(dy[:k].double().cpu())
        ops.TILE_OVERRIDE = 9
        s9 = ops.conv2d_bwd_weight(dy[:k].contiguous(), x[:k].contiguous(), 7, 7, pad=3)
        ops.TILE_OVERRIDE = 6
        s6 = ops.conv2d_bwd_weight(dy[:k].contiguous(), x[:k].contiguous(), 7, 7, pad=3)
        ops.TILE_OVERRIDE = 0
        sc = w.grad.abs().max().item()
        e9 = (s9.double().cpu() - w.grad).abs().max().item() / sc
        e6 = (s6.double().cpu() - w.grad).abs().max().item() / sc
        flops = 2.0 * n * hw * hw * cout * cin * 49
        t6, t9 = timeit(lambda: wg(6)), timeit(lambda: wg(9))
        print(f"{name:<22} {flops / 1e9:7.1f} | {t6:12.1f} {flops / t6 / 1e6:6.1f} | {t9:13.1f} {flops / t9 / 1e6:6.1f} {t6 / t9:5.2f} | {e6:.2e} {e9:.2e}")
    sys.exit(0)
print(f"{'shape':<18} {'GFLOP':>7} | {'fp32 MFMA us':>12} {'TF/s':>6} | {'split-bf16 us':>13} {'TF/s':>6} {'x':>5} | max err vs f64 / max|y|: fp32 MFMA, split-bf16, torch f32")
for name, n, cin, cout, hw, ks in shapes:
    torch.manual_seed(1)
    x = torch.randn(n, cin, hw, hw, device=dev)
    if "one-hot" in name:  # the map CNN's real first-layer input: occupancy + one-hot labels, exact in one bf16 piece
        x = (torch.rand(n, cin, hw, hw, device=dev) > 0.9).float()
    w = torch.randn(cout, cin, ks, ks, device=dev) / (cin * ks * ks) ** 0.5
    b = torch.randn(cout, device=dev)
    flops = 2.0 * n * hw * hw * cout * cin * ks * ks
    try:
        y9 = run(x, w, b, 9)
    except Exception as e:  # noqa: BLE001
        print(f"{name:<18} split-bf16 not eligible: {e}")
        continue
    y6 = run(x, w, b, 6)
    k = min(n, 4)
    ref = F.conv2d(x[:k].double().cpu(), w.double().cpu(), b.double().cpu(), padding=ks // 2)
    yt = F.conv2d(x[:k].cpu(), w.cpu(), b.cpu(), padding=ks // 2)
    sc = ref.abs().max().item()
    e6 = (y6[:k].double().cpu() - ref).abs().max().item() / sc
    e9 = (y9[:k].double().cpu() - ref).abs().max().item() / sc
    et = (yt.double() - ref).abs().max().item() / sc
    t6 = timeit(lambda: run(x, w, b, 6))
    t9 = timeit(lambda: run(x, w, b, 9))
    print(f"{name:<18} {flops / 1e9:7.1f} | {t6:12.1f} {flops / t6 / 1e6:6.1f} | {t9:13.1f} {flops / t9 / 1e6:6.1f} {t6 / t9:5.2f} | "
          f"{e6:.2e} {e9:.2e} {et:.2e}")
