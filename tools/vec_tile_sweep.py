"""RedNet's 1x1 convolutions (B frames) under each vector-load GEMM tile (IVLN_VEC_TILE is read once per process:
run once per value).  python tools/vec_tile_sweep.py [B]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import ivln_ce_amd  # noqa: E402,F401
from ivln_ce_amd import ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda:0")
shapes = [(64, 64, 64, 256), (256, 64, 64, 64), (256, 32, 32, 512), (512, 32, 32, 128), (128, 32, 32, 512), (256, 16, 16, 1024),
          (1024, 16, 16, 256), (512, 16, 16, 1024), (512, 8, 8, 2048), (2048, 8, 8, 512), (1024, 8, 8, 2048)]
print("tile env", os.environ.get("IVLN_VEC_TILE", "default"), "B", B)
for Cin, H, W, Cout in shapes:
    x = torch.randn(B, Cin, H, W, device=dev)
    w = torch.randn(Cout, Cin, 1, 1, device=dev)
    sc, sh = torch.rand(Cout, device=dev), torch.rand(Cout, device=dev)
    res = torch.randn(B, Cout, H, W, device=dev)
    y = None
    for _ in range(5):
        y = ops.conv2d(x, w, scale=sc, shift=sh, residual=res, relu=True, out=y)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(50):
        ops.conv2d(x, w, scale=sc, shift=sh, residual=res, relu=True, out=y)
    b.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(b) * 1e3 / 50
    fl = 2.0 * Cout * Cin * B * H * W
    print(f"M={Cout:5d} N={B * H * W:6d} K={Cin:5d}  {us:7.1f} us  {fl / us / 1e6:6.1f} TFLOP/s")
