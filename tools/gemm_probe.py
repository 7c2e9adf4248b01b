"""Run one conv shape of the MFMA implicit-GEMM kernel a few times (for rocprofv3 --pmc passes).
python tools/gemm_probe.py N Cin H W Cout k s p [tile_override]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import ivln_ce_amd  # noqa: E402,F401
from ivln_ce_amd import ops  # noqa: E402

N, Cin, H, W, Cout, k, s, p = [int(a) for a in sys.argv[1:9]]
ops.TILE_OVERRIDE = int(sys.argv[9]) if len(sys.argv) > 9 else 0
dev = torch.device("cuda:0")
x = torch.randn(N, Cin, H, W, device=dev)
w = torch.randn(Cout, Cin, k, k, device=dev)
y = None
for _ in range(5):
    y = ops.conv2d(x, w, stride=s, pad=p, out=y, relu=True)
torch.cuda.synchronize()
print("done", float(y.abs().mean()))
