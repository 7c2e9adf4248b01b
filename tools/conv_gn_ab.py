"""A/B of the two ways to run a conv -> GroupNorm(16) -> ReLU pair of the DD-PPO depth ResNet at rollout batch sizes:
  pair   ivln_gemm_f32 (deferred split-K slabs) + ivln_groupnorm_f32 (slab reduction fused)   2 launches
  fused  ivln_conv_gn_f32 (workgroup per (image, group), whole K inside the block)            1 launch
Each variant is captured 40x back to back in a hipGraph (dependent launches, like the encoder's chain) and replayed;
time per pair = replay time / 40.   python tools/conv_gn_ab.py [envs]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import ivln_ce_amd  # noqa: E402,F401
from ivln_ce_amd import ops  # noqa: E402

DEV = torch.device("cuda:0")
SHAPES = [  # name, Cin, H, W, Cout, k, s, p
    ("stem 7x7 s2 1->32 @128", 1, 128, 128, 32, 7, 2, 3),
    ("l1 1x1 128->32 @32", 128, 32, 32, 32, 1, 1, 0), ("l1 3x3 32->32 @32", 32, 32, 32, 32, 3, 1, 1),
    ("l1 1x1 32->128 @32", 32, 32, 32, 128, 1, 1, 0),
    ("l2 1x1 256->64 @16", 256, 16, 16, 64, 1, 1, 0), ("l2 3x3 64->64 @16", 64, 16, 16, 64, 3, 1, 1),
    ("l2 1x1 64->256 @16", 64, 16, 16, 256, 1, 1, 0),
    ("l3 1x1 512->128 @8", 512, 8, 8, 128, 1, 1, 0), ("l3 3x3 128->128 @8", 128, 8, 8, 128, 3, 1, 1),
    ("l3 1x1 128->512 @8", 128, 8, 8, 512, 1, 1, 0),
    ("l4 1x1 1024->256 @4", 1024, 4, 4, 256, 1, 1, 0), ("l4 3x3 256->256 @4", 256, 4, 4, 256, 3, 1, 1),
    ("l4 1x1 256->1024 @4", 256, 4, 4, 1024, 1, 1, 0),
]
REP = 40


def timed(fn):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3):
            fn()
    torch.cuda.synchronize()
    ops.settle_packed_weights()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(REP):
            fn()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    return 1e3 * a.elapsed_time(b) / (10 * REP)


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    print(f"envs {N}: us per conv+GroupNorm+ReLU pair (dependent launches inside a replayed graph)")
    tot = [0.0, 0.0]
    for name, Cin, H, W, Cout, k, s, p in SHAPES:
        x = torch.randn(N, Cin, H, W, device=DEV)
        w = torch.randn(Cout, Cin, k, k, device=DEV) / (Cin * k * k) ** 0.5
        gn = torch.nn.GroupNorm(16, Cout).to(DEV)

        def pair():
            y = ops.conv2d(x, w, stride=s, pad=p, defer=True)
            return ops.groupnorm(y, gn.weight, gn.bias, 16, 1e-5, relu=True)

        def fused():
            return ops.conv_gn(x, w, gn, stride=s, pad=p, relu=True, force=True)

        assert fused() is not None
        tp, tf = timed(pair), timed(fused)
        tot[0] += tp
        tot[1] += tf
        print(f"  {name:26s} pair {tp:7.2f}   fused {tf:7.2f}   x{tp / tf:.2f}")
    print(f"  sum over the 13 shapes      pair {tot[0]:7.2f}   fused {tot[1]:7.2f}")


if __name__ == "__main__":
    main()
