"""Which kernel / tile should a conv shape take?  Every candidate is captured as a hipGraph of 10 dependent launches (the
product replays graphs: a launch there costs what the graph makes it cost, reduction launches of split-K variants included)
and timed over replays; us per conv.   python tools/conv_cfg_sweep.py [shape-set]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import ivln_ce_amd  # noqa: E402,F401
from ivln_ce_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
# (name, images, Cin, Cout, H, W, KS, stride, weight groups, residual)
SH3 = [("dec 64@64 x8", 8, 64, 64, 64, 64, 3, 1, 0, 0), ("dec 64@128 x8", 8, 64, 64, 128, 128, 3, 1, 0, 0), ("enc 64@64 x16", 16, 64, 64, 64, 64, 3, 1, 2, 0),
       ("enc 128@32 x16", 16, 128, 128, 32, 32, 3, 1, 2, 0), ("dec 128@32 x8", 8, 128, 128, 32, 32, 3, 1, 0, 0), ("enc 256@16 x16", 16, 256, 256, 16, 16, 3, 1, 2, 0),
       ("dec 256@16 x8", 8, 256, 256, 16, 16, 3, 1, 0, 0), ("enc 512@8 x16", 16, 512, 512, 8, 8, 3, 1, 2, 0), ("dec 512@8 x8", 8, 512, 512, 8, 8, 3, 1, 0, 0)]
SH1 = [("256<-64@64 x16", 16, 64, 256, 64, 64, 1, 1, 2, 1), ("64<-256@64 x16", 16, 256, 64, 64, 64, 1, 1, 2, 0), ("512<-128@32 x16", 16, 128, 512, 32, 32, 1, 1, 2, 1),
       ("128<-512@32 x16", 16, 512, 128, 32, 32, 1, 1, 2, 0), ("1024<-256@16 x16", 16, 256, 1024, 16, 16, 1, 1, 2, 1), ("256<-1024@16 x16", 16, 1024, 256, 16, 16, 1, 1, 2, 0),
       ("2048<-512@8 x16", 16, 512, 2048, 8, 8, 1, 1, 2, 1), ("512<-2048@8 x16", 16, 2048, 512, 8, 8, 1, 1, 2, 0),
       ("128<-256@64 x16", 16, 256, 128, 64, 64, 1, 1, 2, 0), ("256<-512@32 x16", 16, 512, 256, 32, 32, 1, 1, 2, 0), ("512<-1024@16 x16", 16, 1024, 512, 16, 16, 1, 1, 2, 0),
       ("512<-2048@8 x8", 8, 2048, 512, 8, 8, 1, 1, 0, 0), ("256<-1024@16 x8", 8, 1024, 256, 16, 16, 1, 1, 0, 0), ("128<-512@32 x8", 8, 512, 128, 32, 32, 1, 1, 0, 0),
       ("64<-256@64 x8", 8, 256, 64, 64, 64, 1, 1, 0, 0), ("64<-64@128 x8", 8, 64, 64, 128, 128, 1, 1, 0, 0)]
# candidates: (label, tile_override, splitk)
CAND3 = [("heuristic", 0, True), ("fp32 direct", 6, True), ("bf16 64x512", 21, False), ("bf16 64x256", 22, False), ("bf16 128x256", 23, False),
         ("bf16 64x128", 24, False), ("bf16 64x128 split", 24, True), ("bf16 128x128", 25, False), ("bf16 128x128 split", 25, True),
         ("K/waves 32 px", 14, False), ("K/waves 64 px", 15, False)]
CAND1 = [("heuristic", 0, True), ("fp32", 7, True), ("fp32 scalar", 1, True), ("bf16 64x256", 22, False), ("bf16 128x256", 23, False), ("bf16 64x128", 24, False),
         ("bf16 64x128 split", 24, True), ("bf16 128x128", 25, False), ("bf16 128x128 split", 25, True), ("K over waves", 12, False), ("wave tiles", 13, False)]
which = sys.argv[1] if len(sys.argv) > 1 else "all"
SETS = ([(SH3, CAND3)] if which in ("all", "3x3") else []) + ([(SH1, CAND1)] if which in ("all", "1x1") else [])


def graph_time(f, n=10, reps=20):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            f()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        for _ in range(n):
            f()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / (n * reps)


import ctypes as _C  # noqa: E402

from ivln_ce_amd._lib import lib as _lib  # noqa: E402


def kinds():
    k = (_C.c_longlong * 4)()
    _lib().ivln_conv_split_kinds(k, 0)
    return list(k)


print("(letter: the split-bf16 kernel that ran - t tiled, k 3x3 K over waves, 1 / w the 1x1 forms, f none: an fp32 kernel; ! = result differs from the heuristic's)")
for SH, CAND in SETS:
    print(f"{'shape':<18}" + "".join(f"{c[0]:>20}" for c in CAND))
    for name, n, cin, cout, h, w_, ks, st, G, has_res in SH:
        x = torch.randn(n, cin, h, w_, device=dev)
        wshape = (G, cout, cin, ks, ks) if G else (cout, cin, ks, ks)
        w = torch.randn(*wshape, device=dev) / (cin * ks * ks) ** 0.5
        sc, sh = torch.rand(max(G, 1) * cout, device=dev) + 0.5, torch.randn(max(G, 1) * cout, device=dev)
        out = torch.empty(n, cout, (h + 2 * (ks // 2) - ks) // st + 1, (w_ + 2 * (ks // 2) - ks) // st + 1, device=dev)
        res = torch.randn_like(out) if has_res else None
        line = f"{name:<18}"
        ref = None
        for label, mode, sk in CAND:
            def f():
                ops.TILE_OVERRIDE = mode
                try:
                    ops.conv2d(x, w, stride=st, pad=ks // 2, scale=sc, shift=sh, residual=res, relu=True, out=out, splitk=sk)
                finally:
                    ops.TILE_OVERRIDE = 0
            try:
                k0 = kinds()
                f()
                torch.cuda.synchronize()
                dk = [b - a for a, b in zip(k0, kinds())]
                tag = "".join(c for c, v in zip("tk1w", dk) if v) or "f"
                if ref is None:
                    ref = out.clone()
                elif float((out - ref).abs().max()) > 1e-3 * float(ref.abs().max()):
                    tag += "!"
                line += f"{graph_time(f):17.1f} {tag:<2}"
            except Exception as e:  # noqa: BLE001
                line += f"{'-':>20}"
        print(line, flush=True)
