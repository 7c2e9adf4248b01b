"""Reproducibility under load: RedNet's forward (every split-bf16 kernel form of the round) N times on the same frames - the
scores of every run have to equal the first run's bit for bit (no atomics, fixed summation orders; a sporadic hazard - like the
store hazard tools/dbg_fuse.py found - shows up as a mismatch), eagerly and as a replayed graph beside a busy second stream.
python tools/rednet_repro.py [runs]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch  # noqa: E402

import ivln_ce_amd  # noqa: E402,F401
from det_init import det_fill  # noqa: E402
from ivln_ce_amd import rednet  # noqa: E402

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 300
dev = torch.device("cuda:0")
net = det_fill(rednet.RedNet(rednet.PredictSemantics.CFG), seed=1, conv_gain=0.6).to(dev).eval()
for p in net.parameters():
    p.requires_grad = False
g = torch.Generator().manual_seed(3)
rgb = torch.randn(8, 3, 256, 256, generator=g).to(dev)
dep = torch.randn(8, 1, 256, 256, generator=g).to(dev)
with torch.no_grad():
    ref = net(rgb, dep).clone()
    bad = 0
    for i in range(runs):
        out = net(rgb, dep)
        if not torch.equal(out, ref):
            bad += 1
            d = (out - ref).abs()
            print(f"eager run {i}: {int((d > 0).sum())} elements differ, max {float(d.max()):.3e}")
    print(f"eager: {bad} of {runs} runs differ from the first")
    # replayed, with a second stream hammering the memory system and the CUs
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            net(rgb, dep)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr, stream=s):
        gout = net(rgb, dep)
    noise = torch.cuda.Stream()
    big = torch.randn(64 << 20, device=dev)
    bad = 0
    for i in range(runs):
        with torch.cuda.stream(noise):
            for _ in range(4):
                big.mul_(1.0000001)
        gr.replay()
        torch.cuda.synchronize()
        if not torch.equal(gout, ref):
            bad += 1
            d = (gout - ref).abs()
            print(f"replay {i}: {int((d > 0).sum())} elements differ, max {float(d.max()):.3e}")
    print(f"replayed beside a busy stream: {bad} of {runs} runs differ from the eager first run")
