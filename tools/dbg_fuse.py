import os, sys, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import ivln_ce_amd
from ivln_ce_amd import ops, rednet
from det_init import det_fill
dev = torch.device("cuda:0")
net = det_fill(rednet.RedNet(rednet.PredictSemantics.CFG), seed=1, conv_gain=0.6).to(dev).eval()
for p in net.parameters():
    p.requires_grad = False
g = torch.Generator().manual_seed(3)
rgb = torch.randn(8, 3, 256, 256, generator=g).to(dev)
dep = torch.randn(8, 1, 256, 256, generator=g).to(dev)
orig = ops.conv3x3_then_1x1
calls = []
def both(x, w2, s2, b2, w3, s3, b3, res):
    got = orig(x, w2, s2, b2, w3, s3, b3, res)
    if got is None:
        return None
    y = ops.conv2d(x, w2, pad=1, scale=s2, shift=b2, relu=True)
    two = ops.conv2d(y, w3, scale=s3, shift=b3, residual=res, relu=True)
    d = (got - two).abs()
    info = (tuple(x.shape), tuple(w2.shape), tuple(w3.shape), float(d.max()), float(two.abs().max()), float(y.abs().max()), bool(torch.isfinite(two).all()))
    calls.append(info)
    for rep in range(6):
        g2 = orig(x, w2, s2, b2, w3, s3, b3, res)
        d = (g2 - two).abs()
        idx = (d > 1e-6 * float(two.abs().max())).nonzero()
        if idx.shape[0]:
            el = [(int(i[0]), int(i[1]) // 32, "half", (int(i[1]) % 32 // 4) % 2, "r", (int(i[1]) % 4) + 4 * (int(i[1]) % 32 // 8), "tile", (int(i[2]) // 4, int(i[3]) // 32), "l31", (int(i[2]) % 4) * 8 + (int(i[3]) % 32) // 4, "e", int(i[3]) % 4, round(float(d[tuple(i)]), 4)) for i in idx[:12]]
            print("  BAD rep", rep, tuple(x.shape), "n", idx.shape[0], el)
    return got
ops.conv3x3_then_1x1 = both
with torch.no_grad():
    s1 = net(rgb, dep)
for c in calls:
    print(c)
ops.conv3x3_then_1x1 = lambda *a: None
with torch.no_grad():
    s0 = net(rgb, dep)
print("scores fused vs unfused: max diff", float((s1 - s0).abs().max()), "of", float(s0.abs().max()))
