"""RedNet's deep pixel-starved conv launches with their weights L2-warm (one set, back to back), memory-side-cache-warm (8 sets in
rotation) and HBM-cold (48 sets): what prefetching the next layer's weights could buy (profiles/r06_weight_warmth_probe.txt)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ivln_ce_amd  # noqa
from ivln_ce_amd import ops
dev = "cuda:0"
def timeit(f, reps, n=20):
    for _ in range(2): f()
    torch.cuda.synchronize()
    ops.settle_packed_weights()
    st = torch.cuda.Stream(); gr = torch.cuda.CUDAGraph()
    with torch.cuda.stream(st):
        f(); torch.cuda.synchronize()
        with torch.cuda.graph(gr, stream=st):
            f()
    torch.cuda.synchronize(); gr.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): gr.replay()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / (n * reps) * 1e3
g = torch.Generator().manual_seed(0)
for (name, Cin, Cout, HW, N, ks) in (("3x3 512->512 @8x8 x8", 512, 512, 8, 8, 3), ("3x3 256->256 @16x16 x16", 256, 256, 16, 16, 3),
                                     ("1x1 1024->256 @16x16 x16", 1024, 256, 16, 16, 1), ("1x1 256->1024 @16x16 x16", 256, 1024, 16, 16, 1),
                                     ("1x1 2048->512 @8x8 x16", 2048, 512, 8, 16, 1)):
    x = torch.randn(N, Cin, HW, HW, generator=g).to(dev)
    sc, sh = (torch.rand(Cout, generator=g) + 0.5).to(dev), torch.randn(Cout, generator=g).to(dev)
    out = []
    for nsets in (1, 8, 48):
        ws = [(torch.randn(Cout, Cin, ks, ks, generator=g) / (Cin * ks * ks) ** 0.5).to(dev) for _ in range(nsets)]
        reps = 48
        def f():
            for i in range(reps):
                ops.conv2d(x, ws[i % nsets], pad=ks // 2, scale=sc, shift=sh, relu=True)
        out.append(timeit(f, reps))
        del ws
    mb = Cout * Cin * ks * ks * 6 / 1e6
    print(f"{name}: weights {mb:.1f} MB split | same weights every launch {out[0]:.1f} us | 8 sets in rotation {out[1]:.1f} us | 48 sets {out[2]:.1f} us", flush=True)
