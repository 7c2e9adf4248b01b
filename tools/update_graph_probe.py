"""Feasibility probe: the DAgger update's forward + loss + backward (no optimizer step) captured as ONE hipGraph and replayed,
against the same region issued eagerly.  python tools/update_graph_probe.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
import ivln_ce_amd  # noqa: E402,F401
from ivln_ce_amd import ops, train as _train  # noqa: E402
from ivln_ce_amd.aux_losses import AuxLosses  # noqa: E402

dev = torch.device("cuda:0")
policy = bench.make_policy(dev)
policy = policy[1] if isinstance(policy, tuple) else policy
leg = bench.UpdateLeg(policy, dev, 1)
AuxLosses.activate()
obs, prev, nd, tgt, w = leg.args
T, N = tgt.shape


def region():
    AuxLosses.clear()
    h0 = torch.zeros(N, policy.net.num_recurrent_layers, 512, device=dev)
    with torch.enable_grad():
        feats, rnn_out = policy.build_features(obs, h0, prev, nd, None)
        logits = policy.action_distribution.raw_logits(feats)
    A = logits.shape[-1]
    loss, dlogits = ops.ce_iw_loss(logits.detach().view(T, N, A).contiguous(), tgt.contiguous(), w.to(torch.float32).contiguous(), 1.0)
    roots, grads = [logits], [dlogits.view(T * N, A)]
    if len(AuxLosses) > 0:
        with torch.enable_grad():
            aux = AuxLosses.reduce((w > 0).view(-1))
        roots.append(aux)
        grads.append(torch.ones((), device=dev))
    torch.autograd.backward(roots, grads)
    return loss


def timeit(f, n=10):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / n


for _ in range(3):
    leg.once()
print(f"whole eager update (update_agent): {timeit(leg.once):.2f} ms")
for name, a in (("side stream on", True), ("one stream", False)):
    _train.OVERLAP_INSTRUCTION = a
    for _ in range(2):
        region(); leg.opt.zero_grad()
    print(f"eager forward + loss + backward, {name}: {timeit(lambda: (region(), leg.opt.zero_grad())):.2f} ms")
_train.OVERLAP_INSTRUCTION = False
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        region(); leg.opt.zero_grad()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
ref = None
with torch.cuda.stream(s):
    region()
    ref = leg.opt.grad.clone(); leg.opt.zero_grad()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g, stream=s):
        loss = region()
except Exception as e:  # noqa: BLE001
    print("capture failed:", type(e).__name__, str(e)[:600])
    sys.exit(0)
leg.opt.zero_grad()
g.replay()
torch.cuda.synchronize()
got = leg.opt.grad.clone()
print("replayed gradients equal the eager ones:", bool(torch.equal(got, ref)), "max diff", float((got - ref).abs().max()), "of", float(ref.abs().max()))
leg.opt.zero_grad()
print(f"replayed forward + loss + backward: {timeit(lambda: (g.replay(), leg.opt.zero_grad())):.2f} ms")
