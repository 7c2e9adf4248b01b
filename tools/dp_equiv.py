"""Data-parallel equivalence of the DAgger update.  The reference trains in ONE process
(base_il_trainer.py:211-215: backward, optimizer.step); this repo's N-rank update is `update_agent(..., world=N)` =
per-rank forward / backward on the rank's trajectories, ONE sum all-reduce of the flat gradient bucket, Adam with 1/N
folded in.  Two ranks on one GPU over gloo (RCCL refuses two ranks per device; the collective is the same call):

    rank r      takes trajectories r::2 of a seeded batch of 8, runs one update with world=2
    rank 0      then rebuilds the policy and runs the SINGLE-PROCESS form of the same step: update_agent on shard 0 and on
                shard 1 with step_grad=False (the gradients accumulate in the bucket), FlatAdam.step(world=2, allreduce=False)

and asserts: both replicas bit-identical after the data-parallel step; data-parallel parameters == single-process
parameters to 1e-6 on every element whose summed gradient is at least 1e-6 in magnitude (Adam divides by |g| + 1e-8:
below that bar the step is a function of the last bits of g, see tests/test_gpu_train.py) and within lr everywhere.
Per-rank by design: BatchNorm batch statistics and running buffers (each rank normalises its own 4
trajectories, like the two shard passes of the single-process form) and the progress-monitor (TN, TN) aux term (quirk
Q7: a masked mean over the rank's own rows).

  IVLN_DIST_BACKEND=gloo IVLN_ONE_DEVICE=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
      --master-addr 127.0.0.1 --master-port 29571 tools/dp_equiv.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch  # noqa: E402

import ivln_ce_amd  # noqa: E402,F401
from ivln_ce_amd import dist as D  # noqa: E402
from ivln_ce_amd.aux_losses import AuxLosses  # noqa: E402
from ivln_ce_amd.trainers import FlatAdam, update_agent  # noqa: E402
from ivln_ce_amd.utils import dedupe_instructions  # noqa: E402

rank, _, world = D.init()
assert world == 2, "run with two ranks"
dev = torch.device("cuda:0")
from test_gpu_policy import make_policy  # noqa: E402  (det_fill weights: identical on every rank)

T, N, LR = 10, 8, 2.5e-4
g = torch.Generator().manual_seed(77)
lens = [10, 7, 10, 4, 9, 10, 6, 10]
instr = torch.zeros(N, 200)
for n in range(N):
    L = 20 + 5 * n
    instr[n, :L] = torch.randint(2, 2504, (L,), generator=g).float()
full = {"depth_features": torch.randn(T, N, 128, 4, 4, generator=g), "occupancy_map": (torch.rand(T, N, 64, 64, generator=g) < 0.3).float(),
        "semantic_map": torch.randint(0, 13, (T, N, 64, 64), generator=g).float(), "instruction": instr.repeat(T, 1).view(T, N, 200),
        "progress": torch.rand(T, N, 1, generator=g)}
prev = torch.randint(0, 4, (T, N, 1), generator=g)
nd = torch.ones(T, N, 1, dtype=torch.uint8)
nd[0] = 0
tgt = torch.randint(0, 4, (T, N), generator=g)
w = torch.where(torch.rand(T, N, generator=g) < 0.4, torch.tensor(3.2), torch.tensor(1.0))
for n, L in enumerate(lens):
    w[L:, n] = 0
    tgt[L:, n] = 0


def shard(r):
    """Trajectories r::2 as a time-major (T * 4, ...) batch on the device."""
    sel = list(range(r, N, 2))
    obs = {k: v[:, sel].reshape(T * len(sel), *v.shape[2:]).contiguous() for k, v in full.items()}
    obs = {k: v.to(dev) for k, v in dedupe_instructions(obs).items()}
    return (obs, prev[:, sel].reshape(-1, 1).to(dev), nd[:, sel].reshape(-1, 1).to(dev), tgt[:, sel].contiguous().to(dev),
            w[:, sel].contiguous().to(dev))


AuxLosses.activate()
pol = make_policy(use_pm=True, train=True)
opt = FlatAdam(pol, lr=LR)
before = opt.flat.detach().clone()
loss = update_agent(pol, opt, *shard(rank), hidden_size=512, world=world)[0]
sums = D.gather_objects((float(opt.flat.double().sum().item()), float(opt.flat.double().abs().sum().item())))
assert sums[0] == sums[1], f"replicas differ after the all-reduced step: {sums}"
if rank == 0:
    pol1 = make_policy(use_pm=True, train=True)
    opt1 = FlatAdam(pol1, lr=LR)
    assert torch.equal(opt1.flat, before)
    l0 = update_agent(pol1, opt1, *shard(0), hidden_size=512, step_grad=False)[0]
    l1 = update_agent(pol1, opt1, *shard(1), hidden_size=512, step_grad=False)[0]
    gsum = opt1.grad.detach().clone()
    opt1.step(world=2, allreduce=False)
    assert abs(l0 - loss) < 1e-6, (l0, loss)  # rank 0's own shard: the same forward
    d = (opt.flat - opt1.flat).abs()
    firm = gsum.abs() >= 1e-6
    moved = (opt.flat != before)
    e_firm, e_all = float(d[firm].max()), float(d.max())
    same = float((d == 0).double().mean())
    print(f"dp_equiv: replicas identical; data-parallel vs single-process accumulate: max |diff| {e_firm:.2e} on the "
          f"{int(firm.sum())} elements with |grad| >= 1e-6, {e_all:.2e} anywhere, {100 * same:.2f} % bit-identical, "
          f"{int(moved.sum())} of {opt.numel} elements moved")
    assert int(moved.sum()) > 0.5 * opt.numel
    assert e_firm <= 1e-6, e_firm
    assert e_all <= LR * 1.001, e_all
    # per-rank by design: BatchNorm running buffers follow the rank's own shard (single process: shard 0, then shard 1)
    bn = {k: b for k, b in pol.named_buffers() if k.endswith("running_mean")}
    bn1 = dict(pol1.named_buffers())
    assert any(not torch.equal(b, bn1[k]) for k, b in bn.items()), "expected per-rank BatchNorm buffers to differ from the two-pass form"
    print("dist smoke ok: dp_equiv world", world)
AuxLosses.deactivate()
torch.distributed.barrier()
torch.distributed.destroy_process_group()
