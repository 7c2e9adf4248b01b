"""Which torch (aten) operators still launch kernels / copies inside one DAgger update, and from which line of this
package: one update under torch.profiler (CPU activities with stacks; the GPU kernels each op launches are counted from
the CUDA activities).  python tools/update_torch_ops.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

from bench import UpdateLeg, make_policy  # noqa: E402
from ivln_ce_amd.aux_losses import AuxLosses  # noqa: E402

dev = torch.device("cuda:0")
cfg, policy = make_policy(dev)
ul = UpdateLeg(policy, dev, 1)
AuxLosses.activate()
for _ in range(3):
    ul.once()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    ul.once()
    torch.cuda.synchronize()
AuxLosses.deactivate()
rows = {}
for ev in prof.events():
    if not ev.name.startswith("aten::") or ev.device_type != torch.autograd.DeviceType.CPU:
        continue
    kern = [k for k in ev.kernels] if hasattr(ev, "kernels") else []
    if not kern:
        continue
    where = next((f for f in (ev.stack or []) if "ivln" in f or "bench.py" in f), "?")
    key = (ev.name, where.strip()[-110:], str(ev.input_shapes)[:60])
    r = rows.setdefault(key, [0, 0.0, set()])
    r[0] += 1
    r[1] += sum(k.duration for k in kern)
    r[2].update(k.name[:50] for k in kern)
print(f"{'calls':>5} {'gpu us':>8}  op | call site | shapes | kernels")
for (name, where, shp), (n, us, ks) in sorted(rows.items(), key=lambda kv: -kv[1][0]):
    print(f"{n:5d} {us:8.1f}  {name} | {where} | {shp} | {', '.join(sorted(ks))}")
