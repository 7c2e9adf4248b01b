"""Host-side (Python) profile of the DAgger update step: where the enqueue time of one update_agent call goes.
python tools/update_host_profile.py [iters]"""
import cProfile
import os
import pstats
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402

dev = torch.device("cuda:0")
torch.cuda.set_device(0)
cfg, policy = bench.make_policy(dev)
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20


def barrier():
    torch.cuda.synchronize()


bench.bench_update(policy, dev, 1, barrier, iters=2, warm=2)  # warm-up: caches, workspaces
pr = cProfile.Profile()
pr.enable()
el, info = bench.bench_update(policy, dev, 1, barrier, iters=iters, warm=0)
pr.disable()
print(f"{1e3 * el / iters:.2f} ms per update under the profiler")
pstats.Stats(pr).sort_stats("tottime").print_stats(30)
