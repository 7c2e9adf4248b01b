"""Where does a graph-replayed rollout step spend its time: host enqueue (obs copies, graph launch) or GPU?
python tools/host_overhead.py [envs]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from bench import gen_observations, make_policy  # noqa: E402
from ivln_ce_amd.graphed import GraphedRollout  # noqa: E402
from ivln_ce_amd.obs_transforms import GTSemanticsIterativeMapper  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device("cuda:0")
cfg, policy = make_policy(dev)
tr = GTSemanticsIterativeMapper.from_config(cfg)
obs = [{k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in o.items()} for o in gen_observations(B, 40, 1)]
r = GraphedRollout(policy, [tr], obs[0], deterministic=True, streams=False)
for i in range(20):
    r.step(obs[i % 40])
torch.cuda.synchronize()


def run(fn, n=200):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        fn(i)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return 1e6 * (t1 - t0) / n, 1e6 * (t2 - t0) / n


def replay_only(i):
    r.graphs[r.phase].replay()
    r.phase ^= 1


print("envs", B)
print("load only        host/step %.1f us   total/step %.1f us" % run(lambda i: r.load(obs[i % 40])))
print("replay only      host/step %.1f us   total/step %.1f us" % run(replay_only))
print("load + replay    host/step %.1f us   total/step %.1f us" % run(lambda i: r.step(obs[i % 40])))
tr2 = GTSemanticsIterativeMapper.from_config(cfg)
r2 = GraphedRollout(policy, [tr2], obs[0], deterministic=True, streams="split")
for i in range(20):
    r2.step(obs[i % 40])
print("split replay (3 graphs, 2 streams): host/step %.1f us   total/step %.1f us" % run(lambda i: r2.step(obs[i % 40])))
