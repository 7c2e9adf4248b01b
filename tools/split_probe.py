"""Where the split-graph step spends its time: event timestamps at the end of each of the three graphs.
python tools/split_probe.py [pred [envs]]  (pred: BASELINE configs[2], RedNet labels, 8 envs)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from bench import gen_observations, make_policy  # noqa: E402
from ivln_ce_amd.graphed import GraphedRollout  # noqa: E402
from ivln_ce_amd.obs_transforms import GTSemanticsIterativeMapper, PredictedSemanticsIterativeMapper  # noqa: E402

pred = len(sys.argv) > 1 and sys.argv[1] == "pred"
B = int(sys.argv[2]) if len(sys.argv) > 2 else (8 if pred else 4)
dev = torch.device("cuda:0")
cfg, policy = make_policy(dev)
tr = (PredictedSemanticsIterativeMapper if pred else GTSemanticsIterativeMapper).from_config(cfg)
obs = [{k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in o.items()} for o in gen_observations(B, 40, 1, with_rgb=pred)]
r = GraphedRollout(policy, [tr], obs[0], deterministic=True, streams="split")
for i in range(20):
    r.step(obs[i % 40])
torch.cuda.synchronize()
E = lambda: torch.cuda.Event(enable_timing=True)  # noqa: E731
acc = [0.0, 0.0, 0.0, 0.0]
n = 100
for i in range(n):
    r.load(obs[i % 40])
    main = torch.cuda.current_stream()
    ev = {k: E() for k in ("t0", "gA_start", "gA_end", "gB1_end", "end")}
    ev["t0"].record(main)
    r._replay_split(mark=lambda name, stream: ev[name].record(stream))  # (the runner's own replay order, with timing marks)
    ev["end"].record(main)
    r.phase ^= 1
    torch.cuda.synchronize()
    for j, k in enumerate(("gA_start", "gA_end", "gB1_end", "end")):
        acc[j] += ev["t0"].elapsed_time(ev[k])
print("per step (us), one step in flight at a time: gA start %.1f | gA end %.1f | gB1 end %.1f | step end %.1f"
      % tuple(1e3 * a / n for a in acc))


def alone(fn, stream, n=100):
    torch.cuda.synchronize()
    a, b = E(), E()
    with torch.cuda.stream(stream):
        a.record(stream)
        for _ in range(n):
            fn()
        b.record(stream)
    torch.cuda.synchronize()
    return 1e3 * a.elapsed_time(b) / n


print("alone (us): gA %.1f | gB1 %.1f | gB2 %.1f" % (alone(lambda: [g.replay() for g in r.gA_parts], r.sA), alone(lambda: [g.replay() for g in r.gB1[0]], torch.cuda.current_stream()),
                                                   alone(r.graphs[0].replay, torch.cuda.current_stream())))
