"""Golden vectors for the tour-ordered batching of IterativeDaggerTrainer: the REFERENCE's own `TourSampler` and
`collate_fn` (ivlnce_baselines/trainers/tour_dataset.py:20-205) on small hand-made tour tables / trajectories.
`lmdb` and `msgpack_numpy` (only touched by the dataset's __getitem__, not used here) are empty stubs;
`binpacking.to_constant_bin_number` is absent from the image and is stood in for by a greedy restatement written
HERE (heaviest first into the lightest bin) - so the bin assignment itself is NOT pinned by this golden, only what
the reference does with the bins (row order, tour starts, transposition, the drop_last cut) and the collate.
Build container only:  python tests/golden/gen_tour_golden.py -> tests/golden/tour_batches.json
"""
import importlib.util
import json
import os
import sys
import types

import numpy as np
import torch

OUT = os.path.dirname(os.path.abspath(__file__))


def _greedy_bins(d, n):
    keys = sorted(d, key=lambda k: -d[k])
    bins, load = [{} for _ in range(n)], [0] * n
    for k in keys:
        b = load.index(min(load))
        bins[b][k] = d[k]
        load[b] += d[k]
    return bins


sys.modules["binpacking"] = types.SimpleNamespace(to_constant_bin_number=_greedy_bins)
sys.modules["lmdb"] = types.ModuleType("lmdb")
sys.modules["msgpack_numpy"] = types.ModuleType("msgpack_numpy")
spec = importlib.util.spec_from_file_location("ref_tour_dataset", "/root/reference/ivlnce_baselines/trainers/tour_dataset.py")
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)


def tours(sizes):
    out, nxt = {}, 1  # record 0 is the tour table itself
    for t, n in enumerate(sizes):
        out[f"tour{t}"] = list(range(nxt, nxt + n))
        nxt += n
    return out


def sampler_case(sizes, batch_size, shuffle, drop_last, seed):
    np.random.seed(seed)
    table = tours(sizes)
    s = ref.TourSampler({k: list(v) for k, v in table.items()}, batch_size=batch_size, shuffle=shuffle, drop_last=drop_last)
    return {"sizes": sizes, "batch_size": batch_size, "shuffle": shuffle, "drop_last": drop_last, "seed": seed,
            "batches": [list(map(int, b)) for b in s.batched_idxs], "tour_done_idxs": sorted(int(i) for i in s.tour_done_idxs),
            "iterated": [list(map(int, b)) for b in s]}


def collate_case():
    g = torch.Generator().manual_seed(9)
    lens = [4, 2, 5]
    samples = []
    for n_i, T in enumerate(lens):
        obs = {"feat": torch.randn(T, 3, 2, generator=g), "tokens": torch.randint(0, 50, (T, 6), generator=g)}
        prev = torch.randint(0, 4, (T,), generator=g)
        expert = torch.randint(0, 4, (T,), generator=g)
        w = torch.where(torch.rand(T, generator=g) < 0.5, torch.tensor(3.2), torch.tensor(1.0))
        tour = torch.ones(T, dtype=torch.long)
        tour[0] = n_i % 2
        samples.append((obs, prev, expert, w, tour))
    out = ref.collate_fn([tuple({k: v.clone() for k, v in s[0].items()} if i == 0 else s[i].clone() for i in range(5))
                          for s in samples])
    obs_b, prev_b, ep_b, tour_b, corr_b, w_b = out
    ser = lambda t: {"dtype": str(t.dtype), "shape": list(t.shape), "data": t.flatten().tolist()}  # noqa: E731
    return {"samples": [{"obs": {k: ser(v) for k, v in s[0].items()}, "prev": ser(s[1]), "expert": ser(s[2]),
                         "weights": ser(s[3]), "tour": ser(s[4])} for s in samples],
            "out": {"obs": {k: ser(v) for k, v in obs_b.items()}, "prev": ser(prev_b), "episode": ser(ep_b),
                    "tour": ser(tour_b), "expert": ser(corr_b), "weights": ser(w_b)}}


if __name__ == "__main__":
    cases = [
        sampler_case([5, 3, 4, 2, 6, 1, 3], 3, False, True, 0),
        sampler_case([5, 3, 4, 2, 6, 1, 3], 3, True, True, 5),
        sampler_case([4, 4, 4], 3, False, True, 0),     # all batches full: the cut still drops the last one
        sampler_case([4, 4, 4], 3, False, False, 0),
        sampler_case([7, 1, 1, 2], 2, True, True, 11),
        sampler_case([2, 9], 2, False, True, 0),
    ]
    json.dump({"sampler": cases, "collate": collate_case()}, open(os.path.join(OUT, "tour_batches.json"), "w"))
    for c in cases:
        print(c["sizes"], c["batch_size"], c["shuffle"], c["drop_last"], "->", c["batches"], c["tour_done_idxs"])
