"""Tour-ordered trajectory batches for `IterativeDaggerTrainer` (behaviour of
ivlnce_baselines/trainers/tour_dataset.py:20-289): batch row i walks through the episodes of whole tours in
order, so the recurrent state carried from one update to the next IS the tour memory of that row.

`to_constant_bin_number` restates the one function the reference takes from the `binpacking` package
(benmaier/binpacking, not in this image -> parity unpinned): greedy multiway number partitioning, heaviest item
first, always into the currently lightest bin.
"""
from typing import Dict, List, Sequence, Set

import numpy as np
import torch


def to_constant_bin_number(weights: Dict, n_bins: int) -> List[Dict]:
    """Stable sort by decreasing weight; every item joins the bin with the smallest running sum (lowest index on
    ties).  The package's growing volume bound only delays a placement - it never changes which bin is chosen."""
    bins: List[Dict] = [{} for _ in range(n_bins)]
    load = [0.0] * n_bins
    for key in sorted(weights, key=lambda k: -weights[k]):
        b = load.index(min(load))
        bins[b][key] = weights[key]
        load[b] += weights[key]
    return bins


def pad_time(t: torch.Tensor, n: int, fill) -> torch.Tensor:
    """(T, ...) -> (n, ...): rows T..n-1 hold `fill` (cast to t's dtype)."""
    if t.size(0) == n:
        return t
    return torch.cat([t, t.new_full((n - t.size(0), *t.shape[1:]), fill)], dim=0)


def time_major(seqs: Sequence[torch.Tensor], n: int, fill) -> torch.Tensor:
    """N trajectories (T_i, ...) -> (n, N, ...) padded with `fill`."""
    return torch.stack([pad_time(s, n, fill) for s in seqs], dim=1)


def tour_collate(samples):
    """(obs, prev_actions, expert_actions, weights, tour_mask) per trajectory -> the time-major update batch
    (tour_dataset.py:20-104): observations padded with 1.0 and flattened to (T*N, ...); actions / weights padded
    with 0; tour masks padded with 1; episode masks 0 on the first row block only."""
    obs, prev, expert, w, tour = zip(*samples)
    T = max(p.size(0) for p in prev)
    batch_obs = {k: time_major([o[k] for o in obs], T, 1.0).flatten(0, 1) for k in obs[0]}
    expert_tn = time_major(expert, T, 0)
    episode = torch.ones_like(expert_tn, dtype=torch.uint8)
    episode[0] = 0
    tour_tn = time_major(tour, T, 1).to(torch.uint8)
    return (batch_obs, time_major(prev, T, 0).view(-1, 1), episode.view(-1, 1), tour_tn.view(-1, 1), expert_tn,
            time_major(w, T, 0))


class TourSampler(torch.utils.data.Sampler):
    """Batch sampler (tour_dataset.py:107-205).  Tours are packed into `batch_size` bins of near-equal episode
    count; bin i is the episode sequence of batch row i (episodes of a tour shuffled among themselves when
    `shuffle`); batch j = the j-th episode of every bin.  `tour_done_idxs` = the first episode of every tour,
    where that row's tour memory resets.  `drop_last` reproduces the reference's cut: with the first short batch
    at position s (or s = number of batches when none is short) the batches 0..s-2 are kept."""

    def __init__(self, tours_to_idx: Dict, batch_size: int = 1, shuffle: bool = True, drop_last: bool = True,
                 logger=None) -> None:
        assert batch_size <= len(tours_to_idx)
        rows: List[List[int]] = []
        self.tour_done_idxs: Set[int] = set()
        for packed in to_constant_bin_number({k: len(v) for k, v in tours_to_idx.items()}, batch_size):
            row: List[int] = []
            for tour in packed:
                episodes = tours_to_idx[tour]
                if shuffle:
                    np.random.shuffle(episodes)  # in place, like the reference: the caller's lists are reordered
                self.tour_done_idxs.add(episodes[0])
                row += episodes
            rows.append(row)
        steps = max(len(r) for r in rows)
        batches = [[r[j] for r in rows if j < len(r)] for j in range(steps)]
        if drop_last:
            short = next((j for j, b in enumerate(batches) if len(b) < batch_size), steps)
            batches = batches[:short - 1]
        self.batched_idxs = batches
        self._next = 0
        if logger is not None:
            total = sum(len(v) for v in tours_to_idx.values())
            kept = sum(len(b) for b in batches)
            logger.info(f"TourSampler: {len(tours_to_idx)} tours, {total} episodes -> {len(batches)} batches, "
                        f"{total - kept} episodes dropped")

    def get_num_batches(self) -> int:
        return len(self.batched_idxs)

    def get_tour_done_idxs(self) -> Set[int]:
        return self.tour_done_idxs

    def truncate(self, n: int) -> None:
        """Data-parallel ranks hold different tours: all keep the smallest batch count, so every update's
        gradient all-reduce has a partner on every rank."""
        del self.batched_idxs[n:]

    def __len__(self) -> int:
        return len(self.batched_idxs)

    def __iter__(self):
        return self

    def __next__(self) -> List[int]:  # single pass, like the reference's sampler
        if self._next >= len(self.batched_idxs):
            raise StopIteration
        self._next += 1
        return self.batched_idxs[self._next - 1]


class TourTrajectoryDataset(torch.utils.data.Dataset):
    """Map-style view of a TrajectoryStore (tour_dataset.py:208-289): record idx -> (obs, prev_actions,
    expert_actions, inflection weights, tour mask); the tour mask is 0 on the first step of an episode that opens
    a tour and 1 elsewhere."""

    def __init__(self, store, use_iw: bool, inflection_weight_coef: float = 1.0):
        super().__init__()
        self.store = store
        self.tour_done_idxs = None
        self.inflec_weights = torch.tensor([1.0, inflection_weight_coef if use_iw else 1.0])

    def set_tour_done_idxs(self, tour_done_idxs: Set[int]) -> None:
        self.tour_done_idxs = tour_done_idxs

    def __len__(self):
        return len(self.store)

    def __getitem__(self, idx):
        assert self.tour_done_idxs is not None, "Call set_tour_done_idxs to set tour_done_idxs first."
        obs, prev, expert = self.store.get(idx)
        obs = {k: torch.from_numpy(np.array(v)) for k, v in obs.items()}
        prev, expert = torch.from_numpy(np.array(prev)), torch.from_numpy(np.array(expert))
        turn = torch.ones_like(expert)  # the first step counts as an inflection
        turn[1:] = (expert[1:] != expert[:-1]).long()
        tour_mask = torch.ones_like(prev)
        tour_mask[0] = int(idx not in self.tour_done_idxs)
        return obs, prev, expert, self.inflec_weights[turn], tour_mask
