"""DAgger trainers with the reference's registry names and surface
(`dagger`, `iterative_collection_dagger`; `__init__(config)`, `train()`, `eval()`,
`_eval_checkpoint(checkpoint_path, writer, checkpoint_index)`; checkpoint dict of
base_il_trainer.py:158-168; result files `stats_ckpt_{i}_{split}.json` / `dtw_data_ckpt_...`).

  rollout + aggregation   ivlnce_baselines/trainers/dagger_trainer.py:251-504,
                          iterative_collection_dagger_trainer.py:131-397
  collate / IW dataset    dagger_trainer.py:42-234
  update                  ivlnce_baselines/common/base_il_trainer.py:173-219 (`update_agent`, all HIP)
  eval                    base_il_trainer.py:313-583 (episodic) / :585-928 (iterative, t-nDTW)

MI355X-native differences: policy + mapper run on HIP; the optimizer is one fused Adam kernel over
a flat fp32 parameter bucket; data-parallel training = one process per GPU, envs and trajectory
batches sharded per rank, ONE RCCL all-reduce of the flat gradient bucket per update (dist.py).
The lmdb/msgpack trajectory database is replaced by `TrajectoryStore` (same record content).
"""
import json
import os
import random
import time
from collections import defaultdict
from typing import Dict, List, Optional

import numpy as np
import torch

from . import depth_net
from . import dist as D
from . import ops
from .aux_losses import AuxLosses
from .envs import construct_envs
from .obs_transforms import apply_obs_transforms_batch, apply_obs_transforms_obs_space, get_active_obs_transforms
from .registry import baseline_registry
from .tour_ndtw import compute_tour_ndtw
from .utils import (add_batched_data_to_observations, batch_obs, batch_to, dedupe_instructions,
                    extract_instruction_tokens, trim_instruction_padding)


# ------------------------------------------------------------------------------------------------
# optimizer: fused Adam over a flat bucket (params and grads become views of two flat tensors)
# ------------------------------------------------------------------------------------------------
class FlatAdam:
    """torch.optim.Adam(policy.parameters(), lr) semantics (base_il_trainer.py:78-94) on one flat
    fp32 bucket: p.data / p.grad are views, the step is ONE kernel, the data-parallel gradient
    exchange is ONE all-reduce.  `sem_lr` gives `net.map_encoder.*` its own learning rate
    (MODEL.SEMANTIC_MAP_ENCODER.custom_lr)."""

    def __init__(self, policy, lr=2.5e-4, betas=(0.9, 0.999), eps=1e-8, sem_lr: Optional[float] = None):
        self.all_names = [k for k, _ in policy.named_parameters()]
        named = [(k, p) for k, p in policy.named_parameters() if p.requires_grad]
        self.names = [k for k, _ in named]
        self.params = [p for _, p in named]
        dev = self.params[0].device
        self.offsets, n = [], 0
        for p in self.params:
            self.offsets.append(n)
            n += p.numel()
        self.numel = n
        self.flat = torch.empty(n, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(n, dtype=torch.float32, device=dev)
        for p, o in zip(self.params, self.offsets):
            self.flat[o:o + p.numel()].copy_(p.data.reshape(-1))
            p.data = self.flat[o:o + p.numel()].view_as(p)
            p.grad = self.grad[o:o + p.numel()].view_as(p)
        self.lr, self.betas, self.eps, self.step_count = lr, betas, eps, 0
        self.seg_of = self.seg_lr = None
        self.sem_lr = sem_lr
        if sem_lr is not None:
            seg = torch.zeros(n, dtype=torch.int32)
            for k, p, o in zip(self.names, self.params, self.offsets):
                if k.startswith("net.map_encoder"):
                    seg[o:o + p.numel()] = 1
            self.seg_of = seg.to(dev)
            self.seg_lr = torch.tensor([lr, sem_lr], dtype=torch.float32, device=dev)

    def step(self, world: int = 1, allreduce: bool = True, guard=None):
        """all-reduce (sum) the flat grads when world > 1, then Adam with the 1/world mean folded in;
        the same kernel zeroes the grads (optimizer.zero_grad()).  `allreduce=False`: the bucket already holds the sum
        over the `world` shards (one process that accumulated them - the reference form the data-parallel step is
        tested against, tools/dp_equiv.py)."""
        if world > 1 and allreduce:
            D.allreduce_sum_(self.grad)
        self.step_count += 1
        ops.WEIGHT_EPOCH += 1  # invalidates folded-BN caches (parameters change through raw pointers)
        ops.invalidate_step_caches()  # ... and the per-episode instruction encodings a replayed graph would keep serving
        if self.flat.is_cuda:
            ops.adam_step(self.flat, self.grad, self.exp_avg, self.exp_avg_sq, self.lr, self.step_count,
                          self.betas[0], self.betas[1], self.eps, self.seg_of, self.seg_lr, 1.0 / world, True, guard=guard)
        else:  # CPU path exists only for the gloo multi-process tests of the bucket/all-reduce logic
            raise RuntimeError("FlatAdam.step needs the HIP library (no CPU fallback)")

    def zero_grad(self):
        self.grad.zero_()

    def undo_skipped_step(self):
        """The last `step` was skipped on the device (its guard word was set: the gradients were void): take the step count
        back and drop the void gradients, so that the update can be run again."""
        self.step_count -= 1
        self.grad.zero_()

    # -- checkpoint format: torch.optim.Adam's own -----------------------------------------------------------------
    def _torch_order(self):
        """Parameter names in the index order of the reference's optimizer (base_il_trainer.py:78-94):
        `Adam(policy.parameters())` numbers EVERY parameter, frozen ones included (they just never get state); with
        SEMANTIC_MAP_ENCODER.custom_lr the map encoder's parameters form group 0 and come first."""
        if self.seg_of is None:
            return [self.all_names], self.all_names
        sem = [k for k in self.all_names if k.startswith("net.map_encoder")]
        reg = [k for k in self.all_names if not k.startswith("net.map_encoder")]
        return [sem, reg], sem + reg

    def state_dict(self):
        """`torch.optim.Adam.state_dict()` of the reference's optimizer over the same policy - the "optim_state" of a
        checkpoint either implementation can resume from (base_il_trainer.py:98-106, 158-168)."""
        groups, order = self._torch_order()
        index = {k: i for i, k in enumerate(order)}
        state = {}
        for k, p, o in zip(self.names, self.params, self.offsets):
            n = p.numel()
            state[index[k]] = {"step": torch.tensor(float(self.step_count)),
                               "exp_avg": self.exp_avg[o:o + n].view_as(p).cpu().clone(),
                               "exp_avg_sq": self.exp_avg_sq[o:o + n].view_as(p).cpu().clone()}
        lrs = [self.lr] if self.seg_of is None else [float(self.sem_lr), self.lr]
        pg = [{"lr": lr, "betas": tuple(self.betas), "eps": self.eps, "weight_decay": 0, "amsgrad": False,
               "params": [index[k] for k in names]} for lr, names in zip(lrs, groups)]
        return {"state": state, "param_groups": pg}

    def load_state_dict(self, sd):
        """Accepts torch.optim.Adam's format (a reference checkpoint, or one of ours) and the flat format this class
        wrote in earlier rounds.  Moments land in the flat buckets by parameter NAME through the reference's index
        order; one shared step count (the reference steps every parameter together)."""
        if "param_groups" not in sd:  # rounds 1-2: {"step", "exp_avg", "exp_avg_sq", "names", "offsets", "lr"}
            self.step_count = sd["step"]
            self.exp_avg.copy_(sd["exp_avg"])
            self.exp_avg_sq.copy_(sd["exp_avg_sq"])
            return
        _, order = self._torch_order()
        if sum(len(g["params"]) for g in sd["param_groups"]) != len(order):
            raise ValueError(f"optimizer state numbers {sum(len(g['params']) for g in sd['param_groups'])} parameters, "
                             f"the policy has {len(order)}")
        where = {k: (p, o) for k, p, o in zip(self.names, self.params, self.offsets)}
        steps = set()
        for i, st in sd["state"].items():
            k = order[int(i)]
            if k not in where:
                raise ValueError(f"optimizer state for `{k}`, which does not train here")
            p, o = where[k]
            if tuple(st["exp_avg"].shape) != tuple(p.shape):
                raise ValueError(f"optimizer state of `{k}` has shape {tuple(st['exp_avg'].shape)}, expected {tuple(p.shape)}")
            self.exp_avg[o:o + p.numel()].copy_(st["exp_avg"].reshape(-1))
            self.exp_avg_sq[o:o + p.numel()].copy_(st["exp_avg_sq"].reshape(-1))
            steps.add(int(st["step"]))
        if len(steps) > 1:
            raise ValueError(f"per-parameter step counts differ ({sorted(steps)}): not a state of one Adam over the policy")
        self.step_count = steps.pop() if steps else 0


_scalars = {}


def _scalar(value, device):
    """Cached 0-d device tensor (the loss scale handed to autograd.backward: no H2D copy per update)."""
    key = (float(value), str(device))
    t = _scalars.get(key)
    if t is None:
        t = _scalars[key] = torch.tensor(float(value), device=device)
    return t


# Opt-in (IVLN_EARLY_LOSS=1): read the two loss values back as soon as the forward is done and let the host run ahead of
# backward + Adam.  In a process that only trains it takes the update from 12.2 to 11.8 ms (the GPU never idles behind
# `.item()`); in a process that has REPLAYED HIPGRAPHS before (every DAgger run: collection, then updates) the same
# pipelined stream runs 25 % slower - 15.4 ms, ~13 us more per launch, on any stream, with or without the side stream,
# the persistent GRU or more hardware queues (measured, round 3) - so the default stays the reference's `.item()` after Adam.
EARLY_LOSS_READBACK = os.environ.get("IVLN_EARLY_LOSS", "0") != "0"
_loss_pins = {}


def _loss_pin(dev):
    """(pinned float32[2], event) per device for update_agent's loss read-back."""
    k = str(dev)
    if k not in _loss_pins:
        _loss_pins[k] = (torch.zeros(2, dtype=torch.float32).pin_memory(), torch.cuda.Event())
    return _loss_pins[k]


def merge_episodic_shards(gathered):
    """Rank 0's view of an episodic evaluation: `gathered` = one (stats_episodes, (agent tours, ground-truth tours)) per
    rank in rank order (dist.gather_objects).  Scenes / envs are sharded by index modulo world (env_utils.py:77-99), so
    episode ids and tour ids never repeat across ranks: plain dictionary unions."""
    stats_episodes, agent_paths, gt_paths = {}, {}, {}
    for s, (a, g) in gathered:
        stats_episodes.update(s)
        agent_paths.update(a)
        gt_paths.update(g)
    return stats_episodes, agent_paths, gt_paths


def merge_iterative_shards(gathered):
    """The same for the iterative evaluation: one (stats per tour, dtw_data per tour, ground-truth tours) per rank.  A
    rank runs whole tours (tour memory never crosses ranks), so a tour's records come from one rank, in step order."""
    stats_tours, dtw_data, gt_paths = {}, {}, {}
    for st, dd, gt in gathered:
        for tour, eps in st.items():
            stats_tours.setdefault(tour, {}).update(eps)
        for tour, pts in dd.items():
            dtw_data.setdefault(tour, []).extend(pts)
        gt_paths.update(gt)
    return stats_tours, dtw_data, gt_paths


def update_agent(policy, optimizer: FlatAdam, observations, prev_actions, not_done_masks, corrected_actions, weights,
                 hidden_size=512, step_grad=True, loss_accumulation_scalar=1, world=1, tour_not_done_masks=None,
                 rnn_states=None):
    """BaseVLNCETrainer._update_agent (base_il_trainer.py:173-219): forward over all T*N rows, weighted
    CE (+ aux), backward, Adam - every arithmetic step on HIP.  Returns python floats like the
    reference (loss, action_loss, aux_loss).  With `rnn_states` (IterativeDaggerTrainer._update_agent,
    iterative_dagger_trainer.py:33-94) the recurrent state carried from the previous batch of the same tours
    seeds the forward (no gradient through it) and the new state is returned as a fourth value."""
    with ops.eager_work_stream(corrected_actions.device):  # ~2000 eager launches: on a stream without graph launches
        return _update_agent(policy, optimizer, observations, prev_actions, not_done_masks, corrected_actions, weights,
                             hidden_size, step_grad, loss_accumulation_scalar, world, tour_not_done_masks, rnn_states)


def _update_agent(policy, optimizer, observations, prev_actions, not_done_masks, corrected_actions, weights, hidden_size,
                  step_grad, loss_accumulation_scalar, world, tour_not_done_masks, rnn_states):
    T, N = corrected_actions.size()
    dev = corrected_actions.device
    carry = rnn_states is not None
    h0 = rnn_states.detach() if carry else torch.zeros(N, policy.net.num_recurrent_layers, hidden_size, device=dev)
    AuxLosses.clear()
    with torch.enable_grad():
        feats, rnn_out = policy.build_features(observations, h0, prev_actions, not_done_masks, tour_not_done_masks)
        logits = policy.action_distribution.raw_logits(feats)  # (T*N, A)
    A = logits.shape[-1]
    scale = 1.0 / loss_accumulation_scalar
    action_loss, dlogits = ops.ce_iw_loss(logits.detach().view(T, N, A).contiguous(), corrected_actions.contiguous(),
                                          weights.to(torch.float32).contiguous(), scale)
    roots, grads = [logits], [dlogits.view(T * N, A)]
    aux_loss = 0.0
    if AuxLosses.is_active() and len(AuxLosses) > 0:
        with torch.enable_grad():
            aux_loss = AuxLosses.reduce((weights > 0).view(-1))  # the reference's own reduction
        roots.append(aux_loss)
        grads.append(_scalar(scale, dev))
    if dev.type == "cuda" and EARLY_LOSS_READBACK:
        # The two loss values exist once the FORWARD is done: their copies to pinned host memory are queued here, in
        # front of the backward pass, and the function ends by waiting for those copies only.  The host goes on to the
        # next update's prologue while the GPU finishes backward + Adam (the next update's kernels queue behind them on
        # the same stream), instead of the GPU idling through that prologue behind a `.item()` that waited for
        # everything (0.3 ms per update, more on a busy host).  Same return values as the reference's `.item()` calls.
        pin, ev = _loss_pin(dev)
        pin[0:1].copy_(action_loss.detach().reshape(1), non_blocking=True)
        if isinstance(aux_loss, torch.Tensor):
            pin[1:2].copy_(aux_loss.detach().reshape(1), non_blocking=True)
        ev.record()
        torch.autograd.backward(roots, grads)
        if step_grad:
            optimizer.step(world)
        ev.synchronize()
        al = float(pin[0])
        ax = float(pin[1]) if isinstance(aux_loss, torch.Tensor) else float(aux_loss)
        ops.seq_sync_poll()  # a timed-out persistent GRU raises here, at the latest one update late (sticky flag)
    else:
        torch.autograd.backward(roots, grads)
        # single rank: the Adam kernel skips the step ON THE DEVICE when this update's persistent GRU timed out (void
        # gradients); the host sees the sticky word after the loss read-back below and runs the update again on the
        # per-timestep launches.  (Several ranks: a rank-local skip would split the replicas - the error is raised.)
        # (not with gradient accumulation: undoing the skipped step clears the whole bucket, earlier micro-batches included -
        #  there the sticky word is an error like on several ranks: ADVICE r5)
        guard = ops.seq_guard_word(dev) if (dev.type == "cuda" and world == 1 and step_grad and loss_accumulation_scalar == 1) else None
        if step_grad:
            optimizer.step(world, guard=guard)
        al = float(action_loss.item())
        ax = float(aux_loss.item()) if isinstance(aux_loss, torch.Tensor) else float(aux_loss)
        if dev.type == "cuda":
            if guard is not None and int(guard) != 0:  # (one 4-byte read on a stream the loss read-back has just drained)
                ops.seq_recover()
                optimizer.undo_skipped_step()
                # the second forward must not count as a second batch: train-mode BatchNorm already folded THIS batch's
                # statistics into its running buffers (they do not depend on the GRUs) - momentum 0 leaves them as they are
                bns = [m for m in policy.modules() if isinstance(m, torch.nn.modules.batchnorm._BatchNorm) and m.training]
                saved = [(m.momentum, None if m.num_batches_tracked is None else m.num_batches_tracked.clone()) for m in bns]
                for m in bns:
                    m.momentum = 0.0
                try:
                    return _update_agent(policy, optimizer, observations, prev_actions, not_done_masks, corrected_actions, weights,
                                         hidden_size, step_grad, loss_accumulation_scalar, world, tour_not_done_masks, rnn_states)
                finally:
                    for m, (mom, nbt) in zip(bns, saved):
                        m.momentum = mom
                        if nbt is not None:
                            m.num_batches_tracked.copy_(nbt)
            ops.check_seq_sync()  # the stream was just synchronised by .item(): a timed-out persistent GRU is an error
    if carry:
        return (al + ax) * scale, al, ax, rnn_out.detach()
    return (al + ax) * scale, al, ax


# ------------------------------------------------------------------------------------------------
# trajectory store + inflection-weighted dataset + collate
# ------------------------------------------------------------------------------------------------
class TrajectoryStore:
    """Replaces the lmdb/msgpack_numpy database (dagger_trainer.py:349-381): record i =
    [obs{key: (T,...)}, prev_actions i64 (T,), oracle_actions i64 (T,)] stored as `<dir>/<i>.npz`."""

    def __init__(self, path):
        self.path = path
        os.makedirs(path, exist_ok=True)

    def __len__(self):
        return len([f for f in os.listdir(self.path) if f.endswith(".npz")])

    def entries(self):
        """lmdb's `stat()["entries"]`: every key, the tour table under "0" included."""
        return len(self) + int(os.path.exists(os.path.join(self.path, "0.json")))

    def clear(self):
        for f in os.listdir(self.path):
            if f.endswith(".npz") or f == "0.json":
                os.remove(os.path.join(self.path, f))

    def put(self, idx, traj_obs: Dict[str, np.ndarray], prev_actions, oracle_actions, tour_id=None):
        d = {"obs/" + k: v for k, v in traj_obs.items()}
        d["prev_actions"] = np.asarray(prev_actions, np.int64)
        d["oracle_actions"] = np.asarray(oracle_actions, np.int64)
        if tour_id is not None:
            d["tour_id"] = np.asarray(str(tour_id))
        np.savez(os.path.join(self.path, f"{idx}.npz"), **d)

    def get(self, idx):
        with np.load(os.path.join(self.path, f"{idx}.npz")) as f:
            obs = {k[4:]: f[k] for k in f.files if k.startswith("obs/")}
            return obs, f["prev_actions"], f["oracle_actions"]

    # the reference keeps {tour_id: [record indices]} as JSON under lmdb key "0" and numbers the trajectories
    # from 1 (iterative_collection_dagger_trainer.py:228-235, 377-385); here it is `0.json` next to 1.npz, 2.npz ...
    def put_tour_index(self, tours_to_idxs):
        with open(os.path.join(self.path, "0.json"), "w") as f:
            json.dump(tours_to_idxs, f)

    def get_tour_index(self):
        path = os.path.join(self.path, "0.json")
        if not os.path.exists(path):
            return {}
        with open(path) as f:
            return json.load(f)


def collate_fn(batch):
    """Trajectories -> one padded time-major batch: what dagger_trainer.py:42-117 returns (observations (T*N, ...) padded
    with 1, previous / corrected actions and weights padded with 0, a not-done mask that is 0 on the first step only),
    built the way the H2D copy wants it: every output is ONE (T, N, ...) slab allocated once - in pinned host memory when
    a GPU is present, so PrefetchLoader's asynchronous copy reads it in place - pre-filled with its pad value, and each
    trajectory is slice-copied into its column.  No per-trajectory pad tensors, no cat / stack intermediates, no second
    pinning copy.  Pinned by tests/golden/collate_golden.json (values and dtypes)."""
    n = len(batch)
    lengths = [int(item[1].shape[0]) for item in batch]
    T = max(lengths)
    pinned = torch.cuda.is_available()

    def slab(like, fill):
        return torch.full((T, n) + tuple(like.shape[1:]), fill, dtype=like.dtype, pin_memory=pinned)

    first_obs, first_prev, first_corr, first_w = batch[0]
    obs = {k: slab(v, 1) for k, v in first_obs.items()}
    prev, corr, weights = slab(first_prev, 0), slab(first_corr, 0), slab(first_w, 0)
    for col, (o, p, c, w) in enumerate(batch):
        L = lengths[col]
        for k, dst in obs.items():
            dst[:L, col] = o[k]
        prev[:L, col], corr[:L, col], weights[:L, col] = p, c, w
    not_done = torch.ones(corr.shape, dtype=torch.uint8, pin_memory=pinned)
    not_done[0] = 0
    flat = {k: v.view((T * n,) + tuple(v.shape[2:])) for k, v in obs.items()}
    return flat, prev.view(-1, 1), not_done.view(-1, 1), corr, weights


def _block_shuffle(lst, block_size):
    """Consecutive runs of `block_size` elements in a shuffled order (the runs stay intact; dagger_trainer.py:120-125's
    contract incl. its draw sequence: ONE `random.shuffle` over as many items as there are runs)."""
    starts = list(range(0, len(lst), block_size))
    random.shuffle(starts)
    out = []
    for s0 in starts:
        out.extend(lst[s0:s0 + block_size])
    return out


class IWTrajectoryDataset(torch.utils.data.IterableDataset):
    """dagger_trainer.py:127-234 over a TrajectoryStore: length-sorted block shuffle, inflection
    weights [1, coef][a_t != a_{t-1}] with the first step an inflection.  `rank/world` shard the
    record indices for data-parallel training."""

    def __init__(self, store: TrajectoryStore, use_iw, inflection_weight_coef=1.0, batch_size=1, rank=0, world=1):
        super().__init__()
        self.store = store
        self.preload_size = batch_size * 100
        self._preload = []
        self.batch_size = batch_size
        self.inflec_weights = torch.tensor([1.0, inflection_weight_coef if use_iw else 1.0])
        n = len(store)
        self.indices = [i for i in range(n) if i % world == rank]
        self.length = len(self.indices)

    def _load_next(self):
        if len(self._preload) == 0:
            if len(self.load_ordering) == 0:
                raise StopIteration
            new_preload, lengths = [], []
            for _ in range(self.preload_size):
                if len(self.load_ordering) == 0:
                    break
                new_preload.append(self.store.get(self.load_ordering.pop()))
                # reference quirk, reproduced (dagger_trainer.py:165): the "length" it sorts a preload by is
                # len(record[0]) = the number of observation KEYS, identical for every record, so the order within
                # a preload is the random `sort_priority` alone (pinned by tests/golden/collate_golden.json)
                lengths.append(len(new_preload[-1][0]))
            sort_priority = list(range(len(lengths)))
            random.shuffle(sort_priority)
            order = sorted(range(len(lengths)), key=lambda k: (lengths[k], sort_priority[k]))
            for idx in _block_shuffle(order, self.batch_size):
                self._preload.append(new_preload[idx])
        return self._preload.pop()

    def __next__(self):
        obs, prev_actions, oracle_actions = self._load_next()
        obs = {k: torch.from_numpy(np.copy(v)) for k, v in obs.items()}
        prev_actions = torch.from_numpy(np.copy(prev_actions))
        oracle_actions = torch.from_numpy(np.copy(oracle_actions))
        inflections = torch.cat([torch.tensor([1], dtype=torch.long), (oracle_actions[1:] != oracle_actions[:-1]).long()])
        return obs, prev_actions, oracle_actions, self.inflec_weights[inflections]

    def __iter__(self):
        self.load_ordering = list(reversed(_block_shuffle(list(self.indices), self.preload_size)))
        return self


# ------------------------------------------------------------------------------------------------
# trainers
# ------------------------------------------------------------------------------------------------
class BaseVLNCETrainer:
    """common/base_il_trainer.py:46-311 + the BaseILTrainer bits of Appendix D."""

    supported_tasks: List[str] = ["VLN-v0"]

    def __init__(self, config=None):
        self.config = config
        self.rank, self.local_rank, self.world = D.world_info()
        self.policy = None
        self.optimizer = None
        if not torch.cuda.is_available():
            raise RuntimeError("the HIP hot path needs a GPU (no CPU fallback)")
        self.device = torch.device("cuda", self.local_rank if self.world > 1 else config.TORCH_GPU_ID)
        self.obs_transforms = []
        self.start_epoch = 0
        self.start_dagger_it = 0
        self.step_id = 0
        self.flush_secs = 30

    # -- setup ---------------------------------------------------------------------------------
    def _make_ckpt_dir(self):
        os.makedirs(self.config.CHECKPOINT_FOLDER, exist_ok=True)

    def _make_results_dir(self):
        os.makedirs(self.config.RESULTS_DIR, exist_ok=True)

    def _get_spaces(self, config, envs=None):
        observation_space = envs.observation_spaces[0]
        action_space = envs.action_spaces[0]
        self.obs_transforms = get_active_obs_transforms(self.config)
        observation_space = apply_obs_transforms_obs_space(observation_space, self.obs_transforms)
        return observation_space, action_space

    def _initialize_policy(self, config, load_from_ckpt, observation_space, action_space):
        from . import latent_policy as _latent  # noqa: F401  (registers LatentCMAPolicy)
        from . import policy as _policy  # noqa: F401  (registers MapCMAPolicy)

        policy_cls = baseline_registry.get_policy(self.config.MODEL.policy_name)
        self.policy = policy_cls.from_config(config=config, observation_space=observation_space,
                                             action_space=action_space)
        self.policy.to(self.device)
        sem = config.MODEL.SEMANTIC_MAP_ENCODER
        self.optimizer = FlatAdam(self.policy, lr=config.IL.lr, sem_lr=sem.lr if sem.custom_lr else None)
        if self.world > 1:  # identical initial weights on every rank
            D.broadcast_(self.optimizer.flat, src=0)
        if load_from_ckpt:
            ckpt = self.load_checkpoint(config.IL.ckpt_to_load, map_location="cpu")
            self.policy.load_state_dict(ckpt["state_dict"])
            if config.IL.is_requeue:
                self.optimizer.load_state_dict(ckpt["optim_state"])
                self.start_epoch = ckpt["epoch"] + 1
                self.start_dagger_it = ckpt.get("dagger_it", 0)
                self.step_id = ckpt["step_id"]

    def save_checkpoint(self, file_name, dagger_it=0, epoch=0, step_id=0):
        """base_il_trainer.py:143-168 (same keys)."""
        if torch.cuda.is_available():
            ops.check_seq_sync()  # (blocking: the checkpoint waits for the GPU anyway)
        if self.rank != 0:
            return
        torch.save(
            {"state_dict": self.policy.state_dict(), "config": self.config, "optim_state": self.optimizer.state_dict(),
             "dagger_it": dagger_it, "epoch": epoch, "step_id": step_id},
            os.path.join(self.config.CHECKPOINT_FOLDER, file_name),
        )

    def load_checkpoint(self, checkpoint_path, *args, **kwargs):
        """`torch.load` of a trainer checkpoint (base_il_trainer.py:170-171).  A checkpoint written by the reference
        pickles its "config" entry as `habitat.config.default.Config` (a yacs CfgNode); where habitat is not installed
        that name resolves to this package's `Config` for the duration of the load, so `pred_it.pth` & co. open."""
        from .config import habitat_config_unpickle_shim

        kwargs.setdefault("weights_only", False)
        with habitat_config_unpickle_shim():
            return torch.load(checkpoint_path, *args, **kwargs)

    def _update_agent(self, observations, prev_actions, not_done_masks, corrected_actions, weights, step_grad=True,
                      loss_accumulation_scalar=1):
        return update_agent(self.policy, self.optimizer, observations, prev_actions, not_done_masks, corrected_actions,
                            weights, self.config.MODEL.STATE_ENCODER.hidden_size, step_grad, loss_accumulation_scalar,
                            self.world)

    @staticmethod
    def _pause_envs(envs_to_pause, envs, recurrent_hidden_states, not_done_masks, prev_actions, batch, rgb_frames=None):
        """base_il_trainer.py:221-256."""
        if len(envs_to_pause) > 0:
            state_index = list(range(envs.num_envs))
            for idx in reversed(envs_to_pause):
                state_index.pop(idx)
                envs.pause_at(idx)
            recurrent_hidden_states = recurrent_hidden_states[state_index]
            not_done_masks = not_done_masks[state_index]
            prev_actions = prev_actions[state_index]
            for k, v in batch.items():
                batch[k] = v[state_index] if torch.is_tensor(v) else [v[i] for i in state_index]
        return envs, recurrent_hidden_states, not_done_masks, prev_actions, batch, rgb_frames

    def _persistent_guard(self, runner, redo_eager):
        """Call right after the stream synchronisation of a policy step.  None when nothing happened (the normal case: one
        read of a pinned host word per plan).  When a persistent kernel of the step timed out - its workgroups were not all
        resident - the failed plans are cleared and retired (depth_net.recover_all: the rest of the run uses the launch
        chain, a warning is logged) and the step's (actions, rnn_states) are computed again from the same inputs:
        `runner.redo_last_step_eagerly()` for a replayed step (the caller then drops the runner), `redo_eager()` else."""
        if self.device.type != "cuda" or not depth_net.any_failed():
            return None
        depth_net.recover_all()
        # the launch path changed (the persistent form is retired): the next capture warms the new path up on its capture
        # streams first - without stepping the mapper, whose world cloud belongs to the rollout in progress (_make_runner)
        self._rewarm_next_capture = True
        if runner is not None:
            actions = runner.redo_last_step_eagerly()
            out = (actions, runner.rnn_states)
        else:
            out = redo_eager()
        torch.cuda.current_stream().synchronize()
        return out

    def _check_mappers(self):
        """Raise if a mapper's device-side error flag is set (IVLN_E_KEYSPACE / IVLN_E_CAPACITY: points were
        dropped, so the maps of this rollout are wrong).  Called where the loops synchronise anyway."""
        for t in self.obs_transforms:
            mm = getattr(t, "mapping_module", None)
            if mm is not None and getattr(mm, "_h", None) is not None:
                mm.check_status()
        depth_net.check_all()  # sticky error word of the persistent depth encoder (a cluster barrier timed out)

    # -- observations -> batch ----------------------------------------------------------------
    def _batch(self, observations, not_done_masks, transform=True):
        observations = extract_instruction_tokens(observations, self.config.TASK_CONFIG.TASK.INSTRUCTION_SENSOR_UUID)
        observations = add_batched_data_to_observations(observations, not_done_masks, "not_done_masks")
        batch = batch_obs(observations, self.device)
        return observations, (apply_obs_transforms_batch(batch, self.obs_transforms) if transform else batch)

    def _graph_eligible(self, sampled=False):
        """The step can be captured when actions are deterministic - or sampled on the device from host uniforms
        (`sampled`: the collection loops, see _RolloutStepper) - and every transformer is an iterative mapper (the
        known-map ones read files on episode reset: host work that cannot live in a graph)."""
        cfg = self.config
        if self.device.type != "cuda":
            return False
        if not sampled and not (bool(getattr(cfg.EVAL, "USE_HIP_GRAPH", True)) and not cfg.EVAL.SAMPLE):
            return False
        if len(cfg.VIDEO_OPTION) > 0:  # the *_viz frames are host-side numpy (a D2H copy per step)
            return False
        if not all(type(t).__name__.endswith("IterativeMapper") for t in self.obs_transforms):
            return False
        # MapCMA needs its mapper in the step; the map-free policies (Latent-CMA) are captured as they are
        return len(self.obs_transforms) > 0 or cfg.MODEL.policy_name != "MapCMAPolicy"

    def _map_before_pause(self, batch):
        """Graph replay runs the mapper INSIDE the step, i.e. after a pause has compacted the batch rows - while the
        reference (and the eager loop) mapped the full batch before the pause.  The mapper keys its world clouds by
        batch row and never re-indexes them (quirk Q5), so the first step after a pause would see other maps than the
        reference's.  When envs are about to pause, the replaying loops therefore map the uncompacted batch eagerly,
        take that one policy step eagerly too, and capture again for the smaller batch afterwards - every step then
        equals the eager loop's bit for bit (tests/test_gpu_train.py)."""
        return apply_obs_transforms_batch(batch, self.obs_transforms)

    def _make_runner(self, batch, rnn_states, prev_actions, first, deterministic=True, extra_keys=()):
        """GraphedRollout for the current number of active envs, seeded with the carried state.  The first
        capture warms up by running the step (the mapper is reset afterwards: nothing has been mapped yet
        and the first real step arrives with not_done_masks == 0 anyway); later captures - envs were paused,
        the batch shrank - must not touch the mapper's world cloud, so they capture without executing.
        A policy in train mode (the collection loops: quirk Q6, BatchNorm batch statistics during rollouts) updates
        its running statistics in every executed step: the warm-up's updates are undone."""
        from .graphed import GraphedRollout

        # the three-graph split is MapCMANet's staging; other policies replay one graph on one stream
        split = type(self.policy).__name__ == "MapCMAPolicy"
        rewarm = (not first) and getattr(self, "_rewarm_next_capture", False)
        self._rewarm_next_capture = False
        keep = {k: v.clone() for k, v in self.policy.named_buffers()} if ((first or rewarm) and self.policy.training) else None
        runner = GraphedRollout(self.policy, self.obs_transforms, batch, deterministic=deterministic,
                                streams="split" if split else False, warmup=2 if first else (1 if rewarm else 0),
                                extra_keys=extra_keys, warmup_mapper=not rewarm)
        if first:
            for t in self.obs_transforms:
                if getattr(t, "mapping_module", None) is not None:
                    t.mapping_module.reset()
        if keep is not None:
            with torch.no_grad():
                for k, v in self.policy.named_buffers():
                    v.copy_(keep[k])
        runner.rnn[runner.phase].copy_(rnn_states)
        runner.prev[runner.phase].copy_(prev_actions)
        return runner

    # -- eval -------------------------------------------------------------------------------------
    def eval(self):
        """BaseILTrainer.eval (Appendix D): a checkpoint file, or every ckpt.N.pth of a folder."""
        D.init()  # multi-rank eval without a preceding train(): the gather / broadcast below need the group
        path = self.config.EVAL_CKPT_PATH_DIR
        if os.path.isfile(path):
            return [self._eval_checkpoint(path, None, 0)]
        out = []
        ckpts = sorted((f for f in os.listdir(path) if f.startswith("ckpt.") and f.endswith(".pth")),
                       key=lambda f: int(f.split(".")[1]))
        for i, f in enumerate(ckpts):
            out.append(self._eval_checkpoint(os.path.join(path, f), None, i))
        return out

    def _setup_eval_config(self, checkpoint_config):
        """BaseILTrainer._setup_eval_config (habitat-lab, Appendix D): the checkpoint's config, overlaid with the
        evaluation config and both command-line option lists."""
        config = self.config.clone()
        config.defrost()
        ckpt_opts = list(getattr(checkpoint_config, "CMD_TRAILING_OPTS", []) or [])
        eval_opts = list(getattr(config, "CMD_TRAILING_OPTS", []) or [])
        config.merge_from_other_cfg(checkpoint_config)
        config.merge_from_other_cfg(self.config)
        config.merge_from_list(ckpt_opts)
        config.merge_from_list(eval_opts)
        config.freeze()
        return config

    def _load_eval_policy(self, config, envs, checkpoint_path):
        observation_space, action_space = self._get_spaces(config, envs=envs)
        self._initialize_policy(config, load_from_ckpt=False, observation_space=observation_space,
                                action_space=action_space)
        # (the reference requires the file; evaluating a freshly initialised policy is how the synthetic runs work)
        if checkpoint_path and os.path.exists(checkpoint_path):
            self.policy.load_state_dict(self.load_checkpoint(checkpoint_path, map_location="cpu")["state_dict"])
        self.policy.eval()

    def _eval_checkpoint(self, checkpoint_path, writer=None, checkpoint_index=0, metrics=None):
        """Evaluate one checkpoint (base_il_trainer.py:313-583; dispatches to `_eval_checkpoint_iterative` when
        TASK_CONFIG.ENVIRONMENT.ITERATIVE.ENABLED).  The loop is the reference's, statement for statement - envs
        constructed with auto-reset off, `reset_at` after every finished episode, an env paused once its next episode
        has already been scored - and pinned to a run of the reference's own loop (tests/golden/iterative_golden.json,
        case "episodic").  MI355X-native differences: mapper + `policy.act` replay as captured hipGraphs when the step
        is eligible (`_graph_eligible`); envs are sharded over ranks and rank 0 merges the statistics.  Returns the
        aggregated statistics (the reference returns None)."""
        import contextlib

        D.init()
        if metrics is None:
            metrics = "distance_to_goal success spl ndtw path_length oracle_success steps_taken".split()
        with contextlib.suppress(Exception):  # sometimes the index does not align with the actual ckpt number
            checkpoint_index = int(checkpoint_path.split(".")[-2])
        if checkpoint_index < getattr(self.config.EVAL, "START_FROM", 0):
            return None
        if self.config.EVAL.USE_CKPT_CONFIG:
            config = self._setup_eval_config(self.load_checkpoint(checkpoint_path, map_location="cpu")["config"])
        else:
            config = self.config.clone()
        config.defrost()
        config.TASK_CONFIG.DATASET.SPLIT = config.EVAL.SPLIT
        config.TASK_CONFIG.DATASET.ROLES = ["guide"]
        config.TASK_CONFIG.DATASET.LANGUAGES = config.EVAL.LANGUAGES
        config.TASK_CONFIG.TASK.NDTW.SPLIT = config.EVAL.SPLIT
        config.TASK_CONFIG.ENVIRONMENT.ITERATOR_OPTIONS.SHUFFLE = False
        config.TASK_CONFIG.ENVIRONMENT.ITERATOR_OPTIONS.MAX_SCENE_REPEAT_STEPS = -1
        config.TASK_CONFIG.ENVIRONMENT.ITERATOR_OPTIONS.SHUFFLE_TOURS = False
        config.TASK_CONFIG.ENVIRONMENT.ITERATOR_OPTIONS.SHUFFLE_EPISODES = False
        config.IL.ckpt_to_load = checkpoint_path
        config.use_pbar = False
        if len(config.VIDEO_OPTION) > 0:
            os.makedirs(config.VIDEO_DIR, exist_ok=True)
        config.freeze()
        if config.TASK_CONFIG.ENVIRONMENT.ITERATIVE.ENABLED:
            return self._eval_checkpoint_iterative(config, writer, checkpoint_index)

        split = config.TASK_CONFIG.DATASET.SPLIT
        fname = os.path.join(config.RESULTS_DIR, f"stats_ckpt_{checkpoint_index}_{split}.json")
        if config.EVAL.SAVE_RESULTS:
            self._make_results_dir()
            if os.path.exists(fname):  # "skipping -- evaluation exists."
                return json.load(open(fname))
        envs = construct_envs(config, None, auto_reset_done=False, rank=self.rank, world=self.world, iterative=False)
        self._load_eval_policy(config, envs, checkpoint_path)
        n = envs.num_envs
        rnn_states = torch.zeros(n, self.policy.net.num_recurrent_layers, config.MODEL.STATE_ENCODER.hidden_size,
                                 device=self.device)
        prev_actions = torch.zeros(n, 1, device=self.device, dtype=torch.long)
        not_done_masks = torch.zeros(n, 1, dtype=torch.uint8, device=self.device)
        use_graph = self._graph_eligible()
        observations = envs.reset()
        observations, batch = self._batch(observations, not_done_masks, transform=not use_graph)
        stats_episodes = {}
        episodes_to_eval = sum(envs.number_of_episodes)
        if config.EVAL.EPISODE_COUNT > -1:
            episodes_to_eval = min(config.EVAL.EPISODE_COUNT, episodes_to_eval)
        runner, captured, eager_once = None, False, False
        t0 = time.time()
        while envs.num_envs > 0 and len(stats_episodes) < episodes_to_eval:
            current_episodes = envs.current_episodes()
            if use_graph and not eager_once:  # mapper + policy.act replayed as captured graphs; the state lives in
                if runner is None:            # the runner's buffers
                    runner = self._make_runner(batch, rnn_states, prev_actions, first=not captured)
                    captured = True
                actions = runner.step(batch)
                rnn_states, prev_actions = runner.rnn_states, actions
                if depth_net.armed():  # (a persistent launch is part of the step: its time-out flag is read after the sync)
                    torch.cuda.current_stream().synchronize()
                    fixed = self._persistent_guard(runner, None)
                    if fixed is not None:
                        (actions, rnn_states), prev_actions, runner = fixed, fixed[0], None
            else:
                eager_once = False

                def act_once(rs=rnn_states, pa=prev_actions, b=batch, m=not_done_masks):
                    with torch.no_grad():
                        return self.policy.act(b, rs, pa, m, deterministic=not config.EVAL.SAMPLE)

                actions, rnn_new = act_once()
                if depth_net.armed():
                    torch.cuda.current_stream().synchronize()
                    fixed = self._persistent_guard(None, act_once)
                    if fixed is not None:
                        actions, rnn_new = fixed
                rnn_states = rnn_new
                prev_actions.copy_(actions)
            outputs = envs.step([a[0].item() for a in actions])
            observations, _, dones, infos = [list(x) for x in zip(*outputs)]
            not_done_masks = torch.tensor([[0] if done else [1] for done in dones], dtype=torch.uint8,
                                          device=self.device)
            if any(dones):
                self._check_mappers()  # episode boundary; the stream was just synchronised by the actions' .item()
            for i in range(envs.num_envs):  # reset envs and observations if necessary
                if not dones[i]:
                    continue
                stats_episodes[current_episodes[i].episode_id] = {k: infos[i][k] for k in metrics}
                observations[i] = envs.reset_at(i)[0]
                prev_actions[i] = torch.zeros(1, dtype=torch.long)
            observations, batch = self._batch(observations, not_done_masks, transform=not use_graph)
            next_episodes = envs.current_episodes()
            envs_to_pause = [i for i in range(envs.num_envs) if next_episodes[i].episode_id in stats_episodes]
            if envs_to_pause and use_graph:
                batch, runner, eager_once = self._map_before_pause(batch), None, True
            envs, rnn_states, not_done_masks, prev_actions, batch, _ = self._pause_envs(
                envs_to_pause, envs, rnn_states, not_done_masks, prev_actions, batch)
        self._check_mappers()
        tours = (envs.dtw_data(), envs.gt_paths()) if hasattr(envs, "dtw_data") else ({}, {})
        gathered = D.gather_objects((stats_episodes, tours))
        envs.close()
        if self.rank != 0:
            return None
        stats_episodes, agent_paths, gt_paths = merge_episodic_shards(gathered)
        aggregated_stats = {}
        num_episodes = len(stats_episodes)
        for stat_key in next(iter(stats_episodes.values())).keys():
            aggregated_stats[stat_key] = sum(v[stat_key] for v in stats_episodes.values()) / num_episodes
        if agent_paths:  # (not in the reference's episodic report) the synthetic env logs whole tours: t-nDTW too
            aggregated_stats["t_ndtw"] = compute_tour_ndtw(agent_paths, gt_paths,
                                                           config.TASK_CONFIG.TASK.NDTW.SUCCESS_DISTANCE)
        if config.EVAL.SAVE_RESULTS:
            with open(fname, "w") as f:
                json.dump(aggregated_stats, f, indent=4)
        if writer is not None:
            for k, v in aggregated_stats.items():
                writer.add_scalar(f"eval_{split}_{k}", v, checkpoint_index + 1)
        return dict(aggregated_stats, episodes=num_episodes, eval_seconds=time.time() - t0)

    def _pause_iterative_envs(self, envs_to_pause, envs, recurrent_hidden_states, agent_episode_not_done_masks,
                              sim_episode_not_done_masks, tour_not_done_masks, action_masks, prev_actions, batch,
                              rgb_frames=None):
        """base_il_trainer.py:258-311: drop the paused envs' rows from every per-env tensor and tell the policy
        (`net.delete_batch_idx`, the Latent-CMA tour memory) which rows went away."""
        if len(envs_to_pause) > 0:
            state_index = list(range(envs.num_envs))
            for idx in reversed(envs_to_pause):
                state_index.pop(idx)
                envs.pause_at(idx)
            recurrent_hidden_states = recurrent_hidden_states[state_index]
            agent_episode_not_done_masks = agent_episode_not_done_masks[state_index]
            sim_episode_not_done_masks = sim_episode_not_done_masks[state_index]
            tour_not_done_masks = tour_not_done_masks[state_index]
            action_masks = action_masks[state_index]
            prev_actions = prev_actions[state_index]
            for k, v in batch.items():
                batch[k] = v[state_index] if torch.is_tensor(v) else [v[i] for i in state_index]
            if rgb_frames is not None:
                rgb_frames = [rgb_frames[i] for i in state_index]
            del_idxs_fn = getattr(self.policy.net, "delete_batch_idx", None)
            if callable(del_idxs_fn):
                del_idxs_fn(envs_to_pause)
        return (envs, recurrent_hidden_states, agent_episode_not_done_masks, sim_episode_not_done_masks,
                tour_not_done_masks, action_masks, prev_actions, batch, rgb_frames)

    def masks_to_tensors(self, agent_episode_dones, sim_episode_dones, tour_dones, produce_actions):
        """iterative_collection_dagger_trainer.py:82-114 (the eval loop builds the same four, :715-744): (n, 1) uint8
        agent-episode / sim-episode / tour not-done masks and the action mask."""
        def nd(dones):
            return torch.tensor([[0] if done else [1] for done in dones], dtype=torch.uint8, device=self.device)

        return (nd(agent_episode_dones), nd(sim_episode_dones), nd(tour_dones),
                torch.tensor(produce_actions, dtype=torch.uint8, device=self.device))

    def _eval_checkpoint_iterative(self, config, writer=None, checkpoint_index=0):
        """Tour-by-tour evaluation (base_il_trainer.py:585-928), statement for statement and pinned to runs of the
        reference's own loop (tests/golden/iterative_golden.json, cases "iterative_*"): the 7-tuple step protocol,
        four masks handed to `act_iterative`, maps reset by the agent-episode or the tour mask
        (EVAL.ITERATIVE_MAP_RESET), statistics taken from the step that ends the agent's episode, `dtw_data` from the
        step that ends the sim episode, `reset_at` + mask patch-up, pause once the next episode has been scored;
        reports `iterative_stats_…`, `iterative_all_stats_…`, `dtw_data_…`; t-nDTW against
        EVAL.ITERATIVE_GT_PATHS[split] (or, for the synthetic env, the expert paths it exposes)."""
        import numbers

        if "Iterative" not in config.ENV_NAME:
            config.defrost()
            config.ENV_NAME = config.TASK_CONFIG.ENVIRONMENT.ITERATIVE.ENV_NAME
            config.freeze()
        split = config.TASK_CONFIG.DATASET.SPLIT
        fname = os.path.join(config.RESULTS_DIR, f"iterative_stats_ckpt_{checkpoint_index}_{split}.json")
        if config.EVAL.SAVE_RESULTS:
            self._make_results_dir()
            if os.path.exists(fname):  # "skipping -- evaluation exists."
                return json.load(open(fname))
        assert self.config.EVAL.ITERATIVE_MAP_RESET in ["episodic", "iterative"], "config.EVAL.ITERATIVE_MAP_RESET not valid"
        episodic_maps = self.config.EVAL.ITERATIVE_MAP_RESET == "episodic"
        envs = construct_envs(config, None, auto_reset_done=False, rank=self.rank, world=self.world, iterative=True)
        self._load_eval_policy(config, envs, config.IL.ckpt_to_load)
        n = envs.num_envs
        rnn_states = torch.zeros(n, self.policy.net.num_recurrent_layers, config.MODEL.STATE_ENCODER.hidden_size,
                                 device=self.device)
        prev_actions = torch.zeros(n, 1, device=self.device, dtype=torch.long)
        agent_episode_not_done_masks = torch.zeros(n, 1, dtype=torch.uint8, device=self.device)
        sim_episode_not_done_masks = torch.zeros(n, 1, dtype=torch.uint8, device=self.device)
        tour_not_done_masks = torch.zeros(n, 1, dtype=torch.uint8, device=self.device)
        action_masks = torch.ones(n, 1, dtype=torch.uint8, device=self.device)
        # a captured step carries ONE policy mask (the agent-episode one, all MapCMA reads: quirk Q11) besides the
        # mapper's reset mask; policies with a tour memory read the tour mask too and run eagerly
        tour_policy = bool(config.MODEL.tour_memory or config.MODEL.tour_memory_variant)
        use_graph = self._graph_eligible() and not tour_policy

        def make_batch(observations):
            reset_masks = agent_episode_not_done_masks if episodic_maps else tour_not_done_masks
            if use_graph and not episodic_maps:  # the mapper resets on `not_done_masks`, the policy on its own key
                observations = add_batched_data_to_observations(observations, agent_episode_not_done_masks,
                                                                "episode_not_done_masks")
            return self._batch(observations, reset_masks, transform=not use_graph)

        observations, _, _ = [list(x) for x in zip(*envs.reset())]
        observations, batch = make_batch(observations)
        stats_tours = defaultdict(dict)  # tour id -> episode id -> stats
        dtw_data = defaultdict(list)     # tour id -> positions
        runner, captured, eager_once = None, False, False
        t0 = time.time()
        while envs.num_envs > 0:
            current_episodes = envs.current_episodes()
            if use_graph and not eager_once:
                if runner is None:
                    runner = self._make_runner(batch, rnn_states, prev_actions, first=not captured)
                    captured = True
                actions = runner.step(batch)
                rnn_states, prev_actions = runner.rnn_states, actions
                if depth_net.armed():
                    torch.cuda.current_stream().synchronize()
                    fixed = self._persistent_guard(runner, None)
                    if fixed is not None:
                        (actions, rnn_states), prev_actions, runner = fixed, fixed[0], None
            else:
                eager_once = False

                def act_once(rs=rnn_states, pa=prev_actions, b=batch, m=(agent_episode_not_done_masks, sim_episode_not_done_masks,
                                                                        tour_not_done_masks, action_masks)):
                    with torch.no_grad():
                        return self.policy.act_iterative(b, rs, pa, *m, deterministic=not config.EVAL.SAMPLE)

                actions, rnn_new = act_once()
                if depth_net.armed():
                    torch.cuda.current_stream().synchronize()
                    fixed = self._persistent_guard(None, act_once)
                    if fixed is not None:
                        actions, rnn_new = fixed
                rnn_states = rnn_new
                prev_actions.copy_(actions)
            outputs = envs.step([a[0].item() for a in actions])
            (observations, _, agent_episode_dones, sim_episode_dones, tour_dones, produce_actions,
             infos) = [list(x) for x in zip(*outputs)]
            (agent_episode_not_done_masks, sim_episode_not_done_masks, tour_not_done_masks,
             action_masks) = self.masks_to_tensors(agent_episode_dones, sim_episode_dones, tour_dones, produce_actions)
            if any(agent_episode_dones):
                self._check_mappers()
            for i in range(envs.num_envs):  # reset envs and observations if necessary
                if not agent_episode_dones[i]:
                    continue
                ep_id, tour_id = current_episodes[i].episode_id, current_episodes[i].tour_id
                if ep_id not in stats_tours[tour_id] and len(infos[i]) > 1:  # the step that ended the agent's episode
                    stats_tours[tour_id][ep_id] = {k: v for k, v in infos[i].items() if isinstance(v, numbers.Number)}
                if not sim_episode_dones[i]:
                    continue
                if "dtw_data" in infos[i]:
                    dtw_data[tour_id].extend(infos[i]["dtw_data"])
                observations[i], tour_done, produce_action = envs.reset_at(i)[0]
                tour_not_done_masks[i] = int(not tour_done)
                action_masks[i] = int(produce_action)
                prev_actions[i] = torch.zeros(1, dtype=torch.long)
            observations, batch = make_batch(observations)
            next_episodes = envs.current_episodes()
            envs_to_pause = [i for i in range(envs.num_envs)
                             if sim_episode_dones[i] and next_episodes[i].episode_id in stats_tours[next_episodes[i].tour_id]]
            if envs_to_pause and use_graph:
                batch, runner, eager_once = self._map_before_pause(batch), None, True
            (envs, rnn_states, agent_episode_not_done_masks, sim_episode_not_done_masks, tour_not_done_masks,
             action_masks, prev_actions, batch, _) = self._pause_iterative_envs(
                envs_to_pause, envs, rnn_states, agent_episode_not_done_masks, sim_episode_not_done_masks,
                tour_not_done_masks, action_masks, prev_actions, batch)
        self._check_mappers()
        gt_local = envs.gt_paths() if hasattr(envs, "gt_paths") else {}
        gathered = D.gather_objects((dict(stats_tours), dict(dtw_data), gt_local))
        envs.close()
        if self.rank != 0:
            return None
        stats_tours, dtw_data, gt_paths = merge_iterative_shards(gathered)
        if config.EVAL.SAVE_RESULTS:  # DTW evaluation data and every episode's stats, for further analysis
            with open(os.path.join(config.RESULTS_DIR, f"dtw_data_ckpt_{checkpoint_index}_{split}.json"), "w") as f:
                json.dump(dtw_data, f, indent=2)
            with open(os.path.join(config.RESULTS_DIR, f"iterative_all_stats_ckpt_{checkpoint_index}_{split}.json"), "w") as f:
                json.dump(stats_tours, f, indent=2)
        aggregated_stats = defaultdict(float)
        for stats_episodes in stats_tours.values():
            for stat_key in next(iter(stats_episodes.values())).keys():
                aggregated_stats[stat_key] += sum(v[stat_key] for v in stats_episodes.values())
        episodes_evaluated = sum(len(v) for v in stats_tours.values())
        for stat_key in aggregated_stats:
            aggregated_stats[stat_key] /= episodes_evaluated
        if os.path.exists(config.EVAL.ITERATIVE_GT_PATHS):
            with open(config.EVAL.ITERATIVE_GT_PATHS, "r") as f:
                gt_paths = json.load(f)[split]
        aggregated_stats["tndtw"] = compute_tour_ndtw(
            agent_paths=dtw_data, gt_paths=gt_paths, success_distance=config.TASK_CONFIG.TASK.NDTW.SUCCESS_DISTANCE)
        if config.EVAL.SAVE_RESULTS:
            with open(fname, "w") as f:
                json.dump(aggregated_stats, f, indent=4)
        if writer is not None:
            for k, v in aggregated_stats.items():
                writer.add_scalar(f"eval_{split}_{k}", v, checkpoint_index + 1)
        return dict(aggregated_stats, episodes=episodes_evaluated, eval_seconds=time.time() - t0)

    def inference(self):
        raise NotImplementedError("quirk Q10: no reference trainer defines inference() either")


class PrefetchLoader:
    """Keeps the update loop fed ("so updates are not input-bound"): a worker thread reads
    and collates the next trajectory batches (npz decode, padding), pins them and copies them to the GPU on
    its own stream while the previous update runs; the consumer only waits on the copy's event.  Order is the
    wrapped loader's order (one worker), so runs are reproducible.  Replaces the reference's DataLoader
    workers (dagger_trainer.py:585-596)."""

    def __init__(self, loader, device, depth=2):
        self.loader, self.device, self.depth = loader, device, depth

    def __iter__(self):
        import queue
        import threading

        q = queue.Queue(maxsize=self.depth)
        stop = threading.Event()  # set when the consumer leaves early (break / exception): the worker must not
        #                           sit forever on a full queue holding pinned batches

        def put(item):
            while not stop.is_set():
                try:
                    q.put(item, timeout=0.1)
                    return True
                except queue.Full:
                    continue
            return False

        cuda = self.device.type == "cuda"
        stream = torch.cuda.Stream(self.device) if cuda else None

        def pin(x):
            if isinstance(x, dict):
                return {k: pin(v) for k, v in x.items()}
            return x.pin_memory() if (cuda and torch.is_tensor(x) and not x.is_pinned()) else x  # (collate_fn's slabs already are)

        def work():
            try:
                for batch in self.loader:
                    # host side, before the H2D copy; batch[-2] = corrected actions (T, N)
                    obs_h = dedupe_instructions(trim_instruction_padding(batch[0], first_rows=batch[-2].shape[1]))
                    batch = (obs_h,) + tuple(batch[1:])
                    parts = tuple(pin(b) for b in batch)
                    if len(parts) == 5:  # episodic collate: no tour masks (slot 3 of the 6-tuple the loops unpack)
                        parts = parts[:3] + (None,) + parts[3:]
                    if cuda:
                        with torch.cuda.stream(stream):
                            moved = batch_to(parts, self.device)
                            ev = torch.cuda.Event()
                            ev.record(stream)
                    else:
                        moved, ev = batch_to(parts, self.device), None
                    if not put((moved, ev)):
                        return
                put(None)
            except BaseException as e:  # noqa: BLE001 - surface loader errors in the training thread
                put(e)

        th = threading.Thread(target=work, daemon=True)
        th.start()
        try:
            while True:
                item = q.get()
                if item is None:
                    break
                if isinstance(item, BaseException):
                    raise item
                moved, ev = item
                if ev is not None:
                    torch.cuda.current_stream().wait_event(ev)
                    for v in list(moved[0].values()) + [m for m in moved[1:] if m is not None]:
                        v.record_stream(torch.cuda.current_stream())  # allocated on the copy stream, used on this one
                yield moved
        finally:
            stop.set()
            th.join(timeout=5.0)


class _RolloutStepper:
    """One sampled, expert-mixed policy step of a DAgger collection loop (dagger_trainer.py:416-427, 469-472;
    iterative_collection_dagger_trainer.py:300-348) and the per-step host copies the loops store.

    Two ways to the same values:
      reference statements  `policy.act(...)` (torch's Categorical draw) then `where(rand < beta, expert, a)` and
                            `where(expert == -1, 0, .)` in torch - what the scripted stand-in policies of the
                            host-logic goldens run, and any policy without a device sampler;
      device mixing         (`ILPolicy` on a GPU) the action head's own launch draws by inverse CDF from a host
                            uniform, mixes and applies the -1 rule (ivln_linear_sample_f32).  The step is then a pure
                            function of its inputs: mapper + policy replay as captured hipGraphs
                            (`IL.DAGGER.USE_HIP_GRAPH`, default on), bit-identical to the eager launches for the
                            same uniforms (tests/test_gpu_train.py).
    The beta draw always comes from the default host generator (one `rand` per step, like the reference's
    `rand_like`), the sampling uniforms from a generator of their own, so a seeded run mixes identically either way.
    Per step ONE stream synchronisation: actions, the two maps and the frozen encoder's features are copied to pinned
    host buffers together (the reference's hooks did a blocking `.cpu()` each)."""

    def __init__(self, trainer, beta, expert_uuid, iterative):
        self.tr, self.beta, self.expert_uuid, self.iterative = trainer, float(beta), expert_uuid, iterative
        cfg, pol = trainer.config, trainer.policy
        self.dev = trainer.device
        self.on_gpu = self.dev.type == "cuda"
        self.device_mix = self.on_gpu and hasattr(pol, "U_SAMPLE")
        self.feats, self.hooks = trainer._feature_hooks(to_host=not self.on_gpu)
        self.use_graph = (self.device_mix and bool(getattr(cfg.IL.DAGGER, "USE_HIP_GRAPH", True))
                          and trainer._graph_eligible(sampled=True) and type(pol).__name__ == "MapCMAPolicy")
        self.runner, self.captured, self.eager_once = None, False, False
        self.gen = torch.Generator().manual_seed(int(cfg.TASK_CONFIG.SEED) + 7919 * (trainer.rank + 1))
        self.pinned = {}
        if self.device_mix:
            pol.collect_mix = {"beta": self.beta, "expert_uuid": expert_uuid}

    def close(self):
        for h in self.hooks:
            h.remove()
        if self.device_mix:
            self.tr.policy.collect_mix = None

    def before_pause(self, batch):
        """Envs are about to pause: a replaying loop maps the still uncompacted batch now, steps eagerly once and
        captures again for the smaller batch afterwards (BaseVLNCETrainer._map_before_pause)."""
        if self.use_graph:
            batch = self.tr._map_before_pause(batch)
            self.runner, self.eager_once = None, True
        return batch

    def _to_host(self, name, t):
        """Asynchronous D2H into a pinned buffer (one per name and shape); valid after the step's synchronise."""
        if not self.on_gpu:
            return t
        key = (name, tuple(t.shape), t.dtype)
        buf = self.pinned.get(key)
        if buf is None:
            buf = self.pinned[key] = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
        buf.copy_(t, non_blocking=True)
        return buf

    def step(self, batch, rnn_states, prev_actions, masks):
        """masks: (not_done,) or the four iterative masks.  Returns (final actions (n, 1) i64 on the device - what the
        envs are stepped with and the next step's previous actions -, rnn_states, host dict with `actions` (list of
        int), `occ` / `sem` (numpy or None), `depth` / `rgb` (CPU tensors or None))."""
        tr, pol = self.tr, self.tr.policy
        n = prev_actions.shape[0]
        draw = torch.rand((n, 1), dtype=torch.float)  # default generator: the reference's rand_like(actions)
        if self.device_mix:
            batch[pol.U_SAMPLE] = torch.rand((n, 1), dtype=torch.float, generator=self.gen).to(self.dev)
            batch[pol.U_BETA] = draw.to(self.dev)
        replayed = self.use_graph and not self.eager_once
        if replayed:
            if self.runner is None:
                self.runner = tr._make_runner(batch, rnn_states, prev_actions, first=not self.captured,
                                              deterministic=False, extra_keys=(self.expert_uuid,))
                self.captured = True
            runner = self.runner
            actions = runner.step(batch)
            rnn_states = runner.rnn_states
            occ, sem = runner.maps()
            depth = runner.depth_features() if tr._caches_depth() else None
            rgb = None
        else:
            self.eager_once = False
            rnn_in = rnn_states

            def act_once():
                if self.iterative:
                    a, r = pol.act_iterative(batch, rnn_in, prev_actions, *masks, deterministic=False)
                else:
                    a, r = pol.act(batch, rnn_in, prev_actions, masks[0], deterministic=False)
                if not self.device_mix:  # the reference's statements
                    expert = batch[self.expert_uuid].long()
                    a = torch.where(draw.to(a.device) < self.beta, expert, a)
                    a = torch.where(expert == -1, torch.zeros_like(a), a)
                return a, r

            actions, rnn_states = act_once()
            occ, sem = batch.get("occupancy_map"), batch.get("semantic_map")
            depth, rgb = self.feats.get("depth"), self.feats.get("rgb")
        if (occ is None) != (sem is None):
            raise RuntimeError("either both map keys should exist in the batch or neither")

        def host_copies():
            return {"actions": self._to_host("a", actions), "occ": None if occ is None else self._to_host("o", occ),
                    "sem": None if sem is None else self._to_host("s", sem),
                    "depth": None if depth is None else self._to_host("d", depth),
                    "rgb": None if rgb is None else self._to_host("r", rgb)}

        host = host_copies()
        if self.on_gpu:
            torch.cuda.current_stream().synchronize()
            fixed = tr._persistent_guard(runner if replayed else None, None if replayed else act_once) if depth_net.armed() else None
            if fixed is not None:  # a persistent kernel of this step timed out: the values above are void, these are the step's
                actions, rnn_states = fixed
                if replayed:
                    self.runner = None  # (its graphs contain the retired launch: the next step captures again)
                    depth = self.feats.get("depth") if tr._caches_depth() else None  # (the redo ran eagerly: the encoder's hook kept them)
                else:
                    depth, rgb = self.feats.get("depth"), self.feats.get("rgb")
                host = host_copies()
                torch.cuda.current_stream().synchronize()
        if not replayed:
            prev_actions.copy_(actions)
            actions = prev_actions
        host["actions"] = [int(a) for a in host["actions"].view(-1).tolist()]
        for k in ("occ", "sem"):
            if host[k] is not None:
                host[k] = host[k].numpy().copy()
        for k in ("depth", "rgb"):
            if host[k] is not None:
                host[k] = host[k].clone() if self.on_gpu else host[k]
        return actions, rnn_states, host


@baseline_registry.register_trainer(name="dagger")
class DaggerTrainer(BaseVLNCETrainer):
    def __init__(self, config=None):
        self.lmdb_features_dir = config.IL.DAGGER.lmdb_features_dir.format(split=config.TASK_CONFIG.DATASET.SPLIT)
        super().__init__(config)
        if self.world > 1:
            self.lmdb_features_dir = os.path.join(self.lmdb_features_dir, f"rank{self.rank}")
        self.store = TrajectoryStore(self.lmdb_features_dir)

    def _make_dirs(self):
        self._make_ckpt_dir()
        if self.config.EVAL.SAVE_RESULTS:
            self._make_results_dir()

    # -- pieces shared by the two collection loops -------------------------------------------------------------
    def _compact_host_rows(self):
        """Quirk Q12 (reference bug, reproduced by default).  When envs pause during a beta == 1 collection the
        reference compacts the device-side rows (`_pause_envs`, base_il_trainer.py:221-311) but NOT its host-side
        per-env lists (`observations`, `episodes`: dagger_trainer.py:397-412, iterative_collection_dagger_trainer.py:
        275-298): for one step the surviving envs are paired with a neighbour's stale observation, and their
        partial trajectories shift onto the wrong env for the rest of the collection (the last few records of a
        teacher-forcing pass).  Stored data is part of the parity contract, so the default follows the reference;
        `IL.DAGGER.compact_paused_rows: True` (not a reference key) re-indexes the host lists as well."""
        return bool(getattr(self.config.IL.DAGGER, "compact_paused_rows", False))

    def _caches_depth(self):
        cfg = self.config
        return not cfg.MODEL.DEPTH_ENCODER.trainable and cfg.MODEL.DEPTH_ENCODER.cnn_type == "VlnResnetDepthEncoder"

    def _feature_hooks(self, to_host=True):
        """dagger_trainer.py:301-323 / iterative_collection_dagger_trainer.py:184-210: forward hooks that keep the
        frozen encoders' output of the current step (`feats["depth"]`, `feats["rgb"]`), so that stored trajectories
        carry features instead of frames.  `to_host` False: the device tensor is kept and the stepper copies it to
        pinned memory together with the step's other outputs (one synchronisation per step instead of a blocking
        `.cpu()` per hook)."""
        cfg = self.config
        feats, hooks = {}, []

        def keep(name):
            return lambda m, i, o: feats.__setitem__(name, o.detach().cpu() if to_host else o.detach())

        if self._caches_depth():
            hooks.append(self.policy.net.depth_encoder.visual_encoder.register_forward_hook(keep("depth")))
        if not cfg.MODEL.RGB_ENCODER.trainable and hasattr(self.policy.net, "rgb_encoder"):
            hooks.append(self.policy.net.rgb_encoder.cnn.register_forward_hook(keep("rgb")))
        return feats, hooks

    def _store_episode(self, episode, idx, expert_uuid):
        """`save_episode_to_disk` (iterative_collection_dagger_trainer.py:60-80, dagger_trainer.py:352-371): record
        `idx` = [obs{key: (T, ...)} without the expert sensor, prev_actions i64 (T,), oracle_actions i64 (T,)]."""
        traj_obs = batch_obs([step[0] for step in episode], device=torch.device("cpu"))
        del traj_obs[expert_uuid]
        traj_obs = {k: v.numpy() for k, v in traj_obs.items() if torch.is_tensor(v)}
        if self.config.IL.DAGGER.lmdb_fp16:
            traj_obs = {k: v.astype(np.float16) for k, v in traj_obs.items()}
        self.store.put(idx, traj_obs, [step[1] for step in episode], [step[2] for step in episode],
                       tour_id=episode[0][3])

    def _update_dataset(self, data_it):
        """dagger_trainer.py:251-504: roll the policy out with beta-mixed expert actions, cache the frozen encoders'
        features through forward hooks, store finished trajectories.  Returns the number stored.

        The flow follows the reference statement by statement and is pinned to it by
        tests/golden/rollout_golden.json (the reference's own `_update_dataset` on a scripted env): beta = p^it
        with 0^0 = 0 (:295-299), `where(rand < beta, expert, policy)` (:423-427; the draw comes from the host
        generator so that a seeded run is reproducible whatever the device), expert -1 => step with action 0 and
        drop the episode (:469-472, 352), at beta == 1 envs whose next episode was already collected are paused and
        every per-env state row compacted (:312-316, 392-412), a pass over the envs stores EVERY finished episode
        even past `update_size` (:349-386)."""
        cfg = self.config
        envs = construct_envs(cfg, None, rank=self.rank, world=self.world, iterative=False)
        expert_uuid = cfg.IL.DAGGER.expert_policy_sensor_uuid
        n = envs.num_envs
        rnn_states = torch.zeros(n, self.policy.net.num_recurrent_layers, cfg.MODEL.STATE_ENCODER.hidden_size,
                                 device=self.device)
        prev_actions = torch.zeros(n, 1, device=self.device, dtype=torch.long)
        not_done_masks = torch.zeros(n, 1, dtype=torch.uint8, device=self.device)
        p = cfg.IL.DAGGER.p
        beta = 0.0 if p == 0.0 else p ** data_it
        ensure_unique_episodes = beta == 1.0
        stepper = _RolloutStepper(self, beta, expert_uuid, iterative=False)
        observations = envs.reset()
        observations, batch = self._batch(observations, not_done_masks, transform=not stepper.use_graph)
        episodes = [[] for _ in range(n)]
        skips = [False] * n
        dones = [False] * n
        collected, start_id = 0, len(self.store)
        ep_ids_collected = {ep.episode_id for ep in envs.current_episodes()} if ensure_unique_episodes else None
        target = max(1, cfg.IL.DAGGER.update_size // self.world)
        # host mirrors of two device rows the loop records per step (compacted with them on a pause): the previous
        # actions and the expert's action of the CURRENT batch - no device-to-host copy per step for either
        prev_host = [0] * n
        expert_host = [o[expert_uuid].item() for o in observations]
        with torch.no_grad():
            while collected < target:
                envs_to_pause = []
                current_episodes = envs.current_episodes() if ensure_unique_episodes else None
                for i in range(envs.num_envs):
                    # data-parallel ranks stop exactly at their share (no reference counterpart; a single process
                    # stores every episode that finished in this pass, like the reference)
                    # (an EMPTY list can only be met after a quirk-Q12 shift - the reference dies in batch_obs there)
                    if dones[i] and not skips[i] and episodes[i] and (self.world == 1 or collected < target):
                        self._store_episode(episodes[i], start_id + collected, expert_uuid)
                        collected += 1
                        if ensure_unique_episodes:
                            if current_episodes[i].episode_id in ep_ids_collected:
                                envs_to_pause.append(i)
                            else:
                                ep_ids_collected.add(current_episodes[i].episode_id)
                    if dones[i]:
                        episodes[i] = []
                if ensure_unique_episodes and envs_to_pause:
                    keep = [i for i in range(envs.num_envs) if i not in envs_to_pause]
                    batch = stepper.before_pause(batch)
                    envs, rnn_states, not_done_masks, prev_actions, batch, _ = self._pause_envs(
                        envs_to_pause, envs, rnn_states, not_done_masks, prev_actions, batch)
                    prev_host = [prev_host[i] for i in keep]
                    expert_host = [expert_host[i] for i in keep]
                    if self._compact_host_rows():
                        observations = [observations[i] for i in keep]
                        episodes = [episodes[i] for i in keep]
                    if envs.num_envs == 0:
                        break
                if self.world > 1 and collected >= target:
                    break
                actions, rnn_states, host = stepper.step(batch, rnn_states, prev_actions, (not_done_masks,))
                tours_now = [getattr(e, "tour_id", None) for e in envs.current_episodes()]
                experts = expert_host
                for i in range(envs.num_envs):
                    o = dict(observations[i])
                    if host["rgb"] is not None:
                        o["rgb_features"] = host["rgb"][i]
                        o.pop("rgb", None)
                    if host["depth"] is not None:
                        o["depth_features"] = host["depth"][i]
                        o.pop("depth", None)
                    if host["occ"] is not None:
                        o["occupancy_map"], o["semantic_map"] = host["occ"][i], host["sem"][i]
                        for k in ["semantic", "semantic12", "world_robot_pose", "world_robot_orientation", "env_name",
                                  "rgb"]:
                            o.pop(k, None)
                    episodes[i].append((o, prev_host[i], experts[i], tours_now[i]))
                skips = [int(e) == -1 for e in experts]
                prev_actions = actions
                prev_host = host["actions"]
                outputs = envs.step(host["actions"])
                observations, _, dones, _ = [list(x) for x in zip(*outputs)]
                not_done_masks = torch.tensor([[0] if d else [1] for d in dones], dtype=torch.uint8,
                                              device=self.device)
                observations, batch = self._batch(observations, not_done_masks, transform=not stepper.use_graph)
                expert_host = [o[expert_uuid].item() for o in observations]
        stepper.close()
        envs.close()
        self._check_mappers()
        return collected

    def train(self):
        """dagger_trainer.py:506-649."""
        cfg = self.config
        D.init()
        self._make_dirs()
        # (a requeued run continues on the trajectories it already collected)
        if (not cfg.IL.DAGGER.preload_lmdb_features and cfg.IL.DAGGER.drop_existing_lmdb_features
                and not (cfg.IL.load_from_ckpt and cfg.IL.is_requeue)):
            self.store.clear()
        envs = construct_envs(cfg, None, rank=self.rank, world=self.world)
        observation_space, action_space = self._get_spaces(cfg, envs=envs)
        envs.close()
        self._initialize_policy(cfg, cfg.IL.load_from_ckpt, observation_space, action_space)
        return self._train_loop()

    def _resume_point(self, dagger_it):
        """Requeue (`IL.is_requeue`, base_il_trainer.py:98-106): the checkpoint names the (dagger_it, epoch) it was
        written after.  The run continues with the next epoch of THAT iteration on the trajectories already in the
        store, and every later iteration starts at epoch 0 with a fresh collection.  (The reference restores
        `start_epoch` but its loops never read it - a requeued reference run starts over at iteration 0, epoch 0
        and overwrites its checkpoints; resuming where the checkpoint says is the evident intent.)
        Returns (collect?, first_epoch)."""
        first = getattr(self, "start_dagger_it", 0)
        if dagger_it == first and self.start_epoch > 0:
            return False, self.start_epoch
        return True, 0

    def _train_loop(self):
        cfg = self.config
        log = []
        for dagger_it in range(getattr(self, "start_dagger_it", 0), cfg.IL.DAGGER.iterations):
            step_id = 0
            collect, first_epoch = self._resume_point(dagger_it)
            if collect and not cfg.IL.DAGGER.preload_lmdb_features:
                self._update_dataset(dagger_it + (1 if cfg.IL.load_from_ckpt else 0))
            dataset = IWTrajectoryDataset(self.store, cfg.IL.use_iw, cfg.IL.inflection_weight_coef, cfg.IL.batch_size)
            loader = torch.utils.data.DataLoader(dataset, batch_size=cfg.IL.batch_size, shuffle=False,
                                                 collate_fn=collate_fn, pin_memory=False, drop_last=True, num_workers=0)
            # every rank owns its own store (preloaded / kept features can leave them unequal): all ranks run the
            # MIN batch count, so no rank issues a gradient all-reduce the others never join
            n_batches = D.allreduce_min_int(dataset.length // cfg.IL.batch_size, self.device)
            AuxLosses.activate()  # only around the updates, never during rollouts (dagger_trainer.py:579)
            for epoch in range(first_epoch, cfg.IL.epochs):
                for bi, (obs_b, prev_b, nd_b, _, corr_b, w_b) in enumerate(PrefetchLoader(loader, self.device)):
                    if bi >= n_batches:
                        break
                    loss, action_loss, aux_loss = self._update_agent(obs_b, prev_b, nd_b, corr_b, w_b)
                    log.append({"dagger_it": dagger_it, "epoch": epoch, "step": step_id, "loss": loss,
                                "action_loss": action_loss, "aux_loss": aux_loss})
                    step_id += 1
                    self.step_id += 1
                self.save_checkpoint(f"ckpt.{dagger_it * cfg.IL.epochs + epoch}.pth", dagger_it=dagger_it, epoch=epoch,
                                     step_id=self.step_id)
            AuxLosses.deactivate()
        return log


@baseline_registry.register_trainer(name="iterative_collection_dagger")
class IterativeCollectionDaggerTrainer(DaggerTrainer):
    """iterative_collection_dagger_trainer.py:24-397 - the trainer every MapCMA experiment YAML names: the update is
    DaggerTrainer's, the collection runs tour by tour on the iterative env protocol (7-tuple steps, oracle phases
    between the episodes of a tour).  Maps are reset by the TOUR mask, so they persist across a tour's episodes
    (:166-168, 373-375); the policy gets all four masks through `act_iterative`; only steps in which the agent
    itself acts are stored (:320-321); with `save_tour_idx_data` the {tour: [record ids]} table is kept as record
    "0" and the trajectories are numbered from 1 (:228-235, 377-385).  Pinned to runs of the reference's own
    `_update_dataset` on a scripted iterative env (tests/golden/iterative_golden.json, "collect")."""

    def add_map_to_observations(self, observations, batch, num_envs):
        """:28-58: copy this step's maps into the per-env observation dicts (what gets stored) and drop the keys
        that only served to build them."""
        map_k_sum = int("occupancy_map" in batch) + int("semantic_map" in batch)
        if map_k_sum == 1:
            raise RuntimeError("either both map keys should exist in the batch or neither")
        if map_k_sum != 2:
            return observations
        occ, sem = batch["occupancy_map"].cpu().numpy(), batch["semantic_map"].cpu().numpy()
        for i in range(num_envs):
            observations[i]["occupancy_map"], observations[i]["semantic_map"] = occ[i], sem[i]
            for k in ["semantic", "semantic12", "world_robot_pose", "world_robot_orientation", "env_name"]:
                observations[i].pop(k, None)
        return observations

    def batch_and_transform(self, observations, not_done_masks):
        """:116-129.  `not_done_masks` is what resets the maps: with the tour masks they live for a whole tour."""
        observations, batch = self._batch(observations, not_done_masks)
        return batch, observations

    def _update_dataset(self, data_it, save_tour_idx_data=False):
        cfg = self.config
        envs = construct_envs(cfg, None, rank=self.rank, world=self.world, iterative=True)
        expert_uuid = cfg.IL.DAGGER.expert_policy_sensor_uuid
        n = envs.num_envs
        rnn_states = torch.zeros(n, self.policy.net.num_recurrent_layers, cfg.MODEL.STATE_ENCODER.hidden_size,
                                 device=self.device)
        prev_actions = torch.zeros(n, 1, device=self.device, dtype=torch.long)
        agent_episode_not_done_masks = torch.zeros(n, 1, dtype=torch.uint8, device=self.device)
        sim_episode_not_done_masks = torch.zeros(n, 1, dtype=torch.uint8, device=self.device)
        tour_not_done_masks = torch.zeros(n, 1, dtype=torch.uint8, device=self.device)
        action_masks = torch.ones(n, 1, dtype=torch.uint8, device=self.device)
        p = cfg.IL.DAGGER.p
        beta = 0.0 if p == 0.0 else p ** data_it  # in Python 0.0 ** 0.0 == 1.0, but we want 0.0
        ensure_unique_episodes = beta == 1.0
        stepper = _RolloutStepper(self, beta, expert_uuid, iterative=True)
        use_graph = stepper.use_graph

        def make_batch(observations):
            # the maps are reset by the TOUR masks; a captured step carries the policy's own mask beside them
            if use_graph:
                observations = add_batched_data_to_observations(observations, agent_episode_not_done_masks,
                                                                "episode_not_done_masks")
                observations, batch = self._batch(observations, tour_not_done_masks, transform=False)
                return batch, observations
            return self.batch_and_transform(observations, tour_not_done_masks)

        observations, _, _ = [list(x) for x in zip(*envs.reset())]
        batch, observations = make_batch(observations)
        episodes = [[] for _ in range(n)]
        skips = [False for _ in range(n)]
        sim_episode_dones = [False for _ in range(n)]
        collected_eps = 0
        ep_ids_collected = {ep.episode_id for ep in envs.current_episodes()} if ensure_unique_episodes else None
        # record numbering (:225-235): lmdb's entry count includes the tour table, which claims key "0" the first
        # time a table is asked for
        tours_to_idxs = defaultdict(list)
        start_id = self.store.entries()
        if save_tour_idx_data:
            if start_id:
                tours_to_idxs.update(self.store.get_tour_index())
            else:
                start_id += 1
        target = max(1, cfg.IL.DAGGER.update_size // self.world)
        # host mirrors of the device rows the loop records per step, compacted with them on a pause
        prev_host = [0] * n
        expert_host = [o[expert_uuid].item() for o in observations]
        acting = [True] * n
        with torch.no_grad():
            while collected_eps < target:
                envs_to_pause = [] if ensure_unique_episodes else None
                current_episodes = envs.current_episodes() if ensure_unique_episodes else None
                for i in range(envs.num_envs):  # when a sim episode is done, save it
                    if not sim_episode_dones[i]:
                        continue
                    if skips[i]:
                        episodes[i] = []
                        continue
                    if self.world > 1 and collected_eps >= target:  # data-parallel ranks stop exactly at their share
                        episodes[i] = []
                        continue
                    if not episodes[i]:  # only after a quirk-Q12 shift (the reference dies in batch_obs there)
                        continue
                    idx = start_id + collected_eps
                    self._store_episode(episodes[i], idx, expert_uuid)
                    tours_to_idxs[str(episodes[i][0][3])].append(idx)
                    collected_eps += 1
                    if ensure_unique_episodes:
                        if current_episodes[i].episode_id in ep_ids_collected:
                            envs_to_pause.append(i)
                        else:
                            ep_ids_collected.add(current_episodes[i].episode_id)
                    episodes[i] = []
                if ensure_unique_episodes:
                    if envs_to_pause:
                        keep = [i for i in range(envs.num_envs) if i not in envs_to_pause]
                        prev_host = [prev_host[i] for i in keep]
                        expert_host = [expert_host[i] for i in keep]
                        acting = [acting[i] for i in keep]
                        batch = stepper.before_pause(batch)
                        if self._compact_host_rows():
                            observations = [observations[i] for i in keep]
                            episodes = [episodes[i] for i in keep]
                    (envs, rnn_states, agent_episode_not_done_masks, sim_episode_not_done_masks, tour_not_done_masks,
                     action_masks, prev_actions, batch, _) = self._pause_iterative_envs(
                        envs_to_pause, envs, rnn_states, agent_episode_not_done_masks, sim_episode_not_done_masks,
                        tour_not_done_masks, action_masks, prev_actions, batch)
                    if envs.num_envs == 0:
                        break
                if self.world > 1 and collected_eps >= target:
                    break
                actions, rnn_states, host = stepper.step(
                    batch, rnn_states, prev_actions,
                    (agent_episode_not_done_masks, sim_episode_not_done_masks, tour_not_done_masks, action_masks))
                if host["occ"] is not None:  # add_map_to_observations (:28-58) from the step's host copies
                    for i in range(envs.num_envs):
                        observations[i]["occupancy_map"], observations[i]["semantic_map"] = host["occ"][i], host["sem"][i]
                        for k in ["semantic", "semantic12", "world_robot_pose", "world_robot_orientation", "env_name"]:
                            observations[i].pop(k, None)
                for i, current_episode in enumerate(envs.current_episodes()):
                    if not acting[i]:  # only add steps if the agent is acting: skip oracle phases
                        continue
                    if host["depth"] is not None:
                        observations[i]["depth_features"] = host["depth"][i]
                        del observations[i]["depth"]
                    if host["rgb"] is not None:
                        observations[i]["rgb_features"] = host["rgb"][i]
                    if "rgb" in observations[i]:
                        del observations[i]["rgb"]
                    observations[i].pop("episode_not_done_masks", None)  # (the captured step's extra mask key)
                    episodes[i].append((observations[i], prev_host[i], expert_host[i], current_episode.tour_id))
                skips = [int(e) == -1 for e in expert_host]
                prev_actions = actions
                prev_host = host["actions"]
                outputs = envs.step(host["actions"])
                (observations, _, agent_episode_dones, sim_episode_dones, tour_dones, produce_actions,
                 _) = [list(x) for x in zip(*outputs)]
                (agent_episode_not_done_masks, sim_episode_not_done_masks, tour_not_done_masks,
                 action_masks) = self.masks_to_tensors(agent_episode_dones, sim_episode_dones, tour_dones,
                                                       produce_actions)
                acting = [bool(p) for p in produce_actions]
                batch, observations = make_batch(observations)
                expert_host = [o[expert_uuid].item() for o in observations]
        stepper.close()
        envs.close()
        self._check_mappers()
        if save_tour_idx_data:
            self.store.put_tour_index(tours_to_idxs)
            return dict(tours_to_idxs)
        return collected_eps


@baseline_registry.register_trainer(name="iterative_dagger")
class IterativeDaggerTrainer(IterativeCollectionDaggerTrainer):
    """iterative_dagger_trainer.py:31-283: tour-ordered updates.  Batches come from `TourSampler` (row i of every
    batch continues the tours of bin i), the recurrent state returned by one update seeds the next one
    (detached), and the policy's memory mode decides what of it survives: nothing (episodic policies), everything
    (`tour_memory`) or only the tour-long slot (`tour_memory_variant`)."""

    def _update_agent(self, observations, prev_actions, episode_not_done_masks, tour_not_done_masks,
                      corrected_actions, weights, step_grad=True, loss_accumulation_scalar=1, rnn_states=None):
        mc = self.config.MODEL
        T, N = corrected_actions.size()
        L = self.policy.net.num_recurrent_layers
        reset_memory = not (mc.tour_memory or mc.tour_memory_variant)
        if rnn_states is None or reset_memory:
            rnn_states = torch.zeros(N, L, mc.STATE_ENCODER.hidden_size, device=self.device)
        if mc.tour_memory_variant:  # only the tour-long slot survives a batch boundary (:58-60)
            rnn_states = rnn_states.clone()
            rnn_states[:, : L - 1] = 0
        return update_agent(self.policy, self.optimizer, observations, prev_actions, episode_not_done_masks,
                            corrected_actions, weights, mc.STATE_ENCODER.hidden_size, step_grad,
                            loss_accumulation_scalar, self.world, tour_not_done_masks=tour_not_done_masks,
                            rnn_states=rnn_states)

    def _train_loop(self):
        from .tour_batches import TourSampler, TourTrajectoryDataset, tour_collate

        cfg = self.config
        log = []
        for dagger_it in range(getattr(self, "start_dagger_it", 0), cfg.IL.DAGGER.iterations):
            step_id = 0
            collect, first_epoch = self._resume_point(dagger_it)
            if cfg.IL.DAGGER.preload_lmdb_features or not collect:
                tours_to_idxs = self.store.get_tour_index()
            else:
                tours_to_idxs = self._update_dataset(dagger_it + (1 if cfg.IL.load_from_ckpt else 0),
                                                     save_tour_idx_data=True)
            AuxLosses.activate()
            for epoch in range(first_epoch, cfg.IL.epochs):
                dataset = TourTrajectoryDataset(self.store, cfg.IL.use_iw, cfg.IL.inflection_weight_coef)
                sampler = TourSampler({k: list(v) for k, v in tours_to_idxs.items()}, batch_size=cfg.IL.batch_size,
                                      shuffle=True, drop_last=True)
                if self.world > 1:
                    sampler.truncate(D.allreduce_min_int(len(sampler), self.device))
                dataset.set_tour_done_idxs(sampler.get_tour_done_idxs())
                loader = torch.utils.data.DataLoader(dataset, batch_sampler=sampler, collate_fn=tour_collate,
                                                     pin_memory=False, num_workers=0)
                rnn_states = torch.zeros(cfg.IL.batch_size, self.policy.net.num_recurrent_layers,
                                         cfg.MODEL.STATE_ENCODER.hidden_size, device=self.device)
                for obs_b, prev_b, ep_b, tour_b, corr_b, w_b in PrefetchLoader(loader, self.device):
                    loss, action_loss, aux_loss, rnn_states = self._update_agent(
                        obs_b, prev_b, ep_b, tour_b, corr_b, w_b, rnn_states=rnn_states)
                    log.append({"dagger_it": dagger_it, "epoch": epoch, "step": step_id, "loss": loss,
                                "action_loss": action_loss, "aux_loss": aux_loss})
                    step_id += 1
                    self.step_id += 1
                self.save_checkpoint(f"ckpt.{dagger_it * cfg.IL.epochs + epoch}.pth", dagger_it=dagger_it, epoch=epoch,
                                     step_id=self.step_id)
            AuxLosses.deactivate()
        return log
