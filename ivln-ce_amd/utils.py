"""Glue with the reference's names (ivlnce_baselines/common/utils.py:12-135): observation batching
and host->device moves.  No arithmetic."""
from collections import defaultdict
from typing import Dict, List, Optional, Set, Tuple

import numpy as np
import torch


def extract_instruction_tokens(observations: List[Dict], instruction_sensor_uuid: str, tokens_uuid: str = "tokens"):
    """utils.py:12-35."""
    if instruction_sensor_uuid not in observations[0] or instruction_sensor_uuid == "pointgoal_with_gps_compass":
        return observations
    for i in range(len(observations)):
        if isinstance(observations[i][instruction_sensor_uuid], dict) and tokens_uuid in observations[i][instruction_sensor_uuid]:
            observations[i][instruction_sensor_uuid] = observations[i][instruction_sensor_uuid]["tokens"]
        else:
            break
    return observations


def batch_obs(observations: List[Dict], device: Optional[torch.device] = None,
              ignore_keys: Optional[Set[str]] = None) -> Dict:
    """utils.py:57-92: list of per-env dicts -> dict of stacked device tensors (uint32 -> int32;
    `env_name` kept as a list)."""
    if ignore_keys is None:
        ignore_keys = {"env_name"}
    batch = defaultdict(list)
    for obs in observations:
        for sensor in obs:
            v = obs[sensor]
            if isinstance(v, np.ndarray) and v.dtype == np.uint32:
                v = np.int32(v)
            if sensor not in ignore_keys:
                v = torch.as_tensor(v)
            batch[sensor].append(v)
    out: Dict = {}
    for sensor in batch:
        if sensor not in ignore_keys:
            out[sensor] = torch.stack(batch[sensor], dim=0).to(device)
        else:
            out[sensor] = batch[sensor]
    return out


def batch_to(batch: Tuple, device: torch.device = None, non_blocking: bool = True) -> Tuple:
    """utils.py:95-135: observations cast to float32 on the device, the rest moved as is."""
    (observations_batch, prev_actions_batch, episode_not_done_masks, tour_not_done_mask, corrected_actions_batch,
     weights_batch) = batch
    observations_batch = {
        k: v.to(device=device, dtype=torch.float32, non_blocking=non_blocking) for k, v in observations_batch.items()
    }
    return (
        observations_batch,
        prev_actions_batch.to(device=device, non_blocking=non_blocking),
        episode_not_done_masks.to(device=device, non_blocking=non_blocking),
        tour_not_done_mask.to(device=device, non_blocking=non_blocking) if tour_not_done_mask is not None else None,
        corrected_actions_batch.to(device=device, non_blocking=non_blocking),
        weights_batch.to(device=device, non_blocking=non_blocking),
    )


def trim_instruction_padding(observations: Dict, key: str = "instruction", multiple: int = 8,
                             first_rows: Optional[int] = None) -> Dict:
    """Drop the all-padding tail of a HOST-side token batch (rows, 200) before it is copied to the GPU: the
    reference's packed bi-LSTM / pad_packed_sequence only ever produces the batch's longest instruction
    (instruction_encoder.py:70-92), so every consumer (W_ih GEMM, text_k, the attention axis) sees Lmax columns,
    not 200.  Lmax is rounded up to `multiple` (16-byte loads along the token axis); columns beyond a row's own
    length stay masked exactly as before.  A device tensor is returned unchanged (finding Lmax would be a sync).

    `first_rows` = N for a collated time-major (T*N, 200) batch: only the t = 0 rows are measured.  `collate_fn`
    pads the observations of finished trajectories with 1.0 (dagger_trainer.py:66-70), so a padded timestep row
    counts 200 "tokens" and would disable the trim for every batch of unequal trajectory lengths; the instruction
    of a trajectory is the same at every timestep and t = 0 is never padding.  Padded rows keep their first Lmax
    ones - those timesteps carry zero loss weight, so what the encoder makes of them never reaches a gradient."""
    t = observations.get(key)
    if t is None or not torch.is_tensor(t) or t.is_cuda or t.dim() != 2:
        return observations
    L = t.shape[1]
    probe = t if first_rows is None else t[:first_rows]
    longest = int((probe != 0).sum(dim=1).max().item()) if probe.shape[0] > 0 else L
    keep = min(L, max(multiple, -(-longest // multiple) * multiple))
    if keep < L:
        observations = dict(observations)
        observations[key] = t[:, :keep].contiguous()
    return observations


def dedupe_instructions(observations: Dict, key: str = "instruction") -> Dict:
    """HOST side, update batches: a collated time-major batch repeats each trajectory's token row at every timestep
    (T*N rows, N + 1 distinct ones: the trajectories and collate_fn's all-ones padding row).  Adds
    `instruction_unique` (U, L) and `instruction_index` (T*N,) so that the policy encodes U sequences instead of T*N
    (policy.MapCMANet.forward_hip); the gradient is identical - the per-row gradients are summed onto the shared
    encoding in row order.  Left alone when nothing repeats or the tensor is already on the device."""
    t = observations.get(key)
    if t is None or not torch.is_tensor(t) or t.is_cuda or t.dim() != 2 or t.shape[0] < 2:
        return observations
    uniq, inv = torch.unique(t, dim=0, return_inverse=True)
    if uniq.shape[0] * 2 > t.shape[0]:
        return observations
    observations = dict(observations)
    observations[key + "_unique"] = uniq.contiguous()
    observations[key + "_index"] = inv.to(torch.int32).contiguous()
    return observations


def add_batched_data_to_observations(observations: List[Dict], batched_data, batched_data_key: str):
    if batched_data is not None:
        for i in range(len(observations)):
            observations[i][batched_data_key] = batched_data[i]
    return observations
