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


def add_batched_data_to_observations(observations: List[Dict], batched_data, batched_data_key: str):
    if batched_data is not None:
        for i in range(len(observations)):
            observations[i][batched_data_key] = batched_data[i]
    return observations
