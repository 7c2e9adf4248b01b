"""Observation-transformer plugins with the reference's registry names
(ivlnce_baselines/common/obs_transforms.py:30-176): `GTSemanticsIterativeMapper`,
`PredictedSemanticsIterativeMapper`, `GTSemanticsKnownMapper`, `PredictedSemanticsKnownMapper`.
Same constructor, `from_config`, `transform_observation_space` and `forward(dict) -> dict`
behaviour; the map arithmetic runs in the HIP mapper (csrc/mapper.hip)."""
import math
from typing import Dict

import numpy as np
import torch.nn as nn
from torch import Tensor

from .mapping import (
    CameraParameters,
    MapDimensions,
    create_gt_semantics_iterative_mapper,
    create_gt_semantics_known_mapper,
    create_predicted_semantics_iterative_mapper,
    create_predicted_semantics_known_mapper,
    extract_camera_parameters,
    extract_egocentric_map_parameters,
)
from .registry import baseline_registry
from .spaces import Box

try:  # pragma: no cover
    from habitat_baselines.common.obs_transformers import ObservationTransformer  # type: ignore
except Exception:  # noqa: BLE001

    class ObservationTransformer(nn.Module):
        def transform_observation_space(self, observation_space, **kwargs):
            return observation_space

        @classmethod
        def from_config(cls, config):
            raise NotImplementedError

        def forward(self, observations):
            return observations


@baseline_registry.register_obs_transformer()
class Mapper(ObservationTransformer):
    def __init__(self, camera_parameters: CameraParameters, map_dimensions: MapDimensions, visualize=False):
        super().__init__()
        self.camera_parameters = camera_parameters
        self.map_dimensions = map_dimensions
        self.visualize = visualize
        self.mapping_module = None
        self.sizing = {}  # HIP-mapper sizing from the config (table_cells / world_capacity), see from_config
        # obs_transforms.py:46-52: keys deleted after generating the maps
        self.keys_to_delete = ["world_robot_orientation", "world_robot_pose", "semantic", "semantic12", "env_name"]

    def transform_observation_space(self, observation_space):
        r = self.map_dimensions.resolution_meters
        nrows = math.ceil(self.map_dimensions.height_meters / r)
        ncols = math.ceil(self.map_dimensions.width_meters / r)
        for new_key in ["occupancy_map", "semantic_map"]:
            observation_space.spaces[new_key] = Box(low=0, high=255, shape=(nrows, ncols), dtype=np.uint8)
        for key in self.keys_to_delete:
            if key in observation_space.spaces:
                del observation_space.spaces[key]
        return observation_space

    def forward(self, observations: Dict[str, Tensor]) -> Dict[str, Tensor]:
        self.setup_mapping_module(observations)
        observations = self.update_maps_from_observations(observations)
        observations = self.visualize_maps(observations)
        observations = self.delete_extra_information(observations)
        return observations

    def visualize_maps(self, observations):
        """The reference adds `occupancy_map_viz` / `semantic_map_viz` colour frames here when VIDEO_OPTION is set
        (obs_transforms.py:105-113).  Video / visualisation is outside the hot path this package replaces
        (SURVEY.md section 2 row 11), so a request for it fails loudly instead of returning maps without frames."""
        if self.visualize:
            raise NotImplementedError(
                "map visualisation frames (VIDEO_OPTION / visualize=True) are not part of the MI355X hot path; "
                "render `occupancy_map` / `semantic_map` with the reference's visualize_semantic_map.py")
        return observations

    def setup_mapping_module(self, observations: Dict[str, Tensor]):
        raise NotImplementedError

    def update_maps_from_observations(self, observations):
        mem = self.mapping_module(observations)
        observations["occupancy_map"] = mem.occupancy  # aliases of persistent buffers (:101-102)
        observations["semantic_map"] = mem.semantic
        return observations

    # the step in two halves (MappingModule.begin / finish): graphed.GraphedRollout enqueues `begin_maps` beside the network
    # that predicts the labels and `finish_maps` behind it; together they are `forward`
    def begin_maps(self, observations):
        self.setup_mapping_module(observations)
        self.mapping_module.begin(observations)

    def finish_maps(self, observations):
        mem = self.mapping_module.finish(observations)
        observations["occupancy_map"] = mem.occupancy
        observations["semantic_map"] = mem.semantic
        observations = self.visualize_maps(observations)
        return self.delete_extra_information(observations)

    def delete_extra_information(self, observations):
        for key in self.keys_to_delete:
            if key in observations:
                del observations[key]
        return observations

    @classmethod
    def from_config(cls, config, visualize=False):
        camera_parameters = extract_camera_parameters(
            depth_sensor_params=config.TASK_CONFIG.SIMULATOR.DEPTH_SENSOR,
            map_sensor_params=config.RL.POLICY.OBS_TRANSFORMS.EGOCENTRIC_MAPPER,
        )
        dims = extract_egocentric_map_parameters(config.RL.POLICY.OBS_TRANSFORMS.EGOCENTRIC_MAPPER)
        if len(config.VIDEO_OPTION) > 0 or visualize:
            # fail at set-up, not on the first mapper step of a rollout: the *_viz colour frames of the reference
            # (obs_transforms.py:105-113 over visualize_semantic_map.py: cv2 / imutils rendering) are outside the hot
            # path this package replaces
            raise NotImplementedError(
                "VIDEO_OPTION / visualize=True: map visualisation frames are not part of the MI355X hot path; run with "
                "VIDEO_OPTION [] and render `occupancy_map` / `semantic_map` with the reference's visualize_semantic_map.py")
        tr = cls(
            camera_parameters=camera_parameters,
            map_dimensions=dims,
            visualize=(len(config.VIDEO_OPTION) > 0) or visualize,
        )
        mp = config.RL.POLICY.OBS_TRANSFORMS.EGOCENTRIC_MAPPER
        tr.sizing = {k: int(getattr(mp, k)) for k in ("table_cells", "world_capacity") if int(getattr(mp, k, 0) or 0) > 0}
        tr.sizing["b_max"] = 64  # envs per process the mapper is sized for (ivln_mapper_create: <= 64)
        return tr


@baseline_registry.register_obs_transformer()
class GTSemanticsIterativeMapper(Mapper):
    def setup_mapping_module(self, observations):
        if self.mapping_module is None:
            self.mapping_module = create_gt_semantics_iterative_mapper(
                device=observations["depth"].device,
                camera_parameters=self.camera_parameters,
                map_dimensions=self.map_dimensions,
                **self.sizing,
            )


@baseline_registry.register_obs_transformer()
class PredictedSemanticsIterativeMapper(Mapper):
    predicted_semantics = True  # RedNet runs inside the transformer: it, not the policy's depth encoder, bounds the step

    def setup_mapping_module(self, observations):
        if self.mapping_module is None:
            self.mapping_module = create_predicted_semantics_iterative_mapper(
                device=observations["depth"].device,
                camera_parameters=self.camera_parameters,
                map_dimensions=self.map_dimensions,
                **self.sizing,
            )


@baseline_registry.register_obs_transformer()
class GTSemanticsKnownMapper(Mapper):
    def setup_mapping_module(self, observations):
        if self.mapping_module is None:
            self.mapping_module = create_gt_semantics_known_mapper(
                device=observations["depth"].device, map_dimensions=self.map_dimensions, **self.sizing
            )


@baseline_registry.register_obs_transformer()
class PredictedSemanticsKnownMapper(Mapper):
    def setup_mapping_module(self, observations):
        if self.mapping_module is None:
            self.mapping_module = create_predicted_semantics_known_mapper(
                device=observations["depth"].device, map_dimensions=self.map_dimensions, **self.sizing
            )


def get_active_obs_transforms(config):
    """habitat_baselines.common.obs_transformers.get_active_obs_transforms (Appendix D)."""
    out = []
    for name in config.RL.POLICY.OBS_TRANSFORMS.ENABLED_TRANSFORMS:
        cls = baseline_registry.get_obs_transformer(name)
        if cls is None:
            raise ValueError(f"unknown obs transformer {name}")
        out.append(cls.from_config(config))
    return out


def apply_obs_transforms_batch(batch, obs_transforms):
    for t in obs_transforms:
        batch = t(batch)
    return batch


def apply_obs_transforms_obs_space(obs_space, obs_transforms):
    for t in obs_transforms:
        obs_space = t.transform_observation_space(obs_space)
    return obs_space
