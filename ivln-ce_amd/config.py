"""yacs-free experiment config with the reference's key tree.

Mirrors the surface of ``ivlnce_baselines/config/default.py:14-212`` (``_C`` tree and
``get_config(config_paths, opts)``) and the subset of ``habitat_baselines.config.default._C`` /
``habitat_extensions/config/default.py:6-214`` the hot path reads, so the
reference's experiment YAMLs load unchanged through ``run.py --exp-config``.

``Config`` behaves like ``habitat.Config`` (a yacs ``CfgNode`` with ``new_allowed=True``): attribute
access, ``defrost()/freeze()``, ``merge_from_file``, ``merge_from_list``, ``clone``.
"""
import copy
import os
from ast import literal_eval
from typing import List, Optional, Union

import yaml

CONFIG_FILE_SEPARATOR = ","


class Config(dict):
    """Attribute dict standing in for ``habitat.Config`` (yacs CfgNode, new_allowed=True)."""

    _FROZEN = "__frozen__"

    def __init__(self, init=None):
        super().__init__()
        object.__setattr__(self, Config._FROZEN, False)
        if init:
            for k, v in init.items():
                self[k] = Config(v) if isinstance(v, dict) and not isinstance(v, Config) else v

    # attribute access -------------------------------------------------
    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name)

    def __setattr__(self, name, value):
        if self.is_frozen():
            raise AttributeError(
                f"Attempted to set {name} to {value}, but Config is immutable"
            )
        self[name] = value

    def __delattr__(self, name):
        del self[name]

    # yacs API ----------------------------------------------------------
    def is_frozen(self):
        d = object.__getattribute__(self, "__dict__")
        # (an instance unpickled from a yacs CfgNode carries `__immutable__` instead of this class's flag)
        return bool(d.get(Config._FROZEN, d.get("__immutable__", False)))

    def _set_frozen(self, flag):
        object.__setattr__(self, Config._FROZEN, flag)
        object.__getattribute__(self, "__dict__").pop("__immutable__", None)
        for v in self.values():
            if isinstance(v, Config):
                v._set_frozen(flag)

    def defrost(self):
        self._set_frozen(False)

    def freeze(self):
        self._set_frozen(True)

    def clone(self):
        return copy.deepcopy(self)

    def __deepcopy__(self, memo):
        out = Config()
        for k, v in self.items():
            dict.__setitem__(out, k, copy.deepcopy(v, memo))
        object.__setattr__(out, Config._FROZEN, self.is_frozen())
        return out

    def __reduce__(self):  # picklable inside checkpoints ("config" entry)
        return (Config._rebuild, (dict(self), self.is_frozen()))

    @staticmethod
    def _rebuild(d, frozen):
        c = Config(d)
        if frozen:
            c.freeze()
        return c

    def merge_from_other_cfg(self, other):
        for k, v in other.items():
            if isinstance(v, dict) and isinstance(self.get(k), Config):
                self[k].merge_from_other_cfg(v)
            else:
                dict.__setitem__(
                    self, k, Config(v) if isinstance(v, dict) and not isinstance(v, Config) else copy.deepcopy(v)
                )

    def merge_from_file(self, path):
        with open(path, "r") as f:
            data = yaml.safe_load(f) or {}
        self.merge_from_other_cfg(data)

    def merge_from_list(self, opts):
        assert len(opts) % 2 == 0, "opts must be KEY VALUE pairs"
        for full_key, v in zip(opts[0::2], opts[1::2]):
            node = self
            keys = full_key.split(".")
            for k in keys[:-1]:
                if k not in node:
                    dict.__setitem__(node, k, Config())
                node = node[k]
            if isinstance(v, str):
                try:
                    v = literal_eval(v)
                except (ValueError, SyntaxError):
                    pass
            dict.__setitem__(node, keys[-1], v)

    def register_deprecated_key(self, key):
        pass


CN = Config


class habitat_config_unpickle_shim:
    """Context manager: while active, `habitat.config.default.Config` and `yacs.config.CfgNode` - the classes a
    reference checkpoint's "config" entry pickles as (base_il_trainer.py:158-168) - resolve to this package's
    `Config` if the real modules are not importable.  Unpickling a CfgNode sets its instance __dict__
    (`__immutable__` & co.) and its items directly, which a `Config` holds just as well."""

    NAMES = {"habitat": None, "habitat.config": None, "habitat.config.default": "Config", "yacs": None,
             "yacs.config": "CfgNode"}

    def __enter__(self):
        import importlib
        import sys
        import types

        self._made = []
        for name, cls in self.NAMES.items():
            if name in sys.modules:
                continue
            try:
                importlib.import_module(name)
                continue
            except Exception:  # noqa: BLE001 - absent (or broken) package: stand in for it
                pass
            mod = types.ModuleType(name)
            if cls:
                setattr(mod, cls, Config)
            sys.modules[name] = mod
            self._made.append(name)
        return self

    def __exit__(self, *exc):
        import sys

        for name in self._made:
            sys.modules.pop(name, None)
        return False


def _experiment_defaults() -> Config:
    """Key tree of ivlnce_baselines/config/default.py:14-163 plus the habitat_baselines keys the
    trainers read."""
    _C = CN()
    # --- habitat_baselines.config.default subset (un-vendored; Appendix D) ---
    _C.NUM_ENVIRONMENTS = 4
    _C.TORCH_GPU_ID = 0
    _C.CHECKPOINT_FOLDER = "data/checkpoints"
    _C.EVAL_CKPT_PATH_DIR = "data/checkpoints"
    _C.LOG_FILE = "train.log"
    _C.LOG_INTERVAL = 10
    _C.CHECKPOINT_INTERVAL = 50
    _C.NUM_UPDATES = 10000
    _C.SENSORS = ["RGB_SENSOR", "DEPTH_SENSOR"]
    # --- ivlnce_baselines/config/default.py:14-24 ---
    _C.BASE_TASK_CONFIG_PATH = "habitat_extensions/config/vlnce_task.yaml"
    _C.TASK_CONFIG = CN()
    _C.CMD_TRAILING_OPTS = []
    _C.TRAINER_NAME = "dagger"
    _C.ENV_NAME = "VLNCEDaggerEnv"
    _C.ENV_BACKEND = "synthetic"  # envs.construct_envs: the only vector env this package's trainers drive
    _C.SIMULATOR_GPU_IDS = [0]
    _C.VIDEO_OPTION = []
    _C.VIDEO_DIR = "data/videos/debug"
    _C.TENSORBOARD_DIR = "data/tensorboard_dirs/debug"
    _C.RESULTS_DIR = "data/checkpoints/pretrained/evals"
    # --- EVAL (default.py:29-37) ---
    _C.EVAL = CN()
    _C.EVAL.SPLIT = "val_seen"
    _C.EVAL.EPISODE_COUNT = -1
    _C.EVAL.LANGUAGES = ["en-US", "en-IN"]
    _C.EVAL.SAMPLE = False
    _C.EVAL.USE_CKPT_CONFIG = False
    _C.EVAL.SAVE_RESULTS = True
    _C.EVAL.ITERATIVE_MAP_RESET = "iterative"
    # (not a reference key) replay the eval step - mapper + policy.act - as captured hipGraphs
    # (graphed.py); falls back to eager launches for sampled actions and the known-map transformers
    _C.EVAL.USE_HIP_GRAPH = True
    _C.EVAL.ITERATIVE_GT_PATHS = "data/gt_ndtw.json"
    # --- IL (default.py:42-82) ---
    _C.IL = CN()
    _C.IL.lr = 2.5e-4
    _C.IL.batch_size = 5
    _C.IL.epochs = 4
    _C.IL.use_iw = True
    _C.IL.inflection_weight_coef = 3.2
    _C.IL.load_from_ckpt = False
    _C.IL.ckpt_to_load = "data/checkpoints/ckpt.0.pth"
    _C.IL.is_requeue = False
    _C.IL.DAGGER = CN()
    _C.IL.DAGGER.iterations = 10
    _C.IL.DAGGER.update_size = 5000
    _C.IL.DAGGER.p = 0.75
    _C.IL.DAGGER.expert_policy_sensor = "SHORTEST_PATH_SENSOR"
    _C.IL.DAGGER.expert_policy_sensor_uuid = "shortest_path_sensor"
    _C.IL.DAGGER.lmdb_map_size = 1.0e13
    _C.IL.DAGGER.lmdb_fp16 = False
    _C.IL.DAGGER.lmdb_commit_frequency = 500
    _C.IL.DAGGER.preload_lmdb_features = False
    _C.IL.DAGGER.lmdb_features_dir = "data/trajectories_dirs/debug/trajectories.lmdb"
    _C.IL.DAGGER.drop_existing_lmdb_features = True
    # (not reference keys) replay the sampled collection step - mapper + policy.act + beta-mixing - as captured hipGraphs
    # (trainers._RolloutStepper); re-index the host-side per-env lists when envs pause (quirk Q12, trainers.py)
    _C.IL.DAGGER.USE_HIP_GRAPH = True
    _C.IL.DAGGER.compact_paused_rows = False
    # --- RL.POLICY.OBS_TRANSFORMS (default.py:87-93) ---
    _C.RL = CN()
    _C.RL.POLICY = CN()
    _C.RL.POLICY.OBS_TRANSFORMS = CN()
    _C.RL.POLICY.OBS_TRANSFORMS.ENABLED_TRANSFORMS = []
    _C.RL.POLICY.OBS_TRANSFORMS.EGOCENTRIC_MAPPER = CN()
    _C.RL.POLICY.OBS_TRANSFORMS.EGOCENTRIC_MAPPER.resolution_meters = 0.1
    _C.RL.POLICY.OBS_TRANSFORMS.EGOCENTRIC_MAPPER.height_clip = 0.1
    _C.RL.POLICY.OBS_TRANSFORMS.EGOCENTRIC_MAPPER.height_meters = 6.4
    _C.RL.POLICY.OBS_TRANSFORMS.EGOCENTRIC_MAPPER.width_meters = 6.4
    # HIP mapper sizing (not reference keys): dense keep-highest table cells (0 = library default) and world-cloud
    # capacity in points (0 = 2^20 per env); exceeding either raises through MappingModule.check_status()
    _C.RL.POLICY.OBS_TRANSFORMS.EGOCENTRIC_MAPPER.table_cells = 0
    _C.RL.POLICY.OBS_TRANSFORMS.EGOCENTRIC_MAPPER.world_capacity = 0
    # --- MODEL (default.py:98-163) ---
    _C.MODEL = CN()
    _C.MODEL.policy_name = "CMAPolicy"
    _C.MODEL.ablate_depth = False
    _C.MODEL.ablate_rgb = False
    _C.MODEL.ablate_map = False
    _C.MODEL.ablate_instruction = False
    _C.MODEL.tour_memory = False
    _C.MODEL.tour_memory_variant = False
    _C.MODEL.memory_at_end = False
    _C.MODEL.train_unrolled = False
    _C.MODEL.disable_tour_memory = False
    _C.MODEL.INSTRUCTION_ENCODER = CN()
    _C.MODEL.INSTRUCTION_ENCODER.sensor_uuid = "instruction"
    _C.MODEL.INSTRUCTION_ENCODER.vocab_size = 2504
    _C.MODEL.INSTRUCTION_ENCODER.use_pretrained_embeddings = True
    _C.MODEL.INSTRUCTION_ENCODER.embedding_file = (
        "data/datasets/R2R_VLNCE_v1-3_preprocessed/embeddings.json.gz"
    )
    _C.MODEL.INSTRUCTION_ENCODER.dataset_vocab = (
        "data/datasets/R2R_VLNCE_v1-3_preprocessed/train/train.json.gz"
    )
    _C.MODEL.INSTRUCTION_ENCODER.fine_tune_embeddings = False
    _C.MODEL.INSTRUCTION_ENCODER.embedding_size = 50
    _C.MODEL.INSTRUCTION_ENCODER.hidden_size = 128
    _C.MODEL.INSTRUCTION_ENCODER.rnn_type = "LSTM"
    _C.MODEL.INSTRUCTION_ENCODER.final_state_only = True
    _C.MODEL.INSTRUCTION_ENCODER.bidirectional = True
    _C.MODEL.RGB_ENCODER = CN()
    _C.MODEL.RGB_ENCODER.cnn_type = "TorchVisionResNet50"
    _C.MODEL.RGB_ENCODER.output_size = 256
    _C.MODEL.RGB_ENCODER.trainable = False
    _C.MODEL.DEPTH_ENCODER = CN()
    _C.MODEL.DEPTH_ENCODER.cnn_type = "VlnResnetDepthEncoder"
    _C.MODEL.DEPTH_ENCODER.output_size = 128
    _C.MODEL.DEPTH_ENCODER.backbone = "resnet50"
    _C.MODEL.DEPTH_ENCODER.ddppo_checkpoint = "data/ddppo-models/gibson-2plus-resnet50.pth"
    _C.MODEL.DEPTH_ENCODER.trainable = False
    _C.MODEL.SEMANTIC_MAP_ENCODER = CN()
    _C.MODEL.SEMANTIC_MAP_ENCODER.classname = "SemanticMapEncoder"
    _C.MODEL.SEMANTIC_MAP_ENCODER.num_semantic_classes = 13
    _C.MODEL.SEMANTIC_MAP_ENCODER.output_size = 256
    _C.MODEL.SEMANTIC_MAP_ENCODER.channels = 32
    _C.MODEL.SEMANTIC_MAP_ENCODER.last_ch_mult = 4
    _C.MODEL.SEMANTIC_MAP_ENCODER.trainable = True
    _C.MODEL.SEMANTIC_MAP_ENCODER.from_pretrained = False
    _C.MODEL.SEMANTIC_MAP_ENCODER.checkpoint = ""
    _C.MODEL.SEMANTIC_MAP_ENCODER.custom_lr = False
    _C.MODEL.SEMANTIC_MAP_ENCODER.lr = 2.5e-6
    _C.MODEL.STATE_ENCODER = CN()
    _C.MODEL.STATE_ENCODER.hidden_size = 512
    _C.MODEL.STATE_ENCODER.rnn_type = "GRU"
    _C.MODEL.PROGRESS_MONITOR = CN()
    _C.MODEL.PROGRESS_MONITOR.use = False
    _C.MODEL.PROGRESS_MONITOR.alpha = 1.0
    return _C


def _task_defaults() -> Config:
    """Subset of habitat's task config + habitat_extensions/config/default.py:6-172 that the hot
    path reads: sensor geometry, seed, episode length, action count."""
    T = CN()
    T.SEED = 100
    T.ENVIRONMENT = CN()
    T.ENVIRONMENT.MAX_EPISODE_STEPS = 500
    T.ENVIRONMENT.ITERATOR_OPTIONS = CN()
    T.ENVIRONMENT.ITERATOR_OPTIONS.SHUFFLE = True
    T.ENVIRONMENT.ITERATOR_OPTIONS.MAX_SCENE_REPEAT_STEPS = -1
    # iterative (tour-by-tour) evaluation, habitat_extensions/config/default.py:19-44
    T.ENVIRONMENT.ITERATIVE = CN()
    T.ENVIRONMENT.ITERATIVE.ENABLED = False
    T.ENVIRONMENT.ITERATIVE.ENV_NAME = "VLNCEIterativeEnv"
    T.ENVIRONMENT.ITERATIVE.PRECISE_EPISODE_START = False
    T.ENVIRONMENT.ITERATIVE.ORACLE_STOP_ON_ERROR = False
    T.ENVIRONMENT.ITERATIVE.ORACLE_STEP_ERROR_LIMIT = -1
    T.ENVIRONMENT.ITERATIVE.ORACLE_GOAL_PHASE = True
    T.ENVIRONMENT.ITERATIVE.ORACLE_PHASES = True
    T.SIMULATOR = CN()
    T.SIMULATOR.FORWARD_STEP_SIZE = 0.25
    T.SIMULATOR.TURN_ANGLE = 15
    T.SIMULATOR.HABITAT_SIM_V0 = CN()
    T.SIMULATOR.HABITAT_SIM_V0.GPU_DEVICE_ID = 0
    T.SIMULATOR.AGENT_0 = CN()
    T.SIMULATOR.AGENT_0.SENSORS = ["RGB_SENSOR", "DEPTH_SENSOR"]
    T.SIMULATOR.AGENT_0.HEIGHT = 1.5
    T.SIMULATOR.AGENT_0.RADIUS = 0.1
    T.SIMULATOR.RGB_SENSOR = CN()
    T.SIMULATOR.RGB_SENSOR.WIDTH = 224
    T.SIMULATOR.RGB_SENSOR.HEIGHT = 224
    T.SIMULATOR.RGB_SENSOR.HFOV = 90
    T.SIMULATOR.DEPTH_SENSOR = CN()
    T.SIMULATOR.DEPTH_SENSOR.WIDTH = 256
    T.SIMULATOR.DEPTH_SENSOR.HEIGHT = 256
    T.SIMULATOR.DEPTH_SENSOR.HFOV = 90
    T.SIMULATOR.DEPTH_SENSOR.MIN_DEPTH = 0.0
    T.SIMULATOR.DEPTH_SENSOR.MAX_DEPTH = 10.0
    T.SIMULATOR.DEPTH_SENSOR.NORMALIZE_DEPTH = True
    T.SIMULATOR.SEMANTIC_SENSOR = CN()
    T.SIMULATOR.SEMANTIC_SENSOR.WIDTH = 256
    T.SIMULATOR.SEMANTIC_SENSOR.HEIGHT = 256
    T.SIMULATOR.SEMANTIC_SENSOR.HFOV = 90
    T.TASK = CN()
    T.TASK.TYPE = "VLN-v0"
    T.TASK.SUCCESS_DISTANCE = 3.0
    T.TASK.SENSORS = ["INSTRUCTION_SENSOR", "SHORTEST_PATH_SENSOR", "VLN_ORACLE_PROGRESS_SENSOR"]
    T.TASK.POSSIBLE_ACTIONS = ["STOP", "MOVE_FORWARD", "TURN_LEFT", "TURN_RIGHT"]
    T.TASK.MEASUREMENTS = ["DISTANCE_TO_GOAL", "SUCCESS", "SPL", "NDTW", "SDTW", "PATH_LENGTH"]
    T.TASK.NDTW = CN()  # habitat_extensions/config/default.py:108-115
    T.TASK.NDTW.TYPE = "NDTW"
    T.TASK.NDTW.SPLIT = "val_seen"
    T.TASK.NDTW.FDTW = True  # False: exact DTW
    T.TASK.NDTW.GT_PATH = "data/datasets/R2R_VLNCE_v1-3_preprocessed/{split}/{split}_gt.json.gz"
    T.TASK.NDTW.SUCCESS_DISTANCE = 3.0
    T.TASK.SDTW = CN()
    T.TASK.SDTW.TYPE = "SDTW"
    T.TASK.INSTRUCTION_SENSOR_UUID = "instruction"
    T.DATASET = CN()
    T.DATASET.TYPE = "VLN-CE-v1"
    T.DATASET.SPLIT = "train"
    T.DATASET.DATA_PATH = "data/datasets/R2R_VLNCE_v1-3_preprocessed/{split}/{split}.json.gz"
    T.DATASET.SCENES_DIR = "data/scene_datasets/"
    return T


def get_task_config(config_path: Optional[str] = None) -> Config:
    """habitat_extensions/config/default.py:175-214 (`get_extended_config`)."""
    cfg = _task_defaults()
    if config_path:
        for p in config_path.split(CONFIG_FILE_SEPARATOR):
            if os.path.exists(p):
                cfg.merge_from_file(p)
    return cfg


def get_config(
    config_paths: Optional[Union[List[str], str]] = None,
    opts: Optional[list] = None,
) -> Config:
    """Same merge order as ivlnce_baselines/config/default.py:172-212: defaults, then each YAML
    (re-reading TASK_CONFIG whenever BASE_TASK_CONFIG_PATH changes), then CLI opts; frozen."""
    config = _experiment_defaults()
    config.TASK_CONFIG = _task_defaults()
    if config_paths:
        if isinstance(config_paths, str):
            if CONFIG_FILE_SEPARATOR in config_paths:
                config_paths = config_paths.split(CONFIG_FILE_SEPARATOR)
            else:
                config_paths = [config_paths]
        prev_task_config = ""
        for config_path in config_paths:
            config.merge_from_file(config_path)
            if config.BASE_TASK_CONFIG_PATH != prev_task_config:
                config.TASK_CONFIG = get_task_config(config.BASE_TASK_CONFIG_PATH)
                prev_task_config = config.BASE_TASK_CONFIG_PATH
    if opts:
        config.CMD_TRAILING_OPTS = list(opts)
        config.merge_from_list(opts)
    config.freeze()
    return config
