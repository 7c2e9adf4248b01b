"""Import shim that lets the reference's own Python (under /root/reference, read-only) be imported
in the build container to GENERATE golden vectors.  Only the generator
scripts next to this file use it; nothing in tests/, bench.py or the product imports it, and
/root/reference does not exist on the GPU box.

Stubs registered before import: gym, habitat, habitat_baselines (registry + the habitat-lab
arithmetic restated in oracle/habitat_ext_ref.py), torchvision, torch_scatter.
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = "/root/reference"
sys.path.insert(0, REPO)


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install():
    if "ivlnce_baselines" in sys.modules:
        return
    ext = _load("oracle_habitat_ext_ref", os.path.join(REPO, "oracle", "habitat_ext_ref.py"))
    cfgmod = _load("ivln_cfg_for_shim", os.path.join(REPO, "ivln-ce_amd", "config.py"))

    # ---- gym -----------------------------------------------------------
    class Space:
        pass

    class Box(Space):
        def __init__(self, low, high, shape=None, dtype=np.float32):
            self.shape = tuple(shape) if shape is not None else np.asarray(low).shape
            self.low = np.full(self.shape, low) if np.isscalar(low) else np.asarray(low)
            self.high = np.full(self.shape, high) if np.isscalar(high) else np.asarray(high)
            self.dtype = np.dtype(dtype)

    class Dict(Space):
        def __init__(self, spaces):
            self.spaces = dict(spaces)

    class Discrete(Space):
        def __init__(self, n):
            self.n = n

    spaces = _mod("gym.spaces", Box=Box, Dict=Dict, Discrete=Discrete, Space=Space)
    _mod("gym", Space=Space, spaces=spaces)

    # ---- habitat ---------------------------------------------------------
    hab = _mod("habitat", Config=cfgmod.Config)
    _mod("habitat.config", Config=cfgmod.Config)
    _mod("habitat.config.default", Config=cfgmod.Config, CONFIG_FILE_SEPARATOR=",")
    _mod("habitat.core")
    _mod("habitat.core.simulator", Observations=dict)
    hab.logger = types.SimpleNamespace(info=print, warn=print)

    # ---- habitat_baselines ------------------------------------------------
    class _Registry:
        def __init__(self):
            self.policies, self.trainers, self.obs_transformers = {}, {}, {}

        def register_policy(self, cls=None, name=None):
            def wrap(c):
                self.policies[name or c.__name__] = c
                return c
            return wrap(cls) if cls is not None else wrap

        def register_obs_transformer(self, cls=None, name=None):
            def wrap(c):
                self.obs_transformers[name or c.__name__] = c
                return c
            return wrap(cls) if cls is not None else wrap

        def register_trainer(self, cls=None, name=None):
            def wrap(c):
                self.trainers[name or c.__name__] = c
                return c
            return wrap(cls) if cls is not None else wrap

        def get_policy(self, n):
            return self.policies[n]

    reg = _Registry()
    _mod("habitat_baselines")
    _mod("habitat_baselines.common")
    _mod("habitat_baselines.common.baseline_registry", baseline_registry=reg)
    _mod("habitat_baselines.common.tensor_dict", DictTree=dict)

    class ObservationTransformer(nn.Module):
        pass

    _mod("habitat_baselines.common.obs_transformers", ObservationTransformer=ObservationTransformer)
    _mod("habitat_baselines.rl")
    _mod("habitat_baselines.rl.ppo")

    class Net(nn.Module):
        pass

    class Policy(nn.Module):
        pass

    _mod("habitat_baselines.rl.ppo.policy", Net=Net, Policy=Policy)
    resnet_mod = _mod("habitat_baselines.rl.ddppo.policy.resnet", resnet50=ext.resnet50)
    _mod("habitat_baselines.rl.ddppo")
    _mod("habitat_baselines.rl.ddppo.policy", resnet=resnet_mod)
    _mod("habitat_baselines.rl.ddppo.policy.resnet_policy", ResNetEncoder=ext.ResNetEncoder)
    _mod("habitat_baselines.rl.models")
    _mod(
        "habitat_baselines.rl.models.rnn_state_encoder",
        build_rnn_state_encoder=ext.build_rnn_state_encoder,
    )

    # ---- torchvision / torch_scatter ---------------------------------------
    class Compose:
        def __init__(self, ts):
            self.transforms = ts

        def __call__(self, x):
            for t in self.transforms:
                x = t(x)
            return x

    tv_t = _mod("torchvision.transforms", Compose=Compose)
    tv_ref = _load("oracle_torchvision_ref", os.path.join(REPO, "oracle", "torchvision_ref.py"))
    tv_r = _mod("torchvision.models.resnet", model_urls={})
    tv_m = _mod("torchvision.models", resnet=tv_r, resnet50=tv_ref.resnet50)
    _mod("torchvision", transforms=tv_t, models=tv_m)
    _mod("torch_scatter", scatter_max=ext.scatter_max)

    # ---- the reference package, bypassing its __init__ (imports trainers -> habitat) ---
    for pkg in [
        "ivlnce_baselines",
        "ivlnce_baselines.common",
        "ivlnce_baselines.common.mapping_module",
        "ivlnce_baselines.models",
        "ivlnce_baselines.models.encoders",
    ]:
        m = types.ModuleType(pkg)
        m.__path__ = [os.path.join(REF, *pkg.split("."))]
        sys.modules[pkg] = m
    return reg


def default_model_config():
    """ivlnce_baselines/config/default.py:98-163 values, random-init embeddings (no data files)."""
    cfgmod = sys.modules["ivln_cfg_for_shim"]
    cfg = cfgmod._experiment_defaults()
    cfg.MODEL.policy_name = "MapCMAPolicy"
    cfg.MODEL.INSTRUCTION_ENCODER.use_pretrained_embeddings = False
    cfg.MODEL.DEPTH_ENCODER.ddppo_checkpoint = "NONE"
    return cfg


def observation_space():
    sp = sys.modules["gym.spaces"]
    return sp.Dict(
        {
            "depth": sp.Box(0.0, 1.0, (256, 256, 1), np.float32),
            "occupancy_map": sp.Box(0, 255, (64, 64), np.uint8),
            "semantic_map": sp.Box(0, 255, (64, 64), np.uint8),
            "instruction": sp.Box(0, 2504, (200,), np.int64),
        }
    )
