"""Writers of the external weight / checkpoint FILES the plugins load (DD-PPO depth checkpoint, RedNet pickle, embeddings
file, pretrained map-encoder checkpoint, reference-format trainer checkpoint), from the key-name manifests of
tests/golden/loader_manifest.json and det_init.det_value.  Shared by tests/golden/gen_loader_golden.py (which loads the
files through the REFERENCE's own loaders, build container only) and tests/test_gpu_loaders.py (which loads
byte-identical files through this package): the files are 30-330 MB, far too large to commit, the manifests and the
reference's forwards after loading are not.  Nothing here reads /root/reference."""
import gzip
import json
import os
import sys
import types

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
from det_init import det_value  # noqa: E402

CKPT_SEED = 5  # != the seeds the other goldens fill modules with: a forward only matches if the FILE was loaded


def _like(shape, dtype):
    return torch.zeros(shape, dtype=getattr(torch, dtype))


def ddppo_state(man):
    """{"actor_critic.net.visual_encoder.<k>": tensor} + keys of other sub-modules the loader has to skip
    (resnet_encoders.py:50-58: k.split(".")[2:] must start with "visual_encoder")."""
    sd = {}
    for k, shape, dt in man:
        ck = "actor_critic.net.visual_encoder." + k
        sd[ck] = det_value(ck, _like(shape, dt), seed=CKPT_SEED)
    sd["actor_critic.net.prev_action_embedding.weight"] = torch.ones(5, 32)
    sd["actor_critic.net.tgt_embeding.weight"] = torch.ones(32, 3)
    sd["actor_critic.net.state_encoder.rnn.weight_ih_l0"] = torch.ones(8, 8)
    sd["actor_critic.critic.fc.weight"] = torch.ones(1, 512)
    sd["actor_critic.action_distribution.linear.bias"] = torch.ones(4)
    return sd


def write_ddppo_checkpoint(path, man):
    torch.save({"state_dict": ddppo_state(man), "config": None, "extra_state": {"step": 1234}}, path)


def rednet_state(man, prefix="module."):
    """RedNet's state_dict as a DataParallel-trained pickle holds it (every key behind "module.")."""
    return {prefix + k: det_value(prefix + k, _like(shape, dt), seed=CKPT_SEED, conv_gain=0.6) for k, shape, dt in man}


def write_rednet_pickle(path, man, prefix="module."):
    torch.save({"model_state": rednet_state(man, prefix), "epoch": 3, "best_iou": 0.5}, path)


def embeddings_table(vocab, dim):
    g = torch.Generator().manual_seed(91)
    t = 0.5 * torch.randn(vocab, dim, generator=g)
    t[0].zero_()                      # PAD
    t[1] = t[2:].mean(0)              # UNK = mean of the word embeddings (instruction_encoder.py:52-60)
    return t


def write_embeddings_file(path, vocab, dim):
    with gzip.open(path, "wt") as f:
        json.dump(embeddings_table(vocab, dim).tolist(), f)


def map_encoder_state(man):
    sd = {"encoder.cnn." + k: det_value("encoder.cnn." + k, _like(shape, dt), seed=CKPT_SEED) for k, shape, dt in man}
    sd["decoder.deconv.weight"] = torch.ones(3, 3)  # the pretraining task's head: ignored (map_encoder.py:64-69)
    return sd


def write_map_encoder_checkpoint(path, man):
    torch.save({"state_dict": map_encoder_state(man), "epoch": 9}, path)


# ---- reference-format trainer checkpoint ------------------------------------------------------------------------------
def policy_state(man, seed=CKPT_SEED):
    return {k: det_value(k, _like(shape, dt), seed=seed) for k, shape, dt in man}


def adam_state(layout, param_shapes, step=7):
    """`torch.optim.Adam.state_dict()` as the reference's trainer saves it (base_il_trainer.py:158-168): per-index
    {"step", "exp_avg", "exp_avg_sq"} for the parameters that had a gradient, and the param_groups."""
    state = {}
    for i in layout["state_indices"]:
        name = layout["index_to_name"][i]
        like = torch.zeros(param_shapes[name])
        state[i] = {"step": step, "exp_avg": 0.01 * det_value("m." + name, like, seed=1),
                    "exp_avg_sq": (0.01 * det_value("v." + name, like, seed=2)) ** 2}
    groups = [dict(g, betas=tuple(g["betas"])) for g in layout["groups"]]
    return {"state": state, "param_groups": groups}


def install_fake_habitat_config():
    """A stand-in for the class a reference checkpoint's "config" entry pickles as, `habitat.config.default.Config` (a
    yacs CfgNode: a dict subclass whose nodes carry `__immutable__` & co. in their instance __dict__).  Returns the
    class and a function that removes the fake modules again - the product has to unpickle the file WITHOUT them."""
    created = []

    class Config(dict):
        def __init__(self, init=None):
            super().__init__()
            self.__dict__.update({"__immutable__": True, "__deprecated_keys__": set(), "__renamed_keys__": {},
                                  "__new_allowed__": True})
            for k, v in (init or {}).items():
                self[k] = Config(v) if isinstance(v, dict) else v

    Config.__module__, Config.__qualname__ = "habitat.config.default", "Config"
    for name in ("habitat", "habitat.config", "habitat.config.default"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
            created.append(name)
    sys.modules["habitat.config.default"].Config = Config

    def remove():
        for name in created:
            sys.modules.pop(name, None)
        if "habitat.config.default" in sys.modules and getattr(sys.modules["habitat.config.default"], "Config", None) is Config:
            del sys.modules["habitat.config.default"].Config

    return Config, remove
