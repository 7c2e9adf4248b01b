"""Golden data for the external-weight / checkpoint LOADERS that make the plugins drop-in (build container only): every
file format is written here with the REFERENCE's own classes and key names, loaded back through the REFERENCE's own
loader code, and one forward is recorded.  The files themselves are far too large to commit (7 M / 82 M / 13.6 M
parameters), so what is committed is (a) the manifests - key names and shapes exactly as the reference's classes
produce them - from which tests/loader_files.py rebuilds byte-identical files with det_init.det_value, and (b) the
forwards the reference computed after loading them.  -> tests/golden/loader_golden.npz + loader_manifest.json

  ddppo      `VlnResnetDepthEncoder(checkpoint=...)` (models/encoders/resnet_encoders.py:48-61): a DD-PPO checkpoint
             {"state_dict": {"actor_critic.net.visual_encoder.<k>": ...}} with foreign keys beside them
  rednet     `PredictSemantics.load_model` + `convert_weights_cuda_cpu` (mapping_module/mapper.py:758-779): a pickle
             {"model_state": {"module.<k>": ...}} as DataParallel training leaves it
  embeddings `InstructionEncoder._load_embeddings` (models/encoders/instruction_encoder.py:35-39, 52-66): gzip'd JSON
             of a (vocab, 50) list, `use_pretrained_embeddings` with and without `fine_tune_embeddings`
  map_ckpt   `SemanticMapEncoder(from_pretrained=True, checkpoint=...)` (models/encoders/map_encoder.py:62-70):
             {"state_dict": {"encoder.cnn.<k>": ...}}
  trainer    `BaseVLNCETrainer.save_checkpoint` / `_initialize_policy(load_from_ckpt=True)` with `IL.is_requeue`
             (common/base_il_trainer.py:98-106, 143-168): {"state_dict", "config", "optim_state", "dagger_it", "epoch",
             "step_id"} where "config" pickles as `habitat.config.default.Config` and "optim_state" is
             `torch.optim.Adam.state_dict()` over `policy.parameters()` (one group, or two with
             SEMANTIC_MAP_ENCODER.custom_lr) - recorded: the policy's state_dict manifest, the optimizer's index ->
             parameter table for both group layouts, and the loaded policy's act() on a seeded batch."""
import gzip
import json
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import _ref_shim  # noqa: E402

_ref_shim.install()
torch.set_num_threads(4)
from det_init import det_value  # noqa: E402
from loader_files import (CKPT_SEED, embeddings_table, write_ddppo_checkpoint, write_embeddings_file,  # noqa: E402
                          write_map_encoder_checkpoint, write_rednet_pickle)

from ivlnce_baselines.common.mapping_module.mapper import PredictSemantics  # noqa: E402
from ivlnce_baselines.common.mapping_module.rednet import RedNet  # noqa: E402
from ivlnce_baselines.models.encoders.instruction_encoder import InstructionEncoder  # noqa: E402
from ivlnce_baselines.models.encoders.map_encoder import SemanticMapEncoder  # noqa: E402
from ivlnce_baselines.models.encoders.resnet_encoders import VlnResnetDepthEncoder  # noqa: E402
from ivlnce_baselines.models.map_cma_policy import MapCMAPolicy  # noqa: E402


def manifest(sd):
    return [[k, list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in sd.items()]


def main():
    out, man = {}, {}
    space = _ref_shim.observation_space()
    sp = sys.modules["gym.spaces"]
    g = torch.Generator().manual_seed(77)
    with tempfile.TemporaryDirectory() as tmp:
        # ---- DD-PPO depth checkpoint ---------------------------------------------------------------------------
        probe = VlnResnetDepthEncoder(space, output_size=128, checkpoint="NONE", backbone="resnet50", spatial_output=True)
        man["ddppo"] = manifest(probe.visual_encoder.state_dict())
        path = os.path.join(tmp, "ddppo.pth")
        write_ddppo_checkpoint(path, man["ddppo"])
        enc = VlnResnetDepthEncoder(space, output_size=128, checkpoint=path, backbone="resnet50", spatial_output=True)
        enc.eval()
        depth = torch.rand(2, 256, 256, 1, generator=torch.Generator().manual_seed(78))  # (the test draws the same)
        with torch.no_grad():
            out["ddppo_features"] = enc.visual_encoder({"depth": depth}).numpy()
        # ---- RedNet pickle with DataParallel's "module." prefix ----------------------------------------------------
        man["rednet"] = manifest(RedNet({"n_classes": 13, "resnet_pretrained": False}).state_dict())
        path = os.path.join(tmp, "rednet.pkl")
        write_rednet_pickle(path, man["rednet"])
        ps = PredictSemantics()
        ps.model = RedNet({"n_classes": 13, "resnet_pretrained": False})
        ps.load_model(path)
        ps.model.eval()
        rgb = torch.rand(1, 3, 64, 64, generator=g)
        dep = torch.rand(1, 1, 64, 64, generator=g)
        with torch.no_grad():
            out["rednet_rgb"], out["rednet_depth"] = rgb.numpy(), dep.numpy()
            out["rednet_scores"] = ps.model(rgb, dep).numpy()
        # ---- pretrained embeddings ------------------------------------------------------------------------------
        cfg = _ref_shim.default_model_config().MODEL.INSTRUCTION_ENCODER
        path = os.path.join(tmp, "embeddings.json.gz")
        write_embeddings_file(path, cfg.vocab_size, cfg.embedding_size)
        cfg.defrost() if hasattr(cfg, "defrost") else None
        cfg.use_pretrained_embeddings, cfg.embedding_file, cfg.final_state_only = True, path, False
        tokens = torch.zeros(2, 200, dtype=torch.long)
        tokens[0, :37] = torch.randint(2, cfg.vocab_size, (37,), generator=g)
        tokens[1, :80] = torch.randint(2, cfg.vocab_size, (80,), generator=g)
        out["emb_tokens"] = tokens.numpy()
        for tune in (False, True):
            cfg.fine_tune_embeddings = tune
            ie = InstructionEncoder(cfg)
            assert ie.embedding_layer.weight.requires_grad == tune
            sd = ie.state_dict()
            for k, v in sd.items():
                if not k.startswith("embedding_layer"):
                    sd[k] = det_value("net.instruction_encoder." + k, v)
            ie.load_state_dict(sd)
            with torch.no_grad():
                out[f"emb_out_tune{int(tune)}"] = ie({"instruction": tokens}).numpy()
        out["emb_table_head"] = embeddings_table(cfg.vocab_size, cfg.embedding_size)[:4].numpy()
        # ---- pretrained semantic-map encoder --------------------------------------------------------------------
        sm = SemanticMapEncoder(space, 13, 32, 4, True, False, None)
        man["map_ckpt"] = manifest(sm.cnn.state_dict())
        path = os.path.join(tmp, "map_encoder.pth")
        write_map_encoder_checkpoint(path, man["map_ckpt"])
        sm = SemanticMapEncoder(space, 13, 32, 4, False, True, path).eval()
        occ = (torch.rand(2, 64, 64, generator=g) < 0.3).to(torch.uint8)
        sem = (torch.randint(0, 13, (2, 64, 64), generator=g) * occ).to(torch.uint8)
        with torch.no_grad():
            out["map_occ"], out["map_sem"] = occ.numpy(), sem.numpy()
            out["map_features"] = sm({"occupancy_map": occ, "semantic_map": sem}).numpy()
        # ---- trainer checkpoint: state_dict manifest + torch.optim.Adam's index table ---------------------------------
        mc = _ref_shim.default_model_config()
        pol = MapCMAPolicy.from_config(mc, space, sp.Discrete(4))
        man["policy"] = manifest(pol.state_dict())
        man["policy_parameters"] = [[k, list(p.shape), bool(p.requires_grad)] for k, p in pol.named_parameters()]
        for custom in (False, True):
            if custom:  # base_il_trainer.py:78-92: [map-encoder params, the rest]
                sem_p = [p for k, p in pol.named_parameters() if k.startswith("net.map_encoder")]
                reg_p = [p for k, p in pol.named_parameters() if not k.startswith("net.map_encoder")]
                opt = torch.optim.Adam([{"params": sem_p}, {"params": reg_p}], lr=2.5e-4)
                opt.param_groups[0]["lr"] = 1e-3
            else:
                opt = torch.optim.Adam(pol.parameters(), lr=2.5e-4)
            for p in pol.parameters():  # one step, so that every trainable parameter has state
                if p.requires_grad:
                    p.grad = torch.zeros_like(p)
            opt.step()
            sd = opt.state_dict()
            names = {id(p): k for k, p in pol.named_parameters()}
            order = [names[id(p)] for grp in opt.param_groups for p in grp["params"]]
            man[f"adam_custom{int(custom)}"] = {
                "index_to_name": order,
                "state_indices": sorted(int(i) for i in sd["state"]),
                "state_keys": sorted(next(iter(sd["state"].values())).keys()),
                "step_type": type(next(iter(sd["state"].values()))["step"]).__name__,
                "groups": [{k: (v if k != "params" else list(v)) for k, v in grp.items()
                            if k in ("lr", "betas", "eps", "weight_decay", "amsgrad", "params")} for grp in sd["param_groups"]],
            }
        man["checkpoint_keys"] = ["state_dict", "config", "optim_state", "dagger_it", "epoch", "step_id"]
        man["config_pickles_as"] = "habitat.config.default.Config"
    np.savez_compressed(os.path.join(HERE, "loader_golden.npz"), **out)
    json.dump(man, open(os.path.join(HERE, "loader_manifest.json"), "w"))
    for k, v in out.items():
        print(k, v.shape, float(np.abs(v).mean()))
    print({k: (len(v) if isinstance(v, list) else "...") for k, v in man.items()})


if __name__ == "__main__":
    main()
