"""smoke(): one tiny act + one tiny DAgger update of the MapCMA policy on cuda:0, checked against
the torch-CPU oracle.  Lives under tests/ (called by __graft_entry__.smoke()): nothing in the product
package may import the oracle."""
import numpy as np
import torch


def run():
    from oracle.policy_ref import MapCMAPolicyRef  # checker only (smoke), never on the product path

    from ivln_ce_amd.config import get_config
    from ivln_ce_amd.policy import MapCMAPolicy
    from ivln_ce_amd.spaces import Box, Dict, Discrete
    from ivln_ce_amd.synthetic import SyntheticRollout
    from ivln_ce_amd.trainers import FlatAdam, update_agent

    dev = torch.device("cuda:0")
    cfg = get_config(opts=["MODEL.policy_name", "MapCMAPolicy", "MODEL.INSTRUCTION_ENCODER.use_pretrained_embeddings",
                           False, "MODEL.DEPTH_ENCODER.ddppo_checkpoint", "NONE"])
    space = Dict({"depth": Box(0.0, 1.0, (256, 256, 1), np.float32), "occupancy_map": Box(0, 255, (64, 64), np.uint8),
                  "semantic_map": Box(0, 255, (64, 64), np.uint8), "instruction": Box(0, 2504, (200,), np.int64)})
    import os
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    from det_init import det_fill  # crc32(key)-seeded weights: O(1) logits (the default head gain of 0.01 gives logits ~ 0)

    torch.manual_seed(0)
    pol = det_fill(MapCMAPolicy.from_config(cfg, space, Discrete(4)), seed=0)
    ref = MapCMAPolicyRef()
    ref.load_state_dict(pol.state_dict())
    pol, ref = pol.to(dev).eval(), ref.eval()
    B = 2
    obs = SyntheticRollout(B=B, seed=9, n_tokens=30).step()
    g = torch.Generator().manual_seed(1)
    obs["occupancy_map"] = (torch.rand(B, 64, 64, generator=g) < 0.3).to(torch.uint8)
    obs["semantic_map"] = (torch.randint(0, 13, (B, 64, 64), generator=g) * obs["occupancy_map"]).to(torch.uint8)
    rnn, prev, masks = torch.zeros(B, 2, 512), torch.zeros(B, 1, dtype=torch.long), torch.zeros(B, 1, dtype=torch.uint8)
    with torch.no_grad():
        lr, sr, fr = ref.logits(obs, rnn, prev, masks)
        dobs = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in obs.items()}
        f, s = pol.net(dobs, rnn.to(dev), prev.to(dev), masks.to(dev))
        lg = pol.action_distribution.raw_logits(f)
    err = float((lg.cpu() - lr).abs().max())
    ferr, serr = float((f.cpu() - fr).abs().max()), float((s.cpu() - sr).abs().max())
    spread = float(lr.max() - lr.min())
    assert spread > 0.05, f"degenerate smoke weights: oracle logits span only {spread}"
    assert err < 1e-4, f"policy logits mismatch vs oracle: {err}"
    assert ferr < 2e-4 and serr < 2e-4, f"policy features / recurrent state mismatch vs oracle: {ferr} / {serr}"
    # one tiny DAgger update (T=3, N=2) with cached depth features
    pol.train()
    T, N = 3, 2
    tr = {"depth_features": torch.randn(T * N, 128, 4, 4, generator=g).to(dev),
          "occupancy_map": obs["occupancy_map"].repeat(T, 1, 1).float().to(dev),
          "semantic_map": obs["semantic_map"].repeat(T, 1, 1).float().to(dev),
          "instruction": obs["instruction"].repeat(T, 1).float().to(dev)}
    nd = torch.ones(T, N, dtype=torch.uint8)
    nd[0] = 0
    opt = FlatAdam(pol, lr=2.5e-4)
    loss, _, _ = update_agent(pol, opt, tr, torch.randint(0, 4, (T * N, 1), generator=g).to(dev), nd.view(-1, 1).to(dev),
                              torch.randint(0, 4, (T, N), generator=g).to(dev), torch.ones(T, N).to(dev))
    assert np.isfinite(loss)
    print(f"smoke: HIP policy logits within {err:.1e} (span {spread:.2f}), features {ferr:.1e}, recurrent state {serr:.1e} of "
          f"the oracle; one HIP DAgger update, loss {loss:.4f}")
