"""GPU parity of the HIP MapCMA policy forward (through the registry plugin) against the goldens
produced by the reference's own MapCMAPolicy, and against the torch-CPU oracle on other shapes.
fp32 MFMA == fmaf chain, so differences are summation-order only: tolerance 2e-4 abs on
features/states (values O(1)), 1e-4 on log-probs."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
G = os.path.join(os.path.dirname(__file__), "golden")
ATOL = 2e-4


def make_policy(use_pm=False, train=False):
    from det_init import det_fill

    from ivln_ce_amd.config import get_config
    from ivln_ce_amd.policy import MapCMAPolicy
    from ivln_ce_amd.spaces import Box, Dict, Discrete

    cfg = get_config(opts=[
        "MODEL.policy_name", "MapCMAPolicy", "MODEL.INSTRUCTION_ENCODER.use_pretrained_embeddings", False,
        "MODEL.DEPTH_ENCODER.ddppo_checkpoint", "NONE", "MODEL.PROGRESS_MONITOR.use", use_pm,
    ])
    space = Dict({
        "depth": Box(0.0, 1.0, (256, 256, 1), np.float32), "occupancy_map": Box(0, 255, (64, 64), np.uint8),
        "semantic_map": Box(0, 255, (64, 64), np.uint8), "instruction": Box(0, 2504, (200,), np.int64),
    })
    pol = MapCMAPolicy.from_config(cfg, space, Discrete(4))
    det_fill(pol, seed=0)
    pol = pol.to("cuda:0")
    pol.train() if train else pol.eval()
    return pol


def _err(name, got, ref, log):
    e = float(np.abs(got - ref).max())
    log.append(f"{name}: max|err|={e:.3e} (ref max {float(np.abs(ref).max()):.3e})")
    return e


def test_state_dict_keys_identical_to_oracle():
    from oracle.policy_ref import MapCMAPolicyRef

    pol = make_policy()
    ref = MapCMAPolicyRef()
    a = {k: tuple(v.shape) for k, v in pol.state_dict().items()}
    b = {k: tuple(v.shape) for k, v in ref.state_dict().items()}
    assert a == b


def test_act_matches_reference_golden():
    g = np.load(os.path.join(G, "policy_act.npz"))
    pol = make_policy()
    dev = torch.device("cuda:0")
    log = []
    feats = {}
    pol.net.depth_encoder.visual_encoder.register_forward_hook(lambda m, i, o: feats.__setitem__("depth", o.detach().clone()))
    instr = torch.from_numpy(g["instruction"]).to(dev)
    worst = 0.0
    for t in range(2):
        obs = {
            "depth": torch.from_numpy(g[f"depth_{t}"]).to(dev), "occupancy_map": torch.from_numpy(g[f"occ_{t}"]).to(dev),
            "semantic_map": torch.from_numpy(g[f"sem_{t}"]).to(dev), "instruction": instr,
        }
        with torch.no_grad():
            txt, lengths = pol.net.instruction_encoder(obs)
            mp = pol.net.map_encoder(obs)
            f, rnn = pol.net(obs, torch.from_numpy(g[f"rnn_in_{t}"]).to(dev), torch.from_numpy(g[f"prev_{t}"]).to(dev),
                             torch.from_numpy(g[f"masks_{t}"]).to(dev))
            logits = pol.action_distribution.raw_logits(f)
            act = pol._act(f, True)
        if t == 0:
            Lmax = g["txt_feat"].shape[2]
            worst = max(worst, _err("txt", txt.cpu().numpy()[:, :, :Lmax], g["txt_feat"], log))
            assert float(txt[:, :, Lmax:].abs().max()) == 0.0 if Lmax < txt.shape[2] else True
            assert lengths.cpu().tolist() == [80, 23, 200]
        worst = max(worst, _err(f"depth_feat_{t}", feats["depth"].cpu().numpy(), g[f"depth_feat_{t}"], log))
        worst = max(worst, _err(f"map_feat_{t}", mp.cpu().numpy(), g[f"map_feat_{t}"], log))
        worst = max(worst, _err(f"features_{t}", f.cpu().numpy(), g[f"features_{t}"], log))
        worst = max(worst, _err(f"rnn_out_{t}", rnn.cpu().numpy(), g[f"rnn_out_{t}"], log))
        lp = torch.log_softmax(logits, -1).cpu().numpy()
        e = _err(f"logprobs_{t}", lp, g[f"logits_{t}"], log)
        assert act.shape == (3, 1) and act.dtype == torch.int64
        assert np.array_equal(act.cpu().numpy()[:, 0], g[f"logits_{t}"].argmax(-1))
        assert e < 1e-4, "\n".join(log)
    print("\n".join(log))
    os.makedirs("gpurun_out", exist_ok=True)
    open("gpurun_out/policy_parity.log", "w").write("\n".join(log) + "\n")
    assert worst < ATOL, "\n".join(log)


@pytest.mark.parametrize("B", [1, 4, 8])
def test_act_matches_oracle_other_batches(B):
    from det_init import det_fill

    from ivln_ce_amd.synthetic import SyntheticRollout
    from oracle.policy_ref import MapCMAPolicyRef

    torch.set_num_threads(8)
    pol = make_policy()
    ref = det_fill(MapCMAPolicyRef(), seed=0).eval()
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(B)
    roll = SyntheticRollout(B=B, seed=100 + B, n_tokens=40 + 10 * B)
    obs = roll.step()
    obs["occupancy_map"] = (torch.rand(B, 64, 64, generator=g) < 0.4).to(torch.uint8)
    obs["semantic_map"] = (torch.randint(0, 13, (B, 64, 64), generator=g) * obs["occupancy_map"]).to(torch.uint8)
    rnn = 0.1 * torch.randn(B, 2, 512, generator=g)
    prev = torch.randint(0, 4, (B, 1), generator=g)
    masks = (torch.rand(B, 1, generator=g) < 0.7).to(torch.uint8)
    with torch.no_grad():
        lr, sr, fr = ref.logits(obs, rnn, prev, masks)
        dobs = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in obs.items()}
        f, s = pol.net(dobs, rnn.to(dev), prev.to(dev), masks.to(dev))
        lg = pol.action_distribution.raw_logits(f)
    assert float((f.cpu() - fr).abs().max()) < ATOL
    assert float((s.cpu() - sr).abs().max()) < ATOL
    assert float((lg.cpu() - lr).abs().max()) < 1e-4


@pytest.mark.parametrize("streams,depth_mode", [(True, 0), (False, 2), ("split", 0), ("split", 2)])
def test_graphed_multistream_rollout_is_bit_identical_to_eager(streams, depth_mode, same_depth_path):
    """hipGraph replay (forked streams / one stream / three graphs on two streams) must not change a single
    bit of actions / states / maps - with the depth encoder as the launch chain (0) and as the persistent launch (2)."""
    same_depth_path(depth_mode)
    from ivln_ce_amd.config import get_config
    from ivln_ce_amd.graphed import GraphedRollout
    from ivln_ce_amd.obs_transforms import GTSemanticsIterativeMapper
    from ivln_ce_amd.synthetic import SyntheticRollout

    dev = torch.device("cuda:0")
    pol = make_policy()
    cfg = get_config()
    B, steps = 4, 6
    roll = SyntheticRollout(B=B, seed=31)
    obs = [{k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in roll.step().items()} for _ in range(steps)]
    # eager
    tr_e = GTSemanticsIterativeMapper.from_config(cfg)
    rnn = torch.zeros(B, 2, 512, device=dev)
    prev = torch.zeros(B, 1, dtype=torch.long, device=dev)
    eager = []
    for o in obs:
        b = tr_e(dict(o))
        with torch.no_grad():
            a, rnn = pol.act(b, rnn, prev, b["not_done_masks"], deterministic=True)
        prev = a
        eager.append((a.clone(), rnn.clone(), b["occupancy_map"].clone(), b["semantic_map"].clone()))
    # graphed: the capture warm-up replays obs[0] a few times, so restart mapper + policy state afterwards
    tr_g = GTSemanticsIterativeMapper.from_config(cfg)
    runner = GraphedRollout(pol, [tr_g], obs[0], deterministic=True, streams=streams)
    tr_g.mapping_module.reset()
    runner.reset_state()
    for t, o in enumerate(obs):
        a = runner.step(o)
        torch.cuda.synchronize()
        mem = tr_g.mapping_module.map_memory
        assert torch.equal(a, eager[t][0]), f"actions step {t}"
        assert torch.equal(runner.rnn_states, eager[t][1]), f"rnn step {t}"
        assert torch.equal(mem.occupancy, eager[t][2]) and torch.equal(mem.semantic, eager[t][3]), f"maps step {t}"
    tr_g.mapping_module.check_status()


@pytest.mark.parametrize("lens", [[1, 200, 37, 200], [200], [1, 1]])
def test_act_instruction_length_extremes_match_oracle(lens):
    """Instruction of one token and of the maximum 200 tokens (no padding at all) in the same batch: packed
    bi-LSTM lengths, the -1e8 text-attention mask and the zero-padded outputs against the oracle."""
    from det_init import det_fill

    from ivln_ce_amd.synthetic import SyntheticRollout
    from oracle.policy_ref import MapCMAPolicyRef

    torch.set_num_threads(8)
    B = len(lens)
    pol = make_policy()
    ref = det_fill(MapCMAPolicyRef(), seed=0).eval()
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(sum(lens))
    obs = SyntheticRollout(B=B, seed=9).step()
    instr = torch.zeros(B, 200, dtype=torch.int64)
    for b, n in enumerate(lens):
        instr[b, :n] = torch.randint(2, 2504, (n,), generator=g)
    obs["instruction"] = instr
    obs["occupancy_map"] = (torch.rand(B, 64, 64, generator=g) < 0.4).to(torch.uint8)
    obs["semantic_map"] = (torch.randint(0, 13, (B, 64, 64), generator=g) * obs["occupancy_map"]).to(torch.uint8)
    rnn = 0.1 * torch.randn(B, 2, 512, generator=g)
    prev = torch.randint(0, 4, (B, 1), generator=g)
    masks = torch.ones(B, 1, dtype=torch.uint8)
    with torch.no_grad():
        lr, sr, fr = ref.logits(obs, rnn, prev, masks)
        dobs = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in obs.items()}
        f, s = pol.net(dobs, rnn.to(dev), prev.to(dev), masks.to(dev))
        lg = pol.action_distribution.raw_logits(f)
    assert float((f.cpu() - fr).abs().max()) < ATOL
    assert float((s.cpu() - sr).abs().max()) < ATOL
    assert float((lg.cpu() - lr).abs().max()) < 1e-4


@pytest.mark.parametrize("B,lens", [(3, None), (8, None), (4, [1, 200, 17, 80]), (20, None)])
def test_fused_head_matches_unfused_chain_and_oracle(B, lens):
    """ivln_cma_step_fwd (folded operands, five phase kernels) against the unfused op chain and the torch-CPU
    oracle: features, both recurrent states, logits.  The folds reorder sums (state . (W_q^T text_k) instead of
    (W_q state) . text_k), so fused vs unfused agree to rounding (1e-5), each within 2e-4 of the oracle."""
    from det_init import det_fill

    from ivln_ce_amd import ops
    from ivln_ce_amd.synthetic import SyntheticRollout
    from oracle.policy_ref import MapCMAPolicyRef

    torch.set_num_threads(8)
    pol = make_policy()
    ref = det_fill(MapCMAPolicyRef(), seed=0).eval()
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(40 + B)
    obs = SyntheticRollout(B=B, seed=300 + B, n_tokens=64).step()
    if lens is not None:
        for b, n in enumerate(lens):
            obs["instruction"][b] = 0
            obs["instruction"][b, :n] = torch.randint(2, 2504, (n,), generator=g)
    obs["occupancy_map"] = (torch.rand(B, 64, 64, generator=g) < 0.4).to(torch.uint8)
    obs["semantic_map"] = (torch.randint(0, 13, (B, 64, 64), generator=g) * obs["occupancy_map"]).to(torch.uint8)
    rnn = 0.2 * torch.randn(B, 2, 512, generator=g)
    prev = torch.randint(0, 4, (B, 1), generator=g)
    masks = (torch.rand(B, 1, generator=g) < 0.7).to(torch.uint8)
    dobs = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in obs.items()}
    out = {}
    saved = ops.CMA_STEP_MODE
    try:
        for mode in (-1, 0):
            ops.CMA_STEP_MODE = mode
            with torch.no_grad():
                f, s = pol.net(dobs, rnn.to(dev), prev.to(dev), masks.to(dev))
                lg = pol.action_distribution.raw_logits(f)
            out[mode] = (f.cpu(), s.cpu(), lg.cpu())
    finally:
        ops.CMA_STEP_MODE = saved
    for k in range(3):
        assert float((out[0][k] - out[-1][k]).abs().max()) < 2e-5, k
    with torch.no_grad():
        lr, sr, fr = ref.logits(obs, rnn, prev, masks)
    for mode in (-1, 0):
        assert float((out[mode][0] - fr).abs().max()) < ATOL
        assert float((out[mode][1] - sr).abs().max()) < ATOL
        assert float((out[mode][2] - lr).abs().max()) < 1e-4


def test_instruction_front_end_folded_into_a_table_lookup_is_the_same_encoder():
    """Inference fold: embedding lookup + the two W_ih projections of the bi-LSTM = one lookup in
    table[v] = E[v] . [W_ih ; W_ih_rev]^T + b (ivln_embed_gates_f32).  Same outputs and the same `lengths` as the
    unfolded launches, including the reference's quirk that a token counts only if its EMBEDDING row has a non-zero
    element (instruction_encoder.py:70-78) - rows 0 and 7 of the table are zeroed here - and the table is rebuilt
    when a weight changes."""
    from ivln_ce_amd import ops

    pol = make_policy()
    enc = pol.net.instruction_encoder
    with torch.no_grad():
        enc.embedding_layer.weight[0].zero_()
        enc.embedding_layer.weight[7].zero_()
    g = torch.Generator().manual_seed(4)
    tokens = torch.randint(1, 2504, (5, 200), generator=g)
    tokens[0, 60:] = 0
    tokens[1, 10:] = 0
    tokens[2, 3] = 7      # a zero-embedding token inside the sentence: not counted
    tokens[3, :] = 0      # empty instruction
    obs = {"instruction": tokens.to("cuda:0")}
    old = ops.FOLD_INSTRUCTION_GATES
    try:
        ops.FOLD_INSTRUCTION_GATES = False
        with torch.no_grad():
            ref, ref_len = enc(obs)
        ops.FOLD_INSTRUCTION_GATES = True
        with torch.no_grad():
            got, got_len = enc(obs)
            got, got_len = got.clone(), got_len.clone()  # (no-grad outputs live in the per-episode cache's buffers: valid until the next call)
            assert enc.__dict__.get("_gate_cache") is not None
            assert torch.equal(got_len, ref_len) and ref_len.tolist()[:4] == [60, 10, 199, 0]
            assert float((got - ref).abs().max()) < 2e-6
            enc.encoder_rnn.weight_ih_l0.mul_(1.5)  # a weight update invalidates the table
            got2, _ = enc(obs)
            ops.FOLD_INSTRUCTION_GATES = False
            ref2, _ = enc(obs)
        assert float((got2 - ref2).abs().max()) < 2e-6 and float((got2 - got).abs().max()) > 1e-3
    finally:
        ops.FOLD_INSTRUCTION_GATES = old


def _episode_script(B, steps, seed=5):
    """Observations of a rollout in which the instruction behaves like an episode's: constant per env, except
    step 10: env 1 starts a new episode (mask 0, NEW tokens); step 17: env 2's tokens change in mid-episode (no reset);
    step 22: env 0 starts a new episode with the SAME tokens; step 25: env 3 gets a shorter instruction."""
    from ivln_ce_amd.synthetic import SyntheticRollout

    dev = torch.device("cuda:0")
    roll = SyntheticRollout(B=B, seed=seed)
    g = torch.Generator().manual_seed(seed + 1)
    obs, expect_dirty = [], []
    for t in range(steps):
        dirty = [t == 0] * B
        if t == 10:
            roll.instruction[1, :80] = torch.randint(2, 2504, (80,), generator=g)
            dirty[1] = True
        if t == 17:
            roll.instruction[2, 5] = 1234 if int(roll.instruction[2, 5]) != 1234 else 1235
            dirty[2] = True
        if t == 25:
            roll.instruction[3, 40:] = 0
            dirty[3] = True
        o = roll.step()
        if t == 10:
            o["not_done_masks"][1] = 0
        if t == 22:
            o["not_done_masks"][0] = 0
        obs.append({k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in o.items()})
        expect_dirty.append([int(d) for d in dirty])
    return obs, expect_dirty


def _eager_rollout(pol, obs, check_dirty=None):
    from ivln_ce_amd.config import get_config
    from ivln_ce_amd.obs_transforms import GTSemanticsIterativeMapper

    dev = torch.device("cuda:0")
    B = obs[0]["depth"].shape[0]
    tr = GTSemanticsIterativeMapper.from_config(get_config())
    rnn = torch.zeros(B, 2, 512, device=dev)
    prev = torch.zeros(B, 1, dtype=torch.long, device=dev)
    out = []
    for t, o in enumerate(obs):
        b = tr(dict(o))
        with torch.no_grad():
            a, rnn = pol.act(b, rnn, prev, b["not_done_masks"], deterministic=True)
        prev = a
        out.append((a.clone(), rnn.clone()))
        if check_dirty is not None:
            cache = pol.net.instruction_encoder.last_cache
            assert cache is not None and cache.dirty.tolist() == check_dirty[t], (t, cache.dirty.tolist(), check_dirty[t])
    return out


def test_instruction_encoding_cached_per_episode_is_bit_identical_to_always_recompute(same_depth_path):
    """the reference re-runs the instruction bi-LSTM on the same tokens at every step
    (map_cma_policy.py:293, instruction_encoder.py:72-94).  Here a row is re-encoded only when its tokens differ from the
    ones it encoded last (decided on the device: k_embed_gates compares, k_lstm_bidir and the fold conv read the flags).
    Over a 30-step rollout with an episode change, a mid-episode token change, a reset with unchanged tokens and a
    shortened instruction: the same actions and recurrent states, bit for bit, as re-encoding at every step - eagerly
    and as the replayed split graphs - and exactly the expected rows are re-encoded."""
    same_depth_path(0)
    from ivln_ce_amd import ops
    from ivln_ce_amd.config import get_config
    from ivln_ce_amd.graphed import GraphedRollout
    from ivln_ce_amd.obs_transforms import GTSemanticsIterativeMapper

    pol = make_policy()
    B, steps = 4, 30
    obs, expect_dirty = _episode_script(B, steps)
    old = ops.CACHE_INSTRUCTION
    try:
        ops.CACHE_INSTRUCTION = False
        ref = _eager_rollout(pol, obs)
        assert pol.net.instruction_encoder.last_cache is None
        ops.CACHE_INSTRUCTION = True
        got = _eager_rollout(pol, obs, check_dirty=expect_dirty)
        for t in range(steps):
            assert torch.equal(got[t][0], ref[t][0]) and torch.equal(got[t][1], ref[t][1]), f"eager step {t}"
        # replayed: the cache's buffers are shared by the graphs of both phases; the capture's warm-up steps leave obs[0]'s
        # tokens in the cache, the script's first step is all-dirty anyway (reset_state + a fresh mapper)
        tr_g = GTSemanticsIterativeMapper.from_config(get_config())
        runner = GraphedRollout(pol, [tr_g], obs[0], deterministic=True, streams="split")
        tr_g.mapping_module.reset()
        runner.reset_state()
        cache = pol.net.instruction_encoder.step_cache(B, 200, torch.device("cuda:0"))
        cache.invalidate()
        for t, o in enumerate(obs):
            a = runner.step(o)
            torch.cuda.synchronize()
            assert torch.equal(a, ref[t][0]) and torch.equal(runner.rnn_states, ref[t][1]), f"replayed step {t}"
            assert cache.dirty.tolist() == expect_dirty[t], (t, cache.dirty.tolist())
    finally:
        ops.CACHE_INSTRUCTION = old


def test_instruction_cache_follows_weight_changes_and_batch_rows():
    """The cache is keyed by the tokens, so anything else that changes the encoding has to invalidate it: a weight update
    (FlatAdam.step writes through raw pointers: ops.invalidate_step_caches), an in-place edit of an LSTM weight, and rows
    that move when envs pause (quirks Q5 / Q12: the batch is compacted - the tokens of a row then differ)."""
    from ivln_ce_amd import ops

    pol = make_policy()
    enc = pol.net.instruction_encoder
    g = torch.Generator().manual_seed(9)
    tokens = torch.zeros(4, 200, dtype=torch.long)
    tokens[:, :50] = torch.randint(2, 2504, (4, 50), generator=g)
    obs = {"instruction": tokens.to("cuda:0")}

    def fresh(o):
        old = ops.CACHE_INSTRUCTION
        ops.CACHE_INSTRUCTION = False
        try:
            with torch.no_grad():
                r, ln = enc(o)
            return r.clone(), ln.clone()
        finally:
            ops.CACHE_INSTRUCTION = old

    with torch.no_grad():
        a, _ = enc(obs)
        assert enc.last_cache.dirty.tolist() == [1, 1, 1, 1] and torch.equal(a, fresh(obs)[0])
        a, _ = enc(obs)
        assert enc.last_cache.dirty.tolist() == [0, 0, 0, 0] and torch.equal(a, fresh(obs)[0])
        enc.encoder_rnn.weight_hh_l0.mul_(1.1)  # an in-place weight edit: new versions, every row re-encoded
        a, _ = enc(obs)
        assert enc.last_cache.dirty.tolist() == [1, 1, 1, 1] and torch.equal(a, fresh(obs)[0])
        ops.invalidate_step_caches()  # what FlatAdam.step calls
        a, _ = enc(obs)
        assert enc.last_cache.dirty.tolist() == [1, 1, 1, 1]
        # env 1 pauses: rows 2, 3 move up (a 3-row batch is another cache), then the 4-row batch comes back permuted
        o3 = {"instruction": obs["instruction"][[0, 2, 3]].contiguous()}
        a3, _ = enc(o3)
        assert enc.last_cache.rows == 3 and torch.equal(a3, fresh(o3)[0])
        o4 = {"instruction": obs["instruction"][[0, 2, 1, 3]].contiguous()}
        a4, l4 = enc(o4)
        assert enc.last_cache.dirty.tolist() == [0, 1, 1, 0]
        r4, rl4 = fresh(o4)
        assert torch.equal(a4, r4) and torch.equal(l4, rl4)


def test_replayed_step_whose_persistent_encoder_timed_out_is_redone_on_the_launch_chain(same_depth_path):
    """The loops' guard (trainers._persistent_guard) on a replayed rollout: at step 3 a workgroup of the side graph's
    persistent depth encoder "is not resident" (test hook, tests/test_gpu_depth_net.py), its barriers time out and the step's
    features are void.  The host sees the pinned flag after the step's synchronisation, retires the plan and computes the
    policy half of the SAME step again eagerly (GraphedRollout.redo_last_step_eagerly); the rollout goes on on the launch
    chain.  Against the same script on a healthy runner: the same actions, recurrent states within the two encoders'
    distance (2e-4)."""
    same_depth_path(2)
    from ivln_ce_amd import depth_net
    from ivln_ce_amd.config import get_config
    from ivln_ce_amd.graphed import GraphedRollout
    from ivln_ce_amd.obs_transforms import GTSemanticsIterativeMapper

    pol = make_policy()
    for p in pol.net.depth_encoder.visual_encoder.parameters():
        p.requires_grad_(False)
    B, steps = 4, 7
    obs, _ = _episode_script(B, steps, seed=11)

    def run(fail_at, recapture=False, pol=pol):
        tr = GTSemanticsIterativeMapper.from_config(get_config())
        runner = GraphedRollout(pol, [tr], obs[0], deterministic=True, streams="split")
        tr.mapping_module.reset()
        runner.reset_state()
        out, rnn, prev = [], None, None
        for t, o in enumerate(obs):
            if runner is None and recapture:
                # what the loops do on their next step (trainers._make_runner with _rewarm_next_capture): a new capture whose
                # warm-up runs the launch chain on the capture streams WITHOUT stepping the mapper, seeded with the carried state
                n_before = tr.mapping_module.status()
                runner = GraphedRollout(pol, [tr], o, deterministic=True, streams="split", warmup=1, warmup_mapper=False)
                assert tr.mapping_module.status() == n_before, "the re-capture stepped the world cloud"
                runner.rnn[runner.phase].copy_(rnn)
                runner.prev[runner.phase].copy_(prev)
            if runner is not None:
                if t == fail_at:
                    plan = depth_net.plan_for(pol.net.depth_encoder.visual_encoder, torch.device("cuda:0"))
                    with torch.cuda.stream(runner.sA):
                        plan.stream_state()[1][257] = 1
                a = runner.step(o)
                torch.cuda.synchronize()
                rnn = runner.rnn_states
                if depth_net.any_failed():
                    assert t == fail_at
                    assert depth_net.recover_all() == 1
                    a = runner.redo_last_step_eagerly()
                    torch.cuda.synchronize()
                    rnn, prev, runner = runner.rnn_states.clone(), a.clone(), None
            else:  # the rest of the rollout: eager steps, the mapper keeps its state, the encoder runs the launch chain
                b = tr(dict({k: (v.clone() if torch.is_tensor(v) else v) for k, v in o.items()}))
                with torch.no_grad():
                    a, rnn = pol.act(b, rnn, prev, b["not_done_masks"], deterministic=True)
                prev = a.clone()
            out.append((a.clone(), rnn.clone()))
        return out

    healthy = run(-1)
    assert not depth_net.any_failed()
    hurt = run(3)
    for t in range(steps):
        assert torch.equal(hurt[t][0], healthy[t][0]), f"actions step {t}"
        assert float((hurt[t][1] - healthy[t][1]).abs().max()) < 2e-4, f"rnn step {t}"
    # ... and with the rest of the rollout REPLAYED from a capture made after the recovery (the new graphs record
    # the launch chain on streams that have to have run it eagerly first): the same steps as the eager continuation, bit for bit
    pol2 = make_policy()  # (the same deterministic weights; a new encoder object = a plan that has not been retired)
    for p in pol2.net.depth_encoder.visual_encoder.parameters():
        p.requires_grad_(False)
    again = run(3, recapture=True, pol=pol2)
    assert not depth_net.any_failed()
    for t in range(steps):
        assert torch.equal(again[t][0], hurt[t][0]), f"actions step {t} (recaptured)"
        assert torch.equal(again[t][1], hurt[t][1]), f"rnn step {t} (recaptured)"
