"""GPU parity of the HIP backward / loss / Adam against the golden produced by the reference's
own MapCMAPolicy.build_distribution + base_il_trainer loss + autograd (policy_update.npz), and
per-kernel checks of the loss and optimizer kernels against torch.  Tolerances: loss 2e-5 abs,
gradient norms 5e-4 relative, sampled full gradients 1e-3 relative + 2e-6 abs (fp32, order of
summation differs)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
sys.path.insert(0, os.path.dirname(__file__))
G = os.path.join(os.path.dirname(__file__), "golden")
DEV = torch.device("cuda:0")


def _batch(g):
    obs = {k: torch.from_numpy(g[k2]).to(DEV) for k, k2 in [
        ("depth_features", "depth_features"), ("occupancy_map", "occ"), ("semantic_map", "sem"),
        ("instruction", "instruction"), ("progress", "progress")]}
    return (obs, torch.from_numpy(g["prev"]).to(DEV), torch.from_numpy(g["not_done"]).to(DEV),
            torch.from_numpy(g["targets"]).to(DEV), torch.from_numpy(g["weights"]).to(DEV))


def _check_grads(pol, g, log):
    params = dict(pol.named_parameters())
    bad = []
    for k in g.files:
        if k.startswith("gradnorm/"):
            ref = float(g[k])
            p = params[k[9:]]
            got = float(p.grad.norm()) if p.grad is not None else float("nan")
            rel = abs(got - ref) / max(1e-6, abs(ref))
            log.append(f"{k[9:]}: |g|={got:.6e} ref={ref:.6e} rel={rel:.2e}")
            # conv biases in front of a train-mode BatchNorm have an analytically ZERO gradient; both the
            # reference (4.7e-6) and the HIP path (6.6e-8) only hold rounding noise there
            noise = 2e-5 if (".conv.0.bias" in k and "map_encoder" in k) else 1e-7
            if not (rel < 5e-4 or abs(got - ref) < noise):
                bad.append(k)
        elif k.startswith("grad/"):
            got = params[k[5:]].grad.cpu().numpy()
            atol = 2e-5 if (".conv.0.bias" in k and "map_encoder" in k) else 2e-6
            if not np.allclose(got, g[k], atol=atol, rtol=1e-3):
                bad.append(k + f" maxerr={np.abs(got - g[k]).max():.3e}")
    return bad


def test_reference_style_update_matches_golden():
    """The reference's own _update_agent body (torch loss + loss.backward()) on the HIP policy."""
    from test_gpu_policy import make_policy

    from ivln_ce_amd.aux_losses import AuxLosses

    g = np.load(os.path.join(G, "policy_update.npz"))
    pol = make_policy(use_pm=True, train=True)
    obs, prev, nd, tgt, w = _batch(g)
    T, N = tgt.shape
    AuxLosses.activate()
    AuxLosses.clear()
    h0 = torch.zeros(N, 2, 512, device=DEV)
    dist, _ = pol.build_distribution(obs, h0, prev, nd)
    logits = dist.logits.view(T, N, -1)
    ce = F.cross_entropy(logits.permute(0, 2, 1), tgt, reduction="none")
    action_loss = ((w * ce).sum(0) / w.sum(0)).mean()
    aux = AuxLosses.reduce((w > 0).view(-1))
    loss = action_loss + aux
    loss.backward()
    AuxLosses.deactivate()
    log = [f"loss {float(loss):.7f} ref {float(g['loss']):.7f}; action {float(action_loss):.7f} ref "
           f"{float(g['action_loss']):.7f}; aux {float(aux):.7f} ref {float(g['aux_loss']):.7f}"]
    assert np.allclose(logits.detach().cpu().numpy(), g["logits"], atol=1e-5), "logits"
    bad = _check_grads(pol, g, log)
    os.makedirs("gpurun_out", exist_ok=True)
    open("gpurun_out/train_parity.log", "w").write("\n".join(log) + "\n")
    assert abs(float(loss) - float(g["loss"])) < 2e-5, log[0]
    assert abs(float(aux) - float(g["aux_loss"])) < 2e-5, log[0]
    assert not bad, "\n".join(bad + log)
    sd = pol.state_dict()
    for k in g.files:
        if k.startswith("post/"):
            assert np.allclose(sd[k[5:]].cpu().numpy(), g[k], atol=1e-6, rtol=1e-5), k


def test_ce_iw_loss_kernel_matches_torch():
    from ivln_ce_amd import ops

    gen = torch.Generator().manual_seed(3)
    T, N, A = 7, 5, 4
    logits = torch.randn(T, N, A, generator=gen, requires_grad=True)
    tgt = torch.randint(0, A, (T, N), generator=gen)
    w = torch.where(torch.rand(T, N, generator=gen) < 0.3, torch.tensor(3.2), torch.tensor(1.0))
    w[5:, 1] = 0
    ce = F.cross_entropy(logits.permute(0, 2, 1), tgt, reduction="none")
    ref = ((w * ce).sum(0) / w.sum(0)).mean()
    ref.backward()
    loss, dl = ops.ce_iw_loss(logits.detach().to(DEV), tgt.to(DEV), w.to(DEV))
    assert abs(float(loss) - float(ref)) < 1e-6
    assert torch.allclose(dl.cpu(), logits.grad, atol=1e-7, rtol=1e-5)


def test_adam_kernel_matches_torch_optim():
    from ivln_ce_amd import ops

    gen = torch.Generator().manual_seed(4)
    n = 10007
    p0 = torch.randn(n, generator=gen)
    p_ref = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([p_ref], lr=2.5e-4)
    p = p0.clone().to(DEV)
    m, v = torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    for step in range(1, 4):
        gr = torch.randn(n, generator=gen)
        p_ref.grad = gr.clone()
        opt.step()
        gbuf = gr.clone().to(DEV)
        ops.adam_step(p, gbuf, m, v, 2.5e-4, step)
        assert float(gbuf.abs().max()) == 0.0  # zero_grad fused
    assert torch.allclose(p.cpu(), p_ref.detach(), atol=1e-7, rtol=1e-6)


def test_hip_update_agent_matches_reference_loss_and_moves_params():
    """All-HIP update (fused CE + flat Adam): same loss as the golden, params change like torch Adam."""
    from test_gpu_policy import make_policy

    from ivln_ce_amd.trainers import FlatAdam, update_agent

    g = np.load(os.path.join(G, "policy_update.npz"))
    pol = make_policy(use_pm=True, train=True)
    obs, prev, nd, tgt, w = _batch(g)
    opt = FlatAdam(pol, lr=2.5e-4)
    before = {k: v.detach().clone() for k, v in pol.named_parameters() if v.requires_grad}
    from ivln_ce_amd.aux_losses import AuxLosses

    AuxLosses.activate()  # the trainer does this when MODEL.PROGRESS_MONITOR.use (dagger_trainer.py:556-557)
    loss, action_loss, aux = update_agent(pol, opt, obs, prev, nd, tgt, w, hidden_size=512)
    AuxLosses.deactivate()
    assert abs(loss - float(g["loss"])) < 2e-5 and abs(action_loss - float(g["action_loss"])) < 2e-5
    assert abs(aux - float(g["aux_loss"])) < 2e-5
    # first Adam step moves every element by ~lr * sign(grad)
    moved = 0
    for k, v in pol.named_parameters():
        if v.requires_grad:
            d = (v.detach() - before[k]).abs().max().item()
            assert d <= 2.5e-4 * 1.001 + 1e-9, k
            moved += d > 0
    assert moved > 40


def test_update_whose_persistent_gru_timed_out_is_skipped_on_the_device_and_run_again():
    """VERDICT r4 item 7 for the update: the sequence GRUs run as ONE persistent launch each (csrc/gru_seq.hip) whose
    workgroups exchange through bounded spins.  When a workgroup never arrives (word 49 of the sync workspace: workgroup 1
    leaves at entry, what a workgroup the dispatcher could not place looks like) the launch times out and the update's
    gradients are void.  The Adam kernel then skips the step ON THE DEVICE (its guard word = the sticky error), the host
    sees the word after the loss read-back, clears it, switches to per-timestep GRU launches for the rest of the run and
    computes the update again: losses and parameters are those of the undisturbed update (the two GRU forms differ by
    2e-7), nothing raises."""
    from test_gpu_policy import make_policy

    from ivln_ce_amd import ops
    from ivln_ce_amd.aux_losses import AuxLosses
    from ivln_ce_amd.trainers import FlatAdam, update_agent

    g = np.load(os.path.join(G, "policy_update.npz"))
    obs, prev, nd, tgt, w = _batch(g)
    old = ops.SEQ_PERSISTENT
    AuxLosses.activate()
    try:
        ops.SEQ_PERSISTENT = True
        pol_a, pol_b = make_policy(use_pm=True, train=True), make_policy(use_pm=True, train=True)
        opt_a, opt_b = FlatAdam(pol_a, lr=2.5e-4), FlatAdam(pol_b, lr=2.5e-4)
        ra = update_agent(pol_a, opt_a, obs, prev, nd, tgt, w, hidden_size=512)
        torch.cuda.synchronize()
        assert ops._seq_sync_ws, "the update ran its GRUs as persistent launches"
        for ws in ops._seq_sync_ws.values():
            ws[49] = 1
        rb = update_agent(pol_b, opt_b, obs, prev, nd, tgt, w, hidden_size=512)
        torch.cuda.synchronize()
        assert ops.SEQ_PERSISTENT is False and not ops.seq_failed()
        assert all(int(ws[49]) == 0 for ws in ops._seq_sync_ws.values())
        assert opt_b.step_count == opt_a.step_count == 1
        for x, y in zip(ra, rb):
            assert abs(x - y) < 2e-6, (ra, rb)
        worst = max(float((p.detach() - q.detach()).abs().max()) for p, q in zip(pol_a.parameters(), pol_b.parameters()))
        assert worst < 2e-5, worst  # (one Adam step of 2.5e-4; elements whose gradient is below Adam's eps move by lr * g / eps: 2e-7-relative GRU differences show there)
        # ... and the run goes on: the next update uses the per-timestep launches and tracks the undisturbed policy
        ra2 = update_agent(pol_a, opt_a, obs, prev, nd, tgt, w, hidden_size=512)
        rb2 = update_agent(pol_b, opt_b, obs, prev, nd, tgt, w, hidden_size=512)
        assert abs(ra2[0] - rb2[0]) < 1e-4
    finally:
        AuxLosses.deactivate()
        ops.SEQ_PERSISTENT = old


def _tiny_cfg(tmp_path, trainer="dagger", pm=True):
    from ivln_ce_amd.config import get_config

    return get_config(opts=[
        "TRAINER_NAME", trainer, "NUM_ENVIRONMENTS", 2, "MODEL.policy_name", "MapCMAPolicy",
        "MODEL.INSTRUCTION_ENCODER.use_pretrained_embeddings", False, "MODEL.DEPTH_ENCODER.ddppo_checkpoint", "NONE",
        "MODEL.PROGRESS_MONITOR.use", pm, "RL.POLICY.OBS_TRANSFORMS.ENABLED_TRANSFORMS", ["GTSemanticsIterativeMapper"],
        "IL.DAGGER.iterations", 1, "IL.DAGGER.update_size", 4, "IL.DAGGER.p", 0.5, "IL.epochs", 1, "IL.batch_size", 2,
        "IL.DAGGER.lmdb_features_dir", str(tmp_path / "traj"), "CHECKPOINT_FOLDER", str(tmp_path / "ckpt"),
        "RESULTS_DIR", str(tmp_path / "res"), "EVAL_CKPT_PATH_DIR", str(tmp_path / "ckpt"),
    ])


@pytest.mark.parametrize("trainer", ["dagger", "iterative_collection_dagger"])
def test_dagger_trainer_end_to_end(tmp_path, trainer):
    """rollout with beta-mixed expert -> trajectory store -> collate -> HIP update -> checkpoint -> eval."""
    import ivln_ce_amd  # noqa: F401
    from ivln_ce_amd import trainers  # noqa: F401
    from ivln_ce_amd.registry import baseline_registry

    torch.manual_seed(0)
    cfg = _tiny_cfg(tmp_path, trainer)
    tr = baseline_registry.get_trainer(trainer)(cfg)
    log = tr.train()
    assert len(log) >= 1 and all(np.isfinite(l["loss"]) for l in log)
    ck = torch.load(tmp_path / "ckpt" / "ckpt.0.pth", weights_only=False)
    assert set(ck.keys()) == {"state_dict", "config", "optim_state", "dagger_it", "epoch", "step_id"}
    assert "net.map_encoder.cnn.0.conv.0.weight" in ck["state_dict"]
    tr2 = baseline_registry.get_trainer(trainer)(cfg)
    res = tr2.eval()[0]
    assert res["episodes"] == 16 and 0.0 < res["t_ndtw"] <= 1.0
    assert os.path.exists(tmp_path / "res" / "stats_ckpt_0_val_seen.json")


def test_eval_with_graph_replay_matches_eager_eval(tmp_path, same_depth_path):
    """The trainer's eval loop replays mapper + policy.act as captured graphs by default
    (EVAL.USE_HIP_GRAPH); per-episode stats and t-nDTW must equal the eager loop's exactly, including across
    the re-captures when envs run out of episodes and are paused."""
    import ivln_ce_amd  # noqa: F401
    from ivln_ce_amd import trainers  # noqa: F401
    from ivln_ce_amd.config import get_config
    from ivln_ce_amd.registry import baseline_registry

    same_depth_path(2)  # the persistent depth encoder in the replayed and in the eager loop
    out = {}
    for mode in (True, False):
        torch.manual_seed(0)
        cfg = get_config(opts=[
            "TRAINER_NAME", "dagger", "NUM_ENVIRONMENTS", 3, "MODEL.policy_name", "MapCMAPolicy",
            "MODEL.INSTRUCTION_ENCODER.use_pretrained_embeddings", False, "MODEL.DEPTH_ENCODER.ddppo_checkpoint", "NONE",
            "RL.POLICY.OBS_TRANSFORMS.ENABLED_TRANSFORMS", ["GTSemanticsIterativeMapper"],
            "RESULTS_DIR", str(tmp_path / f"res{int(mode)}"), "EVAL_CKPT_PATH_DIR", str(tmp_path / "none.pth"),
            "EVAL.USE_HIP_GRAPH", mode, "EVAL.SAVE_RESULTS", False,
        ])
        tr = baseline_registry.get_trainer("dagger")(cfg)
        res = tr._eval_checkpoint(str(tmp_path / "none.pth"))
        res.pop("eval_seconds")
        out[mode] = res
    assert out[True] == out[False], f"graph {out[True]} vs eager {out[False]}"
    assert out[True]["episodes"] > 0


def test_eval_with_predicted_semantics_graph_replay_matches_eager_eval(tmp_path, same_depth_path):
    """BASELINE configs[2] through the trainer: `PredictedSemanticsIterativeMapper` (RedNet -> mapper) + `MapCMAPolicy.act`
    replayed as captured graphs - the main graph in pieces, the depth encoder released behind RedNet's layer 3, the mapper's
    label-free half at the head of the side graph (graphed.py, round 6) - including the captures WITHOUT a warm-up that follow
    when envs run out of episodes and pause.  Per-episode stats and t-nDTW equal the eager loop's exactly (the eager loop runs
    the depth encoder as conv + GroupNorm pairs too: the same kernels the capture chooses beside RedNet)."""
    import ivln_ce_amd  # noqa: F401
    from ivln_ce_amd import ops, trainers  # noqa: F401
    from ivln_ce_amd.config import get_config
    from ivln_ce_amd.registry import baseline_registry

    same_depth_path(0)
    chain, ops.CHAIN_GN_CONV = ops.CHAIN_GN_CONV, False
    out = {}
    try:
        for mode in (True, False):
            torch.manual_seed(0)
            cfg = get_config(opts=[
                "TRAINER_NAME", "dagger", "NUM_ENVIRONMENTS", 3, "MODEL.policy_name", "MapCMAPolicy",
                "MODEL.INSTRUCTION_ENCODER.use_pretrained_embeddings", False, "MODEL.DEPTH_ENCODER.ddppo_checkpoint", "NONE",
                "RL.POLICY.OBS_TRANSFORMS.ENABLED_TRANSFORMS", ["PredictedSemanticsIterativeMapper"],
                "RESULTS_DIR", str(tmp_path / f"res{int(mode)}"), "EVAL_CKPT_PATH_DIR", str(tmp_path / "none.pth"),
                "EVAL.USE_HIP_GRAPH", mode, "EVAL.SAVE_RESULTS", False,
            ])
            tr = baseline_registry.get_trainer("dagger")(cfg)
            res = tr._eval_checkpoint(str(tmp_path / "none.pth"))
            res.pop("eval_seconds")
            out[mode] = res
    finally:
        ops.CHAIN_GN_CONV = chain
    assert out[True] == out[False], f"graph {out[True]} vs eager {out[False]}"
    assert out[True]["episodes"] > 0


@pytest.mark.parametrize("policy", ["MapCMAPolicy", "LatentCMAPolicy"])
def test_iterative_dagger_trainer_end_to_end(tmp_path, policy):
    """iterative_dagger: tour-by-tour collection (tour table stored as record 0) -> TourSampler batches -> HIP
    updates with the recurrent state carried between batches -> checkpoint -> iterative eval report files."""
    import ivln_ce_amd  # noqa: F401
    from ivln_ce_amd import trainers  # noqa: F401
    from ivln_ce_amd.config import get_config
    from ivln_ce_amd.registry import baseline_registry

    torch.manual_seed(0)
    np.random.seed(0)
    latent = policy == "LatentCMAPolicy"
    opts = [
        "TRAINER_NAME", "iterative_dagger", "NUM_ENVIRONMENTS", 2, "MODEL.policy_name", policy,
        "MODEL.INSTRUCTION_ENCODER.use_pretrained_embeddings", False, "MODEL.DEPTH_ENCODER.ddppo_checkpoint", "NONE",
        "RL.POLICY.OBS_TRANSFORMS.ENABLED_TRANSFORMS", [] if latent else ["GTSemanticsIterativeMapper"],
        "IL.DAGGER.iterations", 1, "IL.DAGGER.update_size", 12, "IL.DAGGER.p", 1.0, "IL.epochs", 2, "IL.batch_size", 2,
        "IL.DAGGER.lmdb_features_dir", str(tmp_path / "traj"), "CHECKPOINT_FOLDER", str(tmp_path / "ckpt"),
        "RESULTS_DIR", str(tmp_path / "res"), "EVAL_CKPT_PATH_DIR", str(tmp_path / "ckpt"),
        "TASK_CONFIG.ENVIRONMENT.ITERATIVE.ENABLED", True,
    ]
    if latent:
        opts += ["MODEL.tour_memory_variant", True, "MODEL.memory_at_end", True]
    cfg = get_config(opts=opts)
    tr = baseline_registry.get_trainer("iterative_dagger")(cfg)
    before = None
    log = tr.train()
    assert len(log) >= 2 and all(np.isfinite(l["loss"]) for l in log), log
    table = tr.store.get_tour_index()
    # (a pass over the envs stores every episode that finished in it, so the count can exceed update_size)
    assert 12 <= sum(len(v) for v in table.values()) <= 13 and min(min(v) for v in table.values()) == 1
    obs, prev, expert = tr.store.get(1)
    assert "depth_features" in obs and ("rgb_features" in obs) == latent and "rgb" not in obs
    if latent:
        assert obs["rgb_features"].shape[1:] == (2048, 4, 4)
    assert os.path.exists(tmp_path / "ckpt" / "ckpt.1.pth")
    res = baseline_registry.get_trainer("iterative_dagger")(cfg).eval()[0]
    assert res["episodes"] == 16 and 0.0 < res["tndtw"] <= 1.0
    assert os.path.exists(tmp_path / "res" / "iterative_stats_ckpt_0_val_seen.json")
    import json

    all_stats = json.load(open(tmp_path / "res" / "iterative_all_stats_ckpt_0_val_seen.json"))
    assert sum(len(v) for v in all_stats.values()) == 16 and len(all_stats) == 6  # 2 envs x 3 tours


@pytest.mark.parametrize("reset", ["iterative", "episodic"])
def test_iterative_eval_graph_replay_matches_eager(tmp_path, reset, same_depth_path):
    """Iterative evaluation keeps the maps for a whole tour (mapper reset by the TOUR mask) while the policy state
    resets per episode: the captured step carries both masks and must reproduce the eager loop exactly."""
    import ivln_ce_amd  # noqa: F401
    from ivln_ce_amd import trainers  # noqa: F401
    from ivln_ce_amd.config import get_config
    from ivln_ce_amd.registry import baseline_registry

    same_depth_path(0 if reset == "iterative" else 2)
    out = {}
    for mode in (True, False):
        torch.manual_seed(0)
        cfg = get_config(opts=[
            "TRAINER_NAME", "dagger", "NUM_ENVIRONMENTS", 3, "MODEL.policy_name", "MapCMAPolicy",
            "MODEL.INSTRUCTION_ENCODER.use_pretrained_embeddings", False, "MODEL.DEPTH_ENCODER.ddppo_checkpoint", "NONE",
            "RL.POLICY.OBS_TRANSFORMS.ENABLED_TRANSFORMS", ["GTSemanticsIterativeMapper"],
            "RESULTS_DIR", str(tmp_path / f"res{int(mode)}"), "EVAL_CKPT_PATH_DIR", str(tmp_path / "none.pth"),
            "EVAL.USE_HIP_GRAPH", mode, "EVAL.SAVE_RESULTS", False, "EVAL.ITERATIVE_MAP_RESET", reset,
            "TASK_CONFIG.ENVIRONMENT.ITERATIVE.ENABLED", True,
        ])
        tr = baseline_registry.get_trainer("dagger")(cfg)
        res = tr._eval_checkpoint(str(tmp_path / "none.pth"))
        res.pop("eval_seconds")
        out[mode] = res
    assert out[True] == out[False], f"graph {out[True]} vs eager {out[False]}"
    assert out[True]["episodes"] > 0 and "tndtw" in out[True]


def test_update_with_host_trimmed_instruction_padding_is_the_same_update():
    """trainers.PrefetchLoader drops the all-padding tail of the token batch on the host
    (utils.trim_instruction_padding): the update must not notice - same loss, same gradients as with the full
    200-column batch (the dropped columns are masked attention positions and zero LSTM steps)."""
    from test_gpu_policy import make_policy

    from ivln_ce_amd.aux_losses import AuxLosses
    from ivln_ce_amd.trainers import FlatAdam, update_agent
    from ivln_ce_amd.utils import trim_instruction_padding

    g = np.load(os.path.join(G, "policy_update.npz"))
    host = {"instruction": torch.from_numpy(g["instruction"])}
    trimmed = trim_instruction_padding(host)["instruction"]
    longest = int((host["instruction"] != 0).sum(1).max())
    assert trimmed.shape[1] == min(200, -(-longest // 8) * 8) and torch.equal(trimmed, host["instruction"][:, :trimmed.shape[1]])
    short = {"instruction": host["instruction"].clone()}
    short["instruction"][:, 40:] = 0
    assert trim_instruction_padding(short)["instruction"].shape[1] == 40
    assert trim_instruction_padding({"instruction": host["instruction"].to(DEV)})["instruction"].shape[1] == 200  # device: untouched
    res = {}
    for name, instr in (("full", short["instruction"]), ("trim", trim_instruction_padding(short)["instruction"])):
        pol = make_policy(use_pm=True, train=True)
        obs, prev, nd, tgt, w = _batch(g)
        obs["instruction"] = instr.to(DEV)
        opt = FlatAdam(pol, lr=2.5e-4)
        AuxLosses.activate()
        try:
            out = update_agent(pol, opt, obs, prev, nd, tgt, w, hidden_size=512, step_grad=False)
        finally:
            AuxLosses.deactivate()
        res[name] = (out, {k: p.grad.detach().clone() for k, p in pol.named_parameters() if p.grad is not None})
    assert abs(res["full"][0][0] - res["trim"][0][0]) < 1e-6
    for k, gfull in res["full"][1].items():
        gt = res["trim"][1][k]
        assert torch.allclose(gfull, gt, rtol=1e-4, atol=1e-7), f"{k}: {float((gfull - gt).abs().max()):.3e}"


def _run_two_ranks_one_gpu(tmp_path, port, *args):
    """Two ranks sharing cuda:0 over gloo (RCCL refuses two ranks per device): control flow of the multi-rank
    paths - env sharding, FlatAdam's all-reduce, rank-0 gather - on the 1-GPU box.  Never a measurement."""
    from conftest import run_torchrun

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, IVLN_DIST_BACKEND="gloo", IVLN_ONE_DEVICE="1")
    return run_torchrun(2, os.path.join(root, "tools", "dist_smoke.py"), (str(tmp_path), *args), env=env)  # (`port`: historical)


@pytest.mark.gpu
def test_two_rank_eval_without_a_preceding_train(tmp_path):
    """eval() / _eval_checkpoint() must set up the process group themselves and rank 0 must report
    BOTH ranks' episodes."""
    r = _run_two_ranks_one_gpu(tmp_path, 29561, "dagger", "eval_only")
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "dist smoke ok: eval_only world 2 episodes 32" in r.stdout


@pytest.mark.gpu
def test_two_rank_dagger_train_keeps_replicas_identical(tmp_path):
    """FlatAdam.step with world > 1 (all-reduce + Adam with 1/world folded in), MIN-reduced batch count, then eval."""
    r = _run_two_ranks_one_gpu(tmp_path, 29563, "dagger")
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "dist smoke ok: dagger world 2" in r.stdout


@pytest.mark.gpu
def test_update_at_bench_shape_T64_N8_matches_oracle():
    """The benched update shape (T = 64 x N = 8 per GPU, configs[3]'s per-rank shard) against the torch-CPU
    oracle's autograd: loss, aux loss, logits and every parameter's gradient norm.  Unequal trajectory lengths
    (zero-weight padded tail rows), inflection weights, 80-token instructions trimmed on the host."""
    from det_init import det_fill
    from test_gpu_policy import make_policy

    from ivln_ce_amd.aux_losses import AuxLosses
    from ivln_ce_amd.utils import trim_instruction_padding
    from oracle.policy_ref import MapCMAPolicyRef

    torch.set_num_threads(8)
    T, N = 64, 8
    TN = T * N
    g = torch.Generator().manual_seed(64)
    lens = [64, 64, 51, 40, 64, 33, 64, 57]
    instr = torch.zeros(N, 200)
    for n in range(N):
        L = 80 - 7 * n
        instr[n, :L] = torch.randint(2, 2504, (L,), generator=g).float()
    obs = {"depth_features": torch.randn(TN, 128, 4, 4, generator=g),
           "occupancy_map": (torch.rand(TN, 64, 64, generator=g) < 0.3).float(),
           "semantic_map": torch.randint(0, 13, (TN, 64, 64), generator=g).float(),
           "instruction": instr.repeat(T, 1), "progress": torch.rand(TN, 1, generator=g)}
    prev = torch.randint(0, 4, (TN, 1), generator=g)
    nd = torch.ones(T, N, dtype=torch.uint8)
    nd[0] = 0
    nd = nd.view(-1, 1)
    tgt = torch.randint(0, 4, (T, N), generator=g)
    infl = torch.ones(T, N, dtype=torch.bool)
    infl[1:] = tgt[1:] != tgt[:-1]
    w = torch.where(infl, torch.tensor(3.2), torch.tensor(1.0))
    for n, L in enumerate(lens):  # collate_fn pads finished trajectories: obs 1.0, actions / weights 0
        w[L:, n] = 0
        tgt[L:, n] = 0
        for k in obs:
            v = obs[k].view(T, N, *obs[k].shape[1:])
            v[L:, n] = 1.0
        prev.view(T, N)[L:, n] = 0
    ref = det_fill(MapCMAPolicyRef(use_pm=True), seed=0).train()
    loss_r, act_r, aux_r, logits_r = ref.update_loss(obs, prev, nd, tgt, w)
    loss_r.backward()
    gref = {k: float(p.grad.norm()) for k, p in ref.named_parameters() if p.grad is not None}

    pol = make_policy(use_pm=True, train=True)
    dobs = {k: v.to(DEV) for k, v in trim_instruction_padding(dict(obs), first_rows=N).items()}
    assert dobs["instruction"].shape[1] == 80
    AuxLosses.activate()
    AuxLosses.clear()
    try:
        h0 = torch.zeros(N, 2, 512, device=DEV)
        dist, _ = pol.build_distribution(dobs, h0, prev.to(DEV), nd.to(DEV))
        logits = dist.logits.view(T, N, -1)
        ce = F.cross_entropy(logits.permute(0, 2, 1), tgt.to(DEV), reduction="none")
        wd = w.to(DEV)
        action_loss = ((wd * ce).sum(0) / wd.sum(0)).mean()
        aux = AuxLosses.reduce((wd > 0).view(-1))
        loss = action_loss + aux
        loss.backward()
    finally:
        AuxLosses.deactivate()
    live = (w > 0).view(T, N, 1).expand_as(logits_r)
    # Categorical stores normalised logits (log-probs); the oracle returns the raw head output
    err = float((torch.log_softmax(logits.detach().cpu(), -1) - torch.log_softmax(logits_r.detach(), -1))[live].abs().max())
    print(f"T64xN8: loss {float(loss):.7f} ref {float(loss_r):.7f} aux {float(aux):.7f} ref {float(aux_r):.7f} "
          f"logits max|err| {err:.2e}")
    assert err < 1e-4
    assert abs(float(loss) - float(loss_r)) < 2e-5 and abs(float(aux) - float(aux_r)) < 2e-5
    bad = []
    for k, p in pol.named_parameters():
        if not p.requires_grad or k not in gref:
            continue
        got = float(p.grad.norm())
        # conv biases in front of a train-mode BatchNorm: analytically ZERO gradient (exact 0 here); the oracle's
        # autograd holds rounding noise there that grows with the batch (2e-4 at 512 rows of 64x64 maps)
        noise = 1e-3 if (".conv.0.bias" in k and "map_encoder" in k) else 1e-7
        if not (abs(got - gref[k]) / max(1e-6, abs(gref[k])) < 1e-3 or abs(got - gref[k]) < noise):
            bad.append(f"{k}: {got:.6e} ref {gref[k]:.6e}")
    assert not bad, "\n".join(bad)


@pytest.mark.gpu
def test_update_with_deduplicated_instruction_rows_is_the_same_update():
    """Loader-side instruction de-duplication (utils.dedupe_instructions): encoding the U unique token rows once and
    letting T*N rows share them (indexed attention, per-row gradients folded in row order) must give the loss and the
    gradients of the plain per-row path - unequal trajectory lengths (collate's all-ones padding rows) included."""
    from test_gpu_policy import make_policy

    from ivln_ce_amd.aux_losses import AuxLosses
    from ivln_ce_amd.trainers import FlatAdam, update_agent
    from ivln_ce_amd.utils import dedupe_instructions, trim_instruction_padding

    T, N = 12, 5
    TN = T * N
    g = torch.Generator().manual_seed(12)
    lens = [12, 9, 12, 4, 7]
    instr = torch.zeros(N, 200)
    for n in range(N):
        L = 30 + 9 * n
        instr[n, :L] = torch.randint(2, 2504, (L,), generator=g).float()
    obs_h = {"depth_features": torch.randn(TN, 128, 4, 4, generator=g),
             "occupancy_map": (torch.rand(TN, 64, 64, generator=g) < 0.3).float(),
             "semantic_map": torch.randint(0, 13, (TN, 64, 64), generator=g).float(),
             "instruction": instr.repeat(T, 1), "progress": torch.rand(TN, 1, generator=g)}
    prev = torch.randint(0, 4, (TN, 1), generator=g)
    nd = torch.ones(T, N, dtype=torch.uint8)
    nd[0] = 0
    tgt = torch.randint(0, 4, (T, N), generator=g)
    w = torch.where(torch.rand(T, N, generator=g) < 0.4, torch.tensor(3.2), torch.tensor(1.0))
    for n, Ln in enumerate(lens):
        w[Ln:, n] = 0
        tgt[Ln:, n] = 0
        for k in obs_h:
            obs_h[k].view(T, N, *obs_h[k].shape[1:])[Ln:, n] = 1.0
    obs_h = trim_instruction_padding(obs_h, first_rows=N)
    dd = dedupe_instructions(obs_h)
    assert dd["instruction_unique"].shape[0] == N + 1 and dd["instruction_index"].shape == (TN,)  # + the padding row
    assert torch.equal(dd["instruction_unique"][dd["instruction_index"].long()], obs_h["instruction"])
    res = []
    for host in (obs_h, dd):
        pol = make_policy(use_pm=True, train=True)
        opt = FlatAdam(pol, lr=2.5e-4)
        obs = {k: v.float().to(DEV) for k, v in host.items()}
        AuxLosses.activate()
        try:
            loss = update_agent(pol, opt, obs, prev.to(DEV), nd.view(-1, 1).to(DEV), tgt.to(DEV), w.to(DEV), step_grad=False)
        finally:
            AuxLosses.deactivate()
        res.append((loss, {k: p.grad.detach().clone() for k, p in pol.named_parameters() if p.requires_grad}))
    (l0, g0), (l1, g1) = res
    assert abs(l0[0] - l1[0]) < 1e-6 and abs(l0[2] - l1[2]) < 1e-6, (l0, l1)
    for k in g0:
        a, b = g0[k], g1[k]
        tol = 1e-5 * max(1.0, float(a.abs().max()))
        assert float((a - b).abs().max()) <= tol, (k, float((a - b).abs().max()), float(a.abs().max()))


@pytest.mark.parametrize("trainer,depth_mode", [("dagger", 0), ("iterative_collection_dagger", 2)])
def test_sampled_collection_replays_as_graphs_bit_identical_to_eager(tmp_path, trainer, depth_mode, same_depth_path):
    """DAgger collection (`policy.act(deterministic=False)` + beta-mixing, dagger_trainer.py:416-427) with the action
    drawn and mixed in the head's own launch from host uniforms: mapper + policy replay as captured hipGraphs
    (IL.DAGGER.USE_HIP_GRAPH) and every stored trajectory - cached depth features, maps, previous and expert actions -
    equals the eager collection's bit for bit, across the re-captures when envs pause (beta == 1 second pass)."""
    import ivln_ce_amd  # noqa: F401
    from ivln_ce_amd import trainers  # noqa: F401
    from ivln_ce_amd.registry import baseline_registry

    same_depth_path(depth_mode)  # one depth-encoder implementation for the replayed and the eager collection
    stores = {}
    real = trainers.construct_envs
    # three episodes per env: the beta == 1 pass runs every env out of new episodes (pauses, re-captures)
    trainers.construct_envs = lambda *a, **k: real(*a, n_episodes=3, episodes_per_tour=2, **k)
    request_cleanup = lambda: setattr(trainers, "construct_envs", real)  # noqa: E731
    for mode in (True, False):
        cfg = _tiny_cfg(tmp_path / f"g{int(mode)}", trainer)
        cfg.defrost()
        cfg.NUM_ENVIRONMENTS = 3
        cfg.IL.DAGGER.update_size = 7
        cfg.IL.DAGGER.USE_HIP_GRAPH = mode
        cfg.freeze()
        torch.manual_seed(0)
        tr = baseline_registry.get_trainer(trainer)(cfg)
        envs = trainers.construct_envs(cfg, None, iterative=trainer != "dagger")
        observation_space, action_space = tr._get_spaces(cfg, envs=envs)
        envs.close()
        tr._initialize_policy(cfg, False, observation_space, action_space)
        recs = []
        for data_it, p in ((1, 0.5), (0, 1.0)):   # beta = 0.5 (sampled + mixed), then beta = 1 (unique episodes, pauses)
            cfg.defrost()
            cfg.IL.DAGGER.p = p
            cfg.freeze()
            torch.manual_seed(5 + data_it)
            n0 = len(tr.store)
            tr._update_dataset(data_it)
            assert len(tr.store) - n0 >= 5
        first = 0 if trainer == "dagger" else 0
        for i in range(first, len(tr.store)):
            recs.append(tr.store.get(i))
        stores[mode] = recs
        bufs = {k: v.clone() for k, v in tr.policy.named_buffers()}
        stores[(mode, "buffers")] = bufs
    request_cleanup()
    assert len(stores[True]) == len(stores[False])
    for (oa, pa, ea), (ob, pb, eb) in zip(stores[True], stores[False]):
        assert sorted(oa) == sorted(ob)
        for k in oa:
            assert oa[k].dtype == ob[k].dtype and np.array_equal(oa[k], ob[k]), k
        assert np.array_equal(pa, pb) and np.array_equal(ea, eb)
    # quirk Q6: BatchNorm runs on batch statistics during collection and updates its running statistics - the
    # captured steps must leave exactly the eager loop's buffers behind (the capture warm-up's updates are undone)
    for k, v in stores[(True, "buffers")].items():
        assert torch.equal(v, stores[(False, "buffers")][k]), k


def test_update_on_the_dedicated_work_stream_equals_the_update_on_the_callers_stream():
    """`ops.eager_work_stream` (trainers.update_agent): the update runs on a stream that never launches hipGraphs and is
    ordered behind / in front of the caller's stream - same loss, same gradients as with the switch off, and the
    caller's stream sees the results without any synchronisation of its own."""
    from test_gpu_policy import make_policy

    from ivln_ce_amd import ops
    from ivln_ce_amd.aux_losses import AuxLosses
    from ivln_ce_amd.trainers import FlatAdam, update_agent

    T, N = 6, 3
    TN = T * N
    g = torch.Generator().manual_seed(21)
    instr = torch.zeros(N, 200)
    instr[:, :25] = torch.randint(2, 2504, (N, 25), generator=g).float()
    obs_h = {"depth_features": torch.randn(TN, 128, 4, 4, generator=g),
             "occupancy_map": (torch.rand(TN, 64, 64, generator=g) < 0.3).float(),
             "semantic_map": torch.randint(0, 13, (TN, 64, 64), generator=g).float(),
             "instruction": instr.repeat(T, 1), "progress": torch.rand(TN, 1, generator=g)}
    prev = torch.randint(0, 4, (TN, 1), generator=g)
    nd = torch.ones(T, N, dtype=torch.uint8)
    nd[0] = 0
    tgt = torch.randint(0, 4, (T, N), generator=g)
    w = torch.ones(T, N)
    res = []
    saved = ops.EAGER_WORK_STREAM
    try:
        for on in (False, True):
            ops.EAGER_WORK_STREAM = on
            pol = make_policy(use_pm=True, train=True)
            opt = FlatAdam(pol, lr=2.5e-4)
            obs = {k: v.float().to(DEV) for k, v in obs_h.items()}
            AuxLosses.activate()
            try:
                loss = update_agent(pol, opt, obs, prev.to(DEV), nd.view(-1, 1).to(DEV), tgt.to(DEV), w.to(DEV),
                                    step_grad=False)
            finally:
                AuxLosses.deactivate()
            # read on the caller's (null) stream, no synchronize: the region's exit ordered it behind the work stream
            res.append((loss, {k: p.grad.detach().clone() for k, p in pol.named_parameters() if p.requires_grad}))
            assert torch.cuda.current_stream() == torch.cuda.default_stream()
    finally:
        ops.EAGER_WORK_STREAM = saved
    (l0, g0), (l1, g1) = res
    assert l0 == l1, (l0, l1)
    for k in g0:  # (the embedding gradient accumulates with atomics: equal to the last bits, not bit for bit)
        a, b = g0[k], g1[k]
        assert float((a - b).abs().max()) <= 1e-6 * max(1.0, float(a.abs().max())), k


# ------------------------------------------------------------------------------------------------
# the BENCHED update path (update_agent: HIP loss + HIP backward + FlatAdam) pinned end to end
# ------------------------------------------------------------------------------------------------
_ZERO_GRAD_BIAS = lambda k: k.startswith("net.map_encoder.cnn.") and k.endswith(".conv.0.bias")  # noqa: E731


@pytest.mark.gpu
@pytest.mark.parametrize("custom_lr", [False, True], ids=["one_group", "custom_lr_two_groups"])
def test_three_hip_updates_track_oracle_plus_torch_adam(custom_lr):
    """Three consecutive `update_agent` + `FlatAdam` steps (what bench.py's update leg and the trainers run) against
    the oracle's loss + autograd + `torch.optim.Adam` built exactly as base_il_trainer.py:78-94 builds it (one group, or
    the two-group MODEL.SEMANTIC_MAP_ENCODER.custom_lr form): every trainable parameter, both Adam moments of every
    element and every BatchNorm buffer after every step.

    The oracle runs in FLOAT64 here: the exact arithmetic both fp32 implementations approximate.  Measured in this
    container on this batch (policy_update.npz): the reference-equivalent torch-CPU fp32 autograd is itself up to
    5.9e-5 (4e-3 of the tensor's largest element) away from its own float64 run on net.map_encoder.cnn.2.conv.0.weight
    (BatchNorm backward cancellation), and the first GPU run of this test found the HIP gradient 5.943e-5 away from the
    fp32 oracle at that same element - i.e. ON the float64 value.  Against fp32 torch the chain could only be pinned to
    the oracle's own noise; against float64 it is pinned to the arithmetic.

    Bars, per step (T = tensor-wide maximum of the reference quantity; the map CNN's weight gradients are ill-conditioned
    in fp32 - second GPU run of this test: HIP 2.7e-5 = 2.3e-3 T away from float64 on cnn.1.conv.0.weight, where the fp32
    torch oracle is 8e-6 away, and the other way round on cnn.2 - so elementwise bars are tensor-relative):
      * Adam's moments of EVERY trainable element, read out of the flat buckets, against torch.optim.Adam's state:
        |exp_avg - ref| <= 3e-7 + 1e-3 |ref| + 5e-3 T; the same for sqrt(exp_avg_sq);
      * parameters: within 1.2 % of the group's learning rate after the first step (3e-6 at lr 2.5e-4), 15 % after the later
        ones, on every element whose reference
        gradient of that step was at
        least max(1e-5, 2e-2 T) in magnitude, and within 2 * steps * lr everywhere.  Adam's step is
        lr * m / (sqrt(v) + eps), i.e. lr * sign(g) in the first steps: where |g| is of the order of the gradient's fp32
        noise the SIGN is noise and an element may legitimately move the other way by lr.
    The conv biases in front of a train-mode BatchNorm are such elements by construction: their gradient is
    analytically zero, autograd leaves rounding noise there (so the reference's Adam random-walks them by +-lr per
    step, with no effect on any output), the HIP backward writes an exact zero and they do not move."""
    from det_init import det_fill
    from test_gpu_policy import make_policy

    from ivln_ce_amd.aux_losses import AuxLosses
    from ivln_ce_amd.trainers import FlatAdam, update_agent
    from oracle.policy_ref import MapCMAPolicyRef

    torch.set_num_threads(8)
    lr, sem_lr, steps = 2.5e-4, 1e-3, 3
    g = np.load(os.path.join(G, "policy_update.npz"))
    obs, prev, nd, tgt, w = _batch(g)
    obs_h = {k: (v.cpu().double() if v.is_floating_point() else v.cpu()) for k, v in obs.items()}
    ref = det_fill(MapCMAPolicyRef(use_pm=True), seed=0).train().double()  # (filled in fp32: the HIP policy's exact weights)
    if custom_lr:
        sem = [p for k, p in ref.named_parameters() if k.startswith("net.map_encoder")]
        reg = [p for k, p in ref.named_parameters() if not k.startswith("net.map_encoder")]
        opt_r = torch.optim.Adam([{"params": sem}, {"params": reg}], lr=lr)
        opt_r.param_groups[0]["lr"] = sem_lr
    else:
        opt_r = torch.optim.Adam(ref.parameters(), lr=lr)
    pol = make_policy(use_pm=True, train=True)
    opt = FlatAdam(pol, lr=lr, sem_lr=sem_lr if custom_lr else None)
    gmin, gmax = {}, {}
    log = []
    AuxLosses.activate()
    try:
        for s in range(1, steps + 1):
            if s > 1:
                # Re-align the parameters (NOT the optimizer states) with the oracle's before every further step: the
                # noise-level elements the two sides moved in opposite directions (2 lr = 1 % of a typical weight) would
                # otherwise perturb the next forward at the 1e-2 level and the comparison would measure that chaos, not
                # Adam.  Each side keeps its own moments and step count, so steps 2 and 3 still exercise the carried
                # state, the bias correction and the two-group learning rates.
                ref_now = dict(ref.named_parameters())
                with torch.no_grad():
                    for k, p in pol.named_parameters():
                        if p.requires_grad:
                            p.copy_(ref_now[k].detach().float().to(DEV))
            opt_r.zero_grad()
            torch.set_default_dtype(torch.float64)  # (tensors the oracle creates itself: initial state, one-hot maps)
            try:
                loss_r, act_r, aux_r, _ = ref.update_loss(obs_h, prev.cpu(), nd.cpu(), tgt.cpu(), w.cpu().double())
                loss_r.backward()
            finally:
                torch.set_default_dtype(torch.float32)
            for k, p in ref.named_parameters():
                if p.grad is not None:
                    a = p.grad.detach().abs()
                    gmin[k] = a  # (this step's |gradient|: the parameters are re-aligned before every step)
                    gmax[k] = float(a.max())
            opt_r.step()
            loss, act, aux = update_agent(pol, opt, obs, prev, nd, tgt, w, hidden_size=512)
            assert abs(loss - float(loss_r)) < 2e-5 and abs(aux - float(aux_r)) < 2e-5, (s, loss, float(loss_r))
            # Adam's moments, every trainable element
            m_err = v_err = 0.0
            st = opt_r.state
            ref_p = dict(ref.named_parameters())
            diag = []
            for k, p, o in zip(opt.names, opt.params, opt.offsets):
                if _ZERO_GRAD_BIAS(k) or ref_p[k] not in st:
                    continue
                n = p.numel()
                m_h, v_h = opt.exp_avg[o:o + n].cpu(), opt.exp_avg_sq[o:o + n].cpu()
                m_r, v_r = st[ref_p[k]]["exp_avg"].reshape(-1), st[ref_p[k]]["exp_avg_sq"].reshape(-1)
                m_h, v_h, v_r = m_h.double(), v_h.double().sqrt(), v_r.sqrt()
                em = float(((m_h - m_r).abs() - 1e-3 * m_r.abs()).max()) - 5e-3 * float(m_r.abs().max())
                ev = float(((v_h - v_r).abs() - 1e-3 * v_r.abs()).max()) - 5e-3 * float(v_r.max())
                m_err, v_err = max(m_err, em), max(v_err, ev)
                diag.append((float((m_h - m_r).abs().max()), float(m_r.abs().max()), em, ev, k))
            diag.sort(reverse=True)
            for e_abs, m_max, em, ev, k in diag[:8]:
                log.append(f"  step {s} exp_avg {k}: max|err| {e_abs:.3e} (max|ref| {m_max:.3e}); beyond the bar: {em:.2e} / sqrt(sq) {ev:.2e}")
            os.makedirs("gpurun_out", exist_ok=True)
            open(f"gpurun_out/update_3steps_{'custom_lr' if custom_lr else 'one_group'}.log", "w").write("\n".join(log) + "\n")
            bad_m = [d for d in diag if d[2] > 3e-7 or d[3] > 3e-7]
            assert not bad_m, f"step {s}: Adam moments off: " + "; ".join(f"{d[4]} exp_avg {d[2]:.2e} sq {d[3]:.2e}" for d in bad_m[:6])
            tot = low = 0
            worst = (0.0, "")
            for k, p in pol.named_parameters():
                if not p.requires_grad:
                    continue
                d = (p.detach().cpu().double() - ref_p[k].detach()).abs()
                firm = gmin[k] >= max(1e-5, 2e-2 * gmax[k])
                tot += d.numel()
                low += int((~firm).sum())
                if _ZERO_GRAD_BIAS(k):
                    continue  # (analytically zero gradient, see the docstring)
                if firm.any():
                    e = float(d[firm].max())
                    if e > worst[0]:
                        worst = (e, k)
                    # (from step 2 on exp_avg = 0.9 m + 0.1 g can cancel - opposite gradient signs in consecutive steps - so the
                    #  same absolute moment error is a larger share of the update: up to 20 % of lr instead of 1.2 %.  Measured
                    #  worst element: 13 % with the map CNN's convs on the fp32 MFMA kernels, 16 % with them on the split-bf16
                    #  kernel - an exp_avg of 1.4e-5 that is 3.7e-6 off, inside the moment bars above and below the 5.3e-6 the
                    #  fp32 kernels leave elsewhere in the same tensor)
                    lr_k = sem_lr if (custom_lr and k.startswith("net.map_encoder")) else lr
                    if e > (0.012 if s == 1 else 0.20) * lr_k:
                        i = int(torch.where(firm, d, torch.zeros_like(d)).reshape(-1).argmax())
                        o = opt.offsets[opt.names.index(k)]
                        raise AssertionError(
                            f"step {s}: {k} differs by {e:.3e} on elements above the gradient bar: element {i}, min |g_ref| "
                            f"{float(gmin[k].reshape(-1)[i]):.3e}, exp_avg hip {float(opt.exp_avg[o + i]):.4e} ref "
                            f"{float(st[ref_p[k]]['exp_avg'].reshape(-1)[i]):.4e}, exp_avg_sq hip {float(opt.exp_avg_sq[o + i]):.4e} ref "
                            f"{float(st[ref_p[k]]['exp_avg_sq'].reshape(-1)[i]):.4e}, p hip {float(p.detach().reshape(-1)[i]):.6e} ref "
                            f"{float(ref_p[k].detach().reshape(-1)[i]):.6e}")
                assert float(d.max()) <= 2 * s * max(lr, sem_lr if custom_lr else lr) * 1.001 + 1e-7, f"step {s}: {k} moved too far"
            for k, b in pol.named_buffers():
                rb = dict(ref.named_buffers())[k]
                if b.dtype.is_floating_point:
                    assert torch.allclose(b.cpu().double(), rb, atol=2e-6, rtol=1e-5), f"step {s}: buffer {k}"
                else:
                    assert int(b) == int(rb), k
            log.append(f"step {s}: loss {loss:.7f} ref {float(loss_r):.7f}; worst firm element {worst[0]:.2e} ({worst[1]}); "
                       f"{low} of {tot} elements below the gradient bar; moments beyond the bar: exp_avg {m_err:.2e}, sqrt(exp_avg_sq) {v_err:.2e}")
            assert low < 0.8 * tot, log[-1]  # (the elementwise parameter check must cover a real share; the moments above cover every element)
    finally:
        AuxLosses.deactivate()
    os.makedirs("gpurun_out", exist_ok=True)
    open(f"gpurun_out/update_3steps_{'custom_lr' if custom_lr else 'one_group'}.log", "w").write("\n".join(log) + "\n")
    print("\n".join(log))


_bench_case_cache = {}


def _bench_shape_case():
    """T = 64 x N = 8 trajectory batch of the bench shape + the oracle's loss and FULL gradients (computed once)."""
    if _bench_case_cache:
        return _bench_case_cache
    from det_init import det_fill

    from oracle.policy_ref import MapCMAPolicyRef

    torch.set_num_threads(8)
    T, N = 64, 8
    TN = T * N
    g = torch.Generator().manual_seed(65)
    lens = [64, 64, 51, 40, 64, 33, 64, 57]
    instr = torch.zeros(N, 200)
    for n in range(N):
        L = 80 - 7 * n
        instr[n, :L] = torch.randint(2, 2504, (L,), generator=g).float()
    obs = {"depth_features": torch.randn(TN, 128, 4, 4, generator=g),
           "occupancy_map": (torch.rand(TN, 64, 64, generator=g) < 0.3).float(),
           "semantic_map": torch.randint(0, 13, (TN, 64, 64), generator=g).float(),
           "instruction": instr.repeat(T, 1), "progress": torch.rand(TN, 1, generator=g)}
    prev = torch.randint(0, 4, (TN, 1), generator=g)
    nd = torch.ones(T, N, dtype=torch.uint8)
    nd[0] = 0
    nd = nd.view(-1, 1)
    tgt = torch.randint(0, 4, (T, N), generator=g)
    infl = torch.ones(T, N, dtype=torch.bool)
    infl[1:] = tgt[1:] != tgt[:-1]
    w = torch.where(infl, torch.tensor(3.2), torch.tensor(1.0))
    for n, L in enumerate(lens):
        w[L:, n] = 0
        tgt[L:, n] = 0
        for k in obs:
            obs[k].view(T, N, *obs[k].shape[1:])[L:, n] = 1.0
        prev.view(T, N)[L:, n] = 0
    ref = det_fill(MapCMAPolicyRef(use_pm=True), seed=0).train()
    loss_r, act_r, aux_r, _ = ref.update_loss(obs, prev, nd, tgt, w)
    loss_r.backward()
    _bench_case_cache.update(T=T, N=N, obs=obs, prev=prev, nd=nd, tgt=tgt, w=w, loss=float(loss_r), aux=float(aux_r),
                             grads={k: p.grad.detach().clone() for k, p in ref.named_parameters() if p.grad is not None})
    return _bench_case_cache


@pytest.mark.gpu
def test_benched_update_path_full_gradients_at_T64_N8():
    """`update_agent` itself (HIP cross-entropy + inflection weights, HIP backward, persistent sequence GRUs, de-duplicated
    instruction rows - the path bench.py's update leg times) at the bench shape, FULL gradient tensors against the
    oracle's autograd, not norms: the map CNN's conv weights, both GRUs, the instruction bi-LSTM and every other trainable
    tensor.  Per tensor: max |got - ref| <= 1e-3 * max |ref| + 2e-7, and the direction cosine >= 1 - 1e-6 (a permuted
    or mis-indexed gradient fails both)."""
    from test_gpu_policy import make_policy

    from ivln_ce_amd.aux_losses import AuxLosses
    from ivln_ce_amd.trainers import FlatAdam, update_agent
    from ivln_ce_amd.utils import dedupe_instructions, trim_instruction_padding

    c = _bench_shape_case()
    N = c["N"]
    pol = make_policy(use_pm=True, train=True)
    opt = FlatAdam(pol, lr=2.5e-4)
    dobs = dedupe_instructions(trim_instruction_padding(dict(c["obs"]), first_rows=N))
    dobs = {k: v.to(DEV) for k, v in dobs.items()}
    AuxLosses.activate()
    try:
        loss, act, aux = update_agent(pol, opt, dobs, c["prev"].to(DEV), c["nd"].to(DEV), c["tgt"].to(DEV), c["w"].to(DEV),
                                      hidden_size=512, step_grad=False)
    finally:
        AuxLosses.deactivate()
    assert abs(loss - c["loss"]) < 2e-5 and abs(aux - c["aux"]) < 2e-5, (loss, c["loss"], aux, c["aux"])
    must = ["net.map_encoder.cnn.0.conv.0.weight", "net.map_encoder.cnn.1.conv.0.weight", "net.map_encoder.cnn.2.conv.0.weight",
            "net.map_encoder.cnn.3.conv.0.weight", "net.state_encoder.rnn.weight_hh_l0", "net.second_state_encoder.rnn.weight_hh_l0",
            "net.instruction_encoder.encoder_rnn.weight_ih_l0", "net.instruction_encoder.encoder_rnn.weight_hh_l0",
            "net.instruction_encoder.encoder_rnn.weight_ih_l0_reverse", "net.instruction_encoder.encoder_rnn.weight_hh_l0_reverse"]
    params = dict(pol.named_parameters())
    assert all(k in params and params[k].requires_grad for k in must)
    log, bad = [], []
    for k, p in params.items():
        if not p.requires_grad or k not in c["grads"]:
            continue
        got, ref = p.grad.detach().cpu().double().reshape(-1), c["grads"][k].double().reshape(-1)
        if _ZERO_GRAD_BIAS(k):
            assert float(got.abs().max()) == 0.0, k  # analytically zero: exact 0 here, rounding noise in the oracle
            continue
        scale = float(ref.abs().max())
        err = float((got - ref).abs().max())
        cos = float(torch.dot(got, ref) / (got.norm() * ref.norm()).clamp_min(1e-300)) if scale > 0 else 1.0
        log.append(f"{k}: max|ref| {scale:.3e} max|err| {err:.3e} cos-1 {cos - 1:.1e}")
        if not (err <= 1e-3 * scale + 2e-7 and (cos >= 1 - 1e-6 or scale < 1e-6)):
            bad.append(log[-1])
    os.makedirs("gpurun_out", exist_ok=True)
    open("gpurun_out/update_T64N8_full_grads.log", "w").write("\n".join(log) + "\n")
    assert not bad, "\n".join(bad)
    assert all(any(line.startswith(k + ":") for line in log) for k in must)


@pytest.mark.gpu
def test_two_rank_update_equals_single_process_accumulation(tmp_path):
    """Data-parallel equivalence (tools/dp_equiv.py): two gloo ranks on this GPU, 4 trajectories each, one
    all-reduced update == one process accumulating both shards' gradients and stepping with 1/2 folded into Adam."""
    env = dict(os.environ, IVLN_DIST_BACKEND="gloo", IVLN_ONE_DEVICE="1")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    from conftest import run_torchrun

    r = run_torchrun(2, os.path.join(root, "tools", "dp_equiv.py"), env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "dist smoke ok: dp_equiv world 2" in r.stdout
    print(r.stdout[-600:])
