"""bench.py - throughput of the MapCMA hot path on MI355X (driver contract: see DESIGN.md section 6).

  python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run)

A "step" = one pass of the hot path over one batch of synthetic observations already resident in
HBM: egocentric mapper (gt semantics) + MapCMAPolicy.act for `--envs` (default 4) parallel envs per
GPU = BASELINE.json configs[1].  `value` = env-steps/s over all ranks (weak scaling: envs per GPU
fixed).  The JSON line also carries
  roofline     - fp32-MFMA implicit-GEMM kernel family (all conv / linear FLOPs of the step):
                 algorithmic FLOPs per step / summed kernel time per step (HIP events on the launch
                 stream) against the 157.3 TFLOP/s fp32-matrix peak
  cpu_baseline - the CPU oracle (torch-CPU policy port + C mapper) timed on this box's host cores
                 on a bounded sample of the same workload (rank 0, N=1 only)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

METRIC = "env-steps/sec (batched MapCMA fwd+bwd) at 1/2/4/8 MI355X; t-nDTW parity"
PEAK_F32_MFMA_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md, v_mfma_f32_32x32x2_f32


def make_policy(device, seed=0):
    from ivln_ce_amd.config import get_config
    from ivln_ce_amd.policy import MapCMAPolicy
    from ivln_ce_amd.spaces import Box, Dict, Discrete

    cfg = get_config(opts=[
        "MODEL.policy_name", "MapCMAPolicy", "MODEL.INSTRUCTION_ENCODER.use_pretrained_embeddings", False,
        "MODEL.DEPTH_ENCODER.ddppo_checkpoint", "NONE",
    ])
    space = Dict({
        "depth": Box(0.0, 1.0, (256, 256, 1), np.float32), "occupancy_map": Box(0, 255, (64, 64), np.uint8),
        "semantic_map": Box(0, 255, (64, 64), np.uint8), "instruction": Box(0, 2504, (200,), np.int64),
    })
    torch.manual_seed(seed)
    pol = MapCMAPolicy.from_config(cfg, space, Discrete(4))
    return cfg, pol.to(device).eval()


def gen_observations(B, n_steps, seed, with_rgb=False):
    from ivln_ce_amd.synthetic import SyntheticRollout

    roll = SyntheticRollout(B=B, seed=seed, with_rgb=with_rgb)
    return [roll.step() for _ in range(n_steps)]


class GemmTimer:
    """Wraps ops.gemm with event pairs on the launch stream; also accumulates algorithmic FLOPs."""

    def __init__(self):
        self.events = []
        self.flops = 0

    def __enter__(self):
        from ivln_ce_amd import ops

        self.ops = ops
        self.orig = ops.gemm

        def timed(desc):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            self.orig(desc)
            b.record()
            self.events.append((a, b))
            self.flops += 2 * desc.M * desc.N * desc.K

        ops.gemm = timed
        return self

    def __exit__(self, *a):
        self.ops.gemm = self.orig

    def total_ms(self):
        """Sum of the event-pair times.  A pair brackets one launch on the launch stream, so it carries the
        launch's dispatch latency as well (about 2.5 us more per launch than rocprofv3's kernel durations,
        profiles/r01_rollout_eager_kernel_stats.csv): the reported TFLOP/s is the conservative figure."""
        torch.cuda.synchronize()
        return sum(a.elapsed_time(b) for a, b in self.events)

def rollout_step(mapper_tr, policy, obs, state):
    """mapper (obs-transform plugin) + policy.act: the per-step body of the reference eval loop
    (base_il_trainer.py:688-703, 841)."""
    batch = dict(obs)
    batch = mapper_tr(batch)
    with torch.no_grad():
        actions, state["rnn"] = policy.act(batch, state["rnn"], state["prev"], batch["not_done_masks"],
                                           deterministic=True)
    state["prev"] = actions
    return actions


def usable_cores():
    """Host cores this process may actually use: affinity mask capped by the cgroup CPU quota
    (the GPU box reports 256 logical CPUs but grants a 16-CPU quota; oversubscribing OpenMP there
    makes the CPU leg ~1000x slower)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:  # noqa: BLE001
        pass
    return max(1, n)


def cpu_baseline(obs_cpu, B, budget_s=12.0):
    """Torch-CPU policy port + C mapper oracle on the host cores (kind = "port")."""
    from oracle.mapper_ref import MapperRef
    from oracle.policy_ref import MapCMAPolicyRef

    torch.manual_seed(0)
    ncores = usable_cores()
    torch.set_num_threads(ncores)
    pol = MapCMAPolicyRef().eval()
    mapper = MapperRef(256, 256)
    rnn = torch.zeros(B, 2, 512)
    prev = torch.zeros(B, 1, dtype=torch.long)

    def step(o):
        nonlocal rnn, prev
        occ, sem = mapper.step(o["depth"].numpy(), o["semantic12"].numpy(), o["world_robot_pose"].numpy(),
                               o["world_robot_orientation"].numpy(), o["not_done_masks"].numpy())
        ob = {"depth": o["depth"], "instruction": o["instruction"], "occupancy_map": torch.from_numpy(occ),
              "semantic_map": torch.from_numpy(sem)}
        with torch.no_grad():
            a, rnn, _ = pol.act(ob, rnn, prev, o["not_done_masks"])
        prev = a

    for o in obs_cpu[:3]:
        step(o)
    n, t0 = 0, time.perf_counter()
    i = 3
    while True:
        step(obs_cpu[i % len(obs_cpu)])
        i += 1
        n += 1
        el = time.perf_counter() - t0
        if (n >= 20 and el > budget_s) or n >= 400 or el > 4 * budget_s:
            break
    return {
        "value": round(B * n / el, 2), "unit": "env-steps/s", "cores": ncores, "kind": "port",
        "sample": f"{n} steps of {B} envs (256x256 depth, gt semantics, 80-token instruction): C mapper oracle + "
                  f"torch-CPU MapCMA port, {ncores} threads, after 3 warm-up steps",
    }


def pmc_traffic(B):
    """HBM bytes per MFMA-family launch from the committed rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE
    in separate runs, gfx950 FETCH correction applied; profiles/r01_rollout_pmc_traffic.json).  A counter
    pass cannot run inside the timed bench, so the figure is the committed one and only for its workload."""
    path = os.path.join(ROOT, "profiles", "r01_rollout_pmc_traffic.json")
    if B != 4 or not os.path.exists(path):
        return None
    try:
        return json.load(open(path))["mfma_family"]["hbm_bytes_per_launch_corrected"]
    except Exception:  # noqa: BLE001
        return None


def log(*a):
    print("[bench]", *a, file=sys.stderr, flush=True)


def bench_update(policy, dev, world, barrier, T=64, N=8, iters=5, warm=2):
    """DAgger update step (base_il_trainer.py:173-219): forward over T*N rows with BPTT, weighted CE +
    progress-monitor loss, hand-written HIP backward, one flat-bucket RCCL all-reduce (world > 1), Adam.
    Same barrier / max-over-ranks clock as the rollout leg; rows/s is the whole-job aggregate."""
    from ivln_ce_amd.trainers import FlatAdam, update_agent

    policy.train()
    opt = FlatAdam(policy, lr=2.5e-4)
    g = torch.Generator().manual_seed(7)
    TN = T * N
    instr = torch.zeros(N, 200)
    instr[:, :80] = torch.randint(2, 2504, (N, 80), generator=g).float()
    from ivln_ce_amd.utils import trim_instruction_padding

    # the trainer's loader drops the all-padding tail of the token batch on the host (trainers.PrefetchLoader):
    # like the reference's packed LSTM, the update only ever sees the batch's longest instruction (80 of 200)
    host = trim_instruction_padding({"instruction": instr.repeat(T, 1)})
    obs = {"depth_features": torch.randn(TN, 128, 4, 4, generator=g).to(dev),
           "occupancy_map": (torch.rand(TN, 64, 64, generator=g) < 0.3).float().to(dev),
           "semantic_map": torch.randint(0, 13, (TN, 64, 64), generator=g).float().to(dev),
           "instruction": host["instruction"].to(dev)}
    prev = torch.randint(0, 4, (TN, 1), generator=g).to(dev)
    nd = torch.ones(T, N, dtype=torch.uint8)
    nd[0] = 0
    nd = nd.view(-1, 1).to(dev)
    tgt = torch.randint(0, 4, (T, N), generator=g).to(dev)
    w = torch.ones(T, N).to(dev)
    for _ in range(warm):
        update_agent(policy, opt, obs, prev, nd, tgt, w, world=world)
    barrier()
    t0 = time.perf_counter()
    for _ in range(iters):
        update_agent(policy, opt, obs, prev, nd, tgt, w, world=world)
    barrier()
    el = time.perf_counter() - t0
    # MFMA kernel family of one update (instrumented pass, outside the timed region).  The event pairs sum
    # per-launch elapsed times, so the pass runs everything on one stream: with the instruction branch on its
    # side stream (the timed configuration) concurrent launches would be counted twice over the same wall time.
    from ivln_ce_amd import train as _train

    overlap, _train.OVERLAP_INSTRUCTION = _train.OVERLAP_INSTRUCTION, False
    try:
        with GemmTimer() as gt:
            update_agent(policy, opt, obs, prev, nd, tgt, w, world=world)
            ms = gt.total_ms()
    finally:
        _train.OVERLAP_INSTRUCTION = overlap
    ach = (gt.flops / (ms * 1e-3)) / 1e12 if ms > 0 else 0.0
    roof = {"bound": "mfma", "achieved": round(ach, 2), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
            "frac": round(ach / PEAK_F32_MFMA_TFLOPS, 4), "flops_per_update": int(gt.flops),
            "launches_per_update": len(gt.events), "kernel_ms_per_update": round(ms, 3),
            "kernel": "fp32 MFMA family: k_conv_direct / k_wgrad_direct / k_gemm_vec / k_gemm"}
    policy.eval()
    return el, {"rows_per_step_per_gpu": TN, "T": T, "N": N, "iters": iters, "roofline": roof}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--envs", type=int, default=4, help="parallel envs per GPU (configs[1]: 4)")
    ap.add_argument("--pred-semantics", action="store_true",
                    help="BASELINE configs[2]: RedNet-predicted semantics feed the mapper (not the headline workload; "
                         "no CPU baseline / roofline legs)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pred-leg", action="store_true", help="skip the pred-semantics leg (extra JSON object)")
    ap.add_argument("--no-update", action="store_true", help="skip the DAgger update-step leg (extra JSON object)")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay")
    ap.add_argument("--single-stream", action="store_true",
                    help="one graph on one stream instead of the default three graphs on two streams (depth ResNet || "
                         "mapper + map CNN + instruction encoder, then the head)")
    ap.add_argument("--streams", action="store_true",
                    help="fork the three encoder branches onto side streams inside the graph (measured SLOWER on "
                         "ROCm 7.2: cross-queue dependencies cost more than the overlap wins)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if os.environ.get("IVLN_BENCH_ONE_DEVICE"):
            # control-flow smoke test of the multi-rank path on a 1-GPU box: every rank on cuda:0, gloo
            # instead of RCCL (which refuses two ranks on one device).  Not a measurement.
            local_rank = 0
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    import __graft_entry__ as ge

    if rank == 0:
        ge.build()
    if world > 1:
        dist.barrier()

    from ivln_ce_amd.obs_transforms import GTSemanticsIterativeMapper, PredictedSemanticsIterativeMapper

    B, K, W = args.envs, args.steps, args.warmup
    cfg, policy = make_policy(dev)
    pred = args.pred_semantics
    mapper_tr = (PredictedSemanticsIterativeMapper if pred else GTSemanticsIterativeMapper).from_config(cfg)
    n_pool = min(W + K, 240)
    obs_cpu = gen_observations(B, n_pool, seed=1234 + rank, with_rgb=pred)
    obs_dev = [{k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in o.items()} for o in obs_cpu]
    state = {"rnn": torch.zeros(B, 2, 512, device=dev), "prev": torch.zeros(B, 1, dtype=torch.long, device=dev)}

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    use_graph = not args.no_graph
    if use_graph:
        from ivln_ce_amd.graphed import GraphedRollout

        log(f"rank {rank}: capturing the step graph")
        mode = args.streams if (args.streams or args.single_stream) else "split"
        runner, launch_note = None, None
        for attempt in ([mode, False] if mode else [False]):
            try:
                runner = GraphedRollout(policy, [mapper_tr], obs_dev[0], deterministic=True, streams=attempt)
                launch_note = ("3 forked streams" if attempt is True else
                               "3 graphs on 2 streams" if attempt == "split" else "1 stream")
                break
            except Exception as e:  # noqa: BLE001 - a capture problem must not cost the measurement
                log(f"rank {rank}: graph capture ({attempt!r}) failed: {type(e).__name__}: {e}")
                torch.cuda.synchronize()
        if runner is None:
            use_graph = False

    if use_graph:
        def do_step(i):
            runner.step(obs_dev[i % n_pool])
    else:
        def do_step(i):
            rollout_step(mapper_tr, policy, obs_dev[i % n_pool], state)

    log(f"rank {rank}: inputs resident, warm-up {W} steps")
    for i in range(W):
        do_step(i)
    barrier()
    log(f"rank {rank}: timing {K} steps")
    t0 = time.perf_counter()
    for i in range(K):
        do_step(W + i)
    barrier()
    el = time.perf_counter() - t0
    mapper_tr.mapping_module.check_status()
    if world > 1:
        t = torch.tensor([el], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())

    log(f"rank {rank}: timed region {el:.3f}s")
    # ---- DAgger update step (fwd + bwd + all-reduce + Adam): reported beside the headline ----
    update = None
    if not args.no_update and not pred:
        if use_graph:
            del runner  # graphs hold the activation pools
        log(f"rank {rank}: update-step leg")
        uel, uinfo = bench_update(policy, dev, world, barrier)
        if world > 1:
            t = torch.tensor([uel], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            uel = float(t.item())
        update = {"value": round(world * uinfo["rows_per_step_per_gpu"] * uinfo["iters"] / uel, 1), "unit": "rows/s",
                  "ms_per_update": round(1e3 * uel / uinfo["iters"], 3), "rows_per_update_per_gpu": uinfo["rows_per_step_per_gpu"],
                  "what": f"DAgger update T={uinfo['T']} x N={uinfo['N']} per GPU from cached depth features: MapCMA forward "
                          "with BPTT, inflection-weighted CE + progress monitor, HIP backward, "
                          + ("one flat RCCL all-reduce, " if world > 1 else "") + "Adam",
                  "roofline": uinfo["roofline"]}
    # ---- roofline of the MFMA implicit-GEMM family: instrumented pass (not part of `value`) ----
    roofline = None
    if rank == 0 and not pred:
        n_inst = min(20, K)
        with GemmTimer() as gt:
            for i in range(n_inst):
                # eager launches are host-bound (~8 us each with the event pair): park the GPU on a spin
                # kernel while the host enqueues the step, so the event pairs time back-to-back kernels
                # (what rocprofv3's per-kernel durations show) and not the host's launch gaps
                torch.cuda._sleep(12_000_000)
                rollout_step(mapper_tr, policy, obs_dev[i % n_pool], state)
            ms = gt.total_ms()
        flops_per_step = gt.flops / n_inst
        launches = len(gt.events) / n_inst
        ach = (gt.flops / (ms * 1e-3)) / 1e12 if ms > 0 else 0.0
        roofline = {
            "bound": "mfma", "achieved": round(ach, 3), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
            "frac": round(ach / PEAK_F32_MFMA_TFLOPS, 5), "traffic": pmc_traffic(B),
            "kernel": "fp32 MFMA family (k_gemm / k_gemm_vec / k_conv_direct): all conv/linear launches of one step",
            "flops_per_step": int(flops_per_step), "launches_per_step": round(launches, 1),
            "kernel_ms_per_step": round(ms / n_inst, 4),
        }

    # ---- configs[2]: RedNet-predicted semantics feeding the mapper (extra object, not the headline) ----
    pred_leg = None
    if not pred and not args.no_pred_leg:
        from ivln_ce_amd.graphed import GraphedRollout

        log(f"rank {rank}: pred-semantics leg")
        ptr = PredictedSemanticsIterativeMapper.from_config(cfg)
        pobs = [{k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in o.items()}
                for o in gen_observations(B, 24, seed=4321 + rank, with_rgb=True)]
        try:
            prun = GraphedRollout(policy, [ptr], pobs[0], deterministic=True, streams="split")
        except Exception as e:  # noqa: BLE001
            log(f"rank {rank}: split capture failed for the pred-semantics leg ({type(e).__name__}: {e}); one stream")
            torch.cuda.synchronize()
            prun = GraphedRollout(policy, [ptr], pobs[0], deterministic=True, streams=False)
        for i in range(6):
            prun.step(pobs[i % 24])
        barrier()
        pk = 40
        t0 = time.perf_counter()
        for i in range(pk):
            prun.step(pobs[(6 + i) % 24])
        barrier()
        pel = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([pel], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            pel = float(t.item())
        pred_leg = {"value": round(world * B * pk / pel, 1), "unit": "env-steps/s", "ms_per_step": round(1e3 * pel / pk, 3),
                    "steps": pk, "what": f"BASELINE configs[2]: RedNet(rgb 224x224 + depth) -> labels -> mapper -> "
                                         f"MapCMAPolicy.act, {B} envs per GPU, graph replay"}
        del prun

    out = {
        "metric": METRIC, "value": round(world * B * K / el, 2), "unit": "env-steps/s", "n_gpus": world,
        "steps": K, "warmup": W, "ms_per_step": round(1e3 * el / K, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {
            "workload": ("BASELINE configs[2]: MapCMA pred-semantics eval step = RedNet(rgb 224x224, depth) + egocentric "
                         "mapper + MapCMAPolicy.act, " if pred else
                         "BASELINE configs[1]: MapCMA gt-semantics eval step = egocentric mapper + MapCMAPolicy.act, ")
                        + f"{B} parallel envs per GPU, 256x256 depth + semantic12, 80-token instruction, random-init "
                        "weights of the reference architecture",
            "envs_per_gpu": B, "parallelism": f"dp{world} (envs sharded, no data-path collective)",
            "launch": ("hipGraph replay, " + launch_note) if use_graph else "eager",
        },
        "roofline": roofline,
    }
    if update is not None:
        out["update_step"] = update
    if pred_leg is not None:
        out["pred_semantics_step"] = pred_leg
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline and not pred:
            log("cpu baseline ...")
            out["cpu_baseline"] = cpu_baseline(obs_cpu, B)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _watchdog(seconds, code, what):
    """A wedged GPU call cannot be interrupted from Python: end the process instead of sitting until the caller's
    limit (which would take the GPU box with it)."""
    import threading

    def _fire():
        sys.stderr.write(f"[bench] watchdog: {what} after {seconds}s, exiting\n")
        sys.stderr.flush()
        os._exit(code)

    t = threading.Timer(seconds, _fire)
    t.daemon = True
    t.start()
    return t


if __name__ == "__main__":
    wd = _watchdog(int(os.environ.get("IVLN_BENCH_LIMIT_S", "1500")), 3, "run not finished")
    main()
    wd.cancel()
    sys.stdout.flush()
    sys.stderr.flush()
    profiled = any("rocprof" in os.environ.get(k, "").lower() for k in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "HSA_TOOLS_LIB"))
    if not profiled:
        os._exit(0)  # the JSON line is out: skip interpreter / runtime teardown (graphs, streams) altogether
    import signal

    signal.signal(signal.SIGALRM, signal.SIG_DFL)
    signal.alarm(120)  # under a profiler its finalizers must run: give teardown two minutes, not forever
