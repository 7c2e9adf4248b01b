"""bench.py - throughput of the MapCMA hot path on MI355X (driver contract: see DESIGN.md section 6).

  python bench.py --gpus N --steps K --warmup W

With N > 1 and no WORLD_SIZE in the environment the parent spawns N ranks itself (one process per GPU through
`python -m torch.distributed.run`, started BEFORE the parent touches the GPU) and relays rank 0's JSON line; when
the driver launches it under torch.distributed.run the ranks are used as they come.

A "step" = one pass of the hot path over one batch of synthetic observations already resident in HBM.
Headline (`value`, `roofline`, `cpu_baseline`): BASELINE.json configs[2], the largest single-GPU configuration -
RedNet-predicted semantics (rgb 224x224 + depth 256x256) -> egocentric mapper -> MapCMAPolicy.act for `--pred-envs`
(default 8) parallel envs per GPU, env-steps/s over all ranks (weak scaling: envs per GPU fixed).  Every leg is timed
as R = 5 repetitions of W warm-up + EXACTLY K steps, each bracketed by barrier + synchronize with the MAX over ranks
taken per repetition; `value` / `ms_per_step` are the MEDIAN repetition and `repetitions` carries min / max.
The JSON line also carries
  roofline             MFMA family of the headline step: achieved = algorithmic FLOPs of one step / the step's WALL time;
                       peak = the rate of the step's FLOP mix with the fp32 MFMA pipe (157.3 TF) and the bf16 MFMA pipe
                       (2.5 PF, six executed FLOPs per algorithmic FLOP of a split-bf16 conv) at their dense peaks;
                       frac = achieved / peak = bound time / wall time (not clamped; > 1 is flagged).  Secondary: `fp32_peak_basis`
                       (rounds 1-4's figure), `kernel_time` (instrumented single-stream pass), `mfma_busy` (committed PMC)
  cpu_baseline         the CPU oracle (torch-CPU RedNet port + C mapper + torch-CPU policy port) on this box's host cores
  gt_semantics_step    BASELINE configs[1] (gt semantics, 4 envs): its own value / roofline / mapper_roofline / cpu_baseline
  update_step          DAgger update T=64 x N=8 per GPU (fwd + bwd + all-reduce + Adam): the same roofline object on the
                       update's WALL time with PMC traffic, `allreduce` (bytes / ms / bus GB/s of the one collective,
                       populated when world > 1), the oracle's update (torch-CPU loss + autograd + Adam) as cpu_baseline
  dagger_collect_step  sampled DAgger collection step at 8 envs (replayed), with the eager figure
  dagger_iteration     64 replayed collection steps alternating with 4 eager updates in ONE process, against the sum of
                       the two legs timed on their own (the graph-replay / eager-update interaction)
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

METRIC = "env-steps/sec (batched MapCMA fwd+bwd) at 1/2/4/8 MI355X; t-nDTW parity"
PEAK_F32_MFMA_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md, v_mfma_f32_32x32x2_f32
# the same guide's dense bf16 MFMA peak; a conv on the split-bf16 kernel (csrc/conv_bf3.hip) issues SIX bf16 MFMA FLOPs per
# algorithmic fp32 FLOP, so its share of a step is bounded by this / 6 = 416.7 "fp32-equivalent" TFLOP/s
PEAK_BF16_MFMA_TFLOPS = 2500.0
SPLIT_PRODUCTS = 6
# the arithmetic type of the path: fp32 in / out / accumulate (how the fp32 products are formed is `dtype_note`'s business)
DTYPE = "f32"
PEAK_HBM_GBS = 8000.0  # same guide: HBM3E 8 TB/s spec (6.3 TB/s achievable)
PROFILE_ROUNDS = ("r06", "r05", "r04", "r03", "r02", "r01")  # committed rocprofv3 summaries, newest first
sys.path.insert(0, os.path.join(ROOT, "tools"))
from kernel_family import FAMILY_KERNELS, base_name  # noqa: E402  (= ivln_family_kernel_names(): ONE definition of the family)

MFMA_FAMILY = "MFMA family (" + " / ".join(FAMILY_KERNELS) + ")"


def log(*a):
    print("[bench]", *a, file=sys.stderr, flush=True)


def make_policy(device, seed=0):
    from ivln_ce_amd.config import get_config
    from ivln_ce_amd.policy import MapCMAPolicy
    from ivln_ce_amd.spaces import Box, Dict, Discrete

    cfg = get_config(opts=[
        "MODEL.policy_name", "MapCMAPolicy", "MODEL.INSTRUCTION_ENCODER.use_pretrained_embeddings", False,
        "MODEL.DEPTH_ENCODER.ddppo_checkpoint", "NONE",
        "MODEL.PROGRESS_MONITOR.use", True,  # the MapCMA experiment YAMLs switch it on (0_train_tf.yaml:30-34)
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


def to_dev(obs_list, dev):
    return [{k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in o.items()} for o in obs_list]


class GemmTimer:
    """Kernel time and algorithmic FLOPs of the MFMA-family launches made while it is active.  Time: the library's
    duration sink (ivln_family_timing_begin / _end, include/ivln_hip.h) - every family launch goes out with a start /
    stop event of its own (hipExtLaunchKernelGGL: the dispatch's begin and end timestamps, the per-kernel figure
    rocprofv3 reports), summed.  FLOPs: counted at the Python entry points from the shapes."""

    MAX_LAUNCHES = 1 << 12

    def __init__(self):
        self.flops = 0
        self.launches = 0
        self.by_kernel = {}  # kernel name -> (launches, ms) of the pass, from the library's own sink
        self._ms = None

    def __enter__(self):
        import ctypes as C

        from ivln_ce_amd import ops
        from ivln_ce_amd._lib import check, lib

        self.ops, self._C, self._lib, self._check = ops, C, lib(), check
        self._lib.ivln_family_timing_begin.argtypes = [C.c_int]
        self._lib.ivln_family_timing_end.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        self.orig = ops.gemm

        def timed(desc):
            self.orig(desc)
            flags = getattr(desc, "_run_flags", None)
            if flags is not None:
                # per-image run flags (the folded attention operands of the cached instruction encoding): only the rows
                # whose tokens changed are computed - count what ran (a read-back: this pass is outside the timed region)
                self.flops += 2 * desc.M * desc.N * desc.K * float(flags.float().mean().item())
                return
            # (a stacked transposed conv multiplies its classes' zero padding too: ops.conv_transpose2d_s2 states the real taps)
            self.flops += getattr(desc, "_algo_flops", 2 * desc.M * desc.N * desc.K)

        self.orig_soft = ops.gemm_soft

        def timed_soft(desc):
            """descriptors only some kernels take (a fused bottleneck tail: the 3x3 conv + the 1x1 conv behind it)"""
            ok = self.orig_soft(desc)
            if ok:
                self.flops += 2 * desc.M * desc.N * desc.K + 2 * desc.fuse_M * desc.N * desc.M * (1 if desc.fuse_A_split else 0)
            return ok

        self.orig_gn_conv = ops.gn_conv

        def timed_gn_conv(x, gn, **kw):
            """GroupNorm + next-conv launch of the depth ResNet chain: its convs run on the same matrix cores"""
            r = self.orig_gn_conv(x, gn, **kw)
            if r is not None:
                if kw.get("front") is not None:  # the 1x1 conv of the front stage (full K, per group block)
                    x0, _, w0 = kw["front"]
                    self.flops += 2 * w0.shape[0] * w0.shape[1] * x0.N * x0.H * x0.W
                for y, cw in ((r[1], kw.get("conv_a")), (r[2], kw.get("conv_b"))):
                    if y is not None:
                        w = cw[0]
                        self.flops += 2 * w.shape[0] * (y.N * y.H * y.W) * w.shape[1] * w.shape[2] * w.shape[3]
            return r

        self.orig_nconv = ops.nconv

        def timed_nconv(x, gn=None, **kw):
            """GroupNorm-on-load conv of layer 1 (k_nconv): full-K convs on the matrix cores"""
            r = self.orig_nconv(x, gn, **kw)
            if r is not None:
                for y, cw in ((r[1], kw.get("conv_a")), (r[2], kw.get("conv_b"))):
                    if y is not None:
                        w = cw[0]
                        self.flops += 2 * w.shape[0] * (y.y.shape[1] * y.y.shape[2] * y.y.shape[3]) * w.shape[1] * w.shape[2] * w.shape[3]
            return r

        from ivln_ce_amd import depth_net

        self.depth_net = depth_net
        self.orig_dn_run = depth_net.DepthNetPlan.run
        timer = self

        def timed_dn_run(plan, depth, out, out_img_stride):
            """the whole depth encoder as one persistent launch (k_depth_net): its 53 convs run on the same matrix cores"""
            ok = timer.orig_dn_run(plan, depth, out, out_img_stride)
            if ok:
                timer.flops += plan.prog.flops_per_image * depth.shape[0]
            return ok

        depth_net.DepthNetPlan.run = timed_dn_run
        ops.gemm = timed
        ops.gemm_soft = timed_soft
        ops.gn_conv = timed_gn_conv
        ops.nconv = timed_nconv
        self._lib.ivln_conv_split_counters.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_longlong), C.c_int]
        self._lib.ivln_conv_split_counters(None, None, 1)  # the convs that take the split-bf16 kernel are tallied by the library
        self.split_flops, self.split_launches = 0.0, 0
        self._check(self._lib.ivln_family_timing_begin(self.MAX_LAUNCHES), "ivln_family_timing_begin")
        return self

    def __exit__(self, *a):
        self.ops.gemm = self.orig
        self.ops.gemm_soft = self.orig_soft
        self.ops.gn_conv = self.orig_gn_conv
        self.ops.nconv = self.orig_nconv
        self.depth_net.DepthNetPlan.run = self.orig_dn_run
        if self._ms is None:
            self.total_ms()

    def total_ms(self):
        """Sum of the family's kernel durations since __enter__ (waits for them); closes the sink."""
        if self._ms is None:
            C = self._C
            ms, n, dropped = C.c_double(0.0), C.c_int(0), C.c_int(0)
            self._check(self._lib.ivln_family_timing_end(C.byref(ms), C.byref(n), C.byref(dropped)),
                        "ivln_family_timing_end")
            if dropped.value:
                raise RuntimeError(f"GemmTimer: {dropped.value} launches beyond MAX_LAUNCHES went untimed")
            self._ms, self.launches = ms.value, n.value
            self._lib.ivln_family_timing_report.argtypes = [C.c_char_p, C.c_int]
            buf = C.create_string_buffer(4096)
            self._check(self._lib.ivln_family_timing_report(buf, 4096), "ivln_family_timing_report")
            for line in buf.value.decode().splitlines():
                name, cnt, kms = line.split()
                self.by_kernel[name] = (int(cnt), float(kms))
            f, k = C.c_double(0.0), C.c_longlong(0)
            self._lib.ivln_conv_split_counters(C.byref(f), C.byref(k), 0)
            self.split_flops, self.split_launches = f.value, k.value
        return self._ms


def stats_of(ms_list):
    """median / min / max of the repetitions' per-step times."""
    v = sorted(ms_list)
    med = v[len(v) // 2] if len(v) % 2 else 0.5 * (v[len(v) // 2 - 1] + v[len(v) // 2])
    return med, v[0], v[-1]


def mixed_bound(flops, split_flops):
    """(bound in ms, peak in TFLOP/s) of a FLOP mix on the matrix cores: the part that ran on the fp32 MFMA kernels is
    priced on the fp32 MFMA peak, the part that ran on the split-bf16 kernels (csrc/conv_bf3.hip) on the bf16 peak at SIX
    executed bf16 FLOPs per algorithmic fp32 FLOP.  `peak` = flops / bound: the rate this mix would run at with both pipes
    at their dense peaks - the denominator of every `roofline.frac` in the line (which a correct FLOP count cannot push past 1)."""
    rest = max(flops - split_flops, 0.0)
    bound_ms = (rest / (PEAK_F32_MFMA_TFLOPS * 1e12) + split_flops * SPLIT_PRODUCTS / (PEAK_BF16_MFMA_TFLOPS * 1e12)) * 1e3
    peak = (flops / (bound_ms * 1e-3)) / 1e12 if bound_ms > 0 else PEAK_F32_MFMA_TFLOPS
    return bound_ms, peak


PEAKS = {"f32_mfma_tflops": PEAK_F32_MFMA_TFLOPS, "bf16_mfma_tflops": PEAK_BF16_MFMA_TFLOPS,
         "bf16_flops_executed_per_split_flop": SPLIT_PRODUCTS,
         "source": "/opt/skills/guides/MI355X_MICROARCH.md (dense peaks: v_mfma_f32_32x32x2_f32, v_mfma_f32_32x32x16_bf16)"}


def split_bf16_part(flops, split_flops, split_launches, n_steps):
    """The family's convs that ran on the split-bf16 kernels: their algorithmic fp32 FLOPs are inside `flops_per_step` /
    `achieved` like everybody else's, but the matrix cores executed SIX bf16 FLOPs for each of them (`mixed_bound`)."""
    sf = split_flops / n_steps
    if sf <= 0:
        return None
    return {
        "algorithmic_flops_per_step": int(sf), "share_of_family_flops": round(sf / (flops / n_steps), 4),
        "launches_per_step": round(split_launches / n_steps, 1),
        "executed_bf16_flops_per_step": int(sf * SPLIT_PRODUCTS),
        "what": "3x3 / 7x7 convs with both operands as three bf16 pieces each, six piece products per fp32 product on "
                "v_mfma_f32_32x32x16_bf16, fp32 accumulation: as close to the exact conv as the fp32 MFMA kernels (tests)",
    }


def pmc_mfma_busy(name):
    """MFMA-pipe busy fraction of EVERY family kernel of a committed PMC pass (profiles/rNN_<name>: one rocprofv3 run with
    `--pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES`, summarised by tools/pmc_stats.py): busy cycles of the 1024 matrix
    pipes / (1024 x SQ_BUSY_CYCLES / 32 shader engines), per kernel name (its template instantiations summed; the family is
    tools/kernel_family.py's list, not a prefix match - round 5 merged k_conv_bf3 with k_conv_bf3_ks and hid
    k_conv1x1_bf3_ks).  `kernel` / `mfma_busy` = the member with the most SQ-busy cycles in the pass.  A counter pass
    cannot run inside the timed bench: the figures are the committed ones, None without a file."""
    import csv

    for rnd in PROFILE_ROUNDS:
        path = os.path.join(ROOT, "profiles", f"{rnd}_{name}")
        if not os.path.exists(path):
            continue
        try:
            busy, sq = {}, {}
            for row in csv.DictReader(open(path)):
                base = base_name(row["Name"])
                d = busy if row["Counter"] == "SQ_VALU_MFMA_BUSY_CYCLES" else sq
                d[base] = d.get(base, 0.0) + float(row["Total"])
            fam = [k for k in FAMILY_KERNELS if sq.get(k, 0.0) > 0]
            if not fam:
                return None
            total = sum(sq[k] for k in fam)
            table = {k: {"mfma_busy": round(busy.get(k, 0.0) / (1024.0 * sq[k] / 32.0), 4),
                         "share_of_family_sq_busy_cycles": round(sq[k] / total, 4)} for k in fam}
            k = max(fam, key=lambda n: sq[n])
            return {"kernel": k, "mfma_busy": table[k]["mfma_busy"], "by_kernel": table,
                    "family_weighted": round(sum(busy.get(k, 0.0) for k in fam) / (1024.0 * total / 32.0), 4),
                    "source": f"profiles/{rnd}_{name}",
                    "formula": "SQ_VALU_MFMA_BUSY_CYCLES / (1024 pipes x SQ_BUSY_CYCLES / 32 shader engines) per kernel name; "
                               "`kernel` = the member with the most SQ-busy cycles; `family_weighted` = the same ratio over "
                               "the whole family"}
        except Exception:  # noqa: BLE001
            return None
    return None


def mfma_roofline(gt, ms, n_steps, traffic, what, wall_ms_per_step=None, busy=None, per="step"):
    """The contract's roofline object for the MFMA family of one step (or update).
      achieved  algorithmic FLOPs of one step / the step's WALL time (`wall_ms_per_step`: the timed, median repetition -
                the step overlaps streams, so summed kernel durations are not a denominator), TFLOP/s
      peak      the rate THIS FLOP mix would run at with the fp32 MFMA pipe and the bf16 MFMA pipe at their dense peaks
                (`mixed_bound`; the per-pipe peaks are in `peaks`)
      frac      achieved / peak = bound time / wall time (NOT clamped: a value above 1 is an accounting error and
                `frac_exceeds_1` says so - tests/test_gpu_bench.py fails on it)
    Secondary figures: `fp32_peak_basis` (the same FLOPs priced on the fp32 MFMA peak alone - rounds 1-4's `frac`; a
    RATIO that can pass 1 where the split-bf16 kernels run faster than the fp32 pipe could), `kernel_time` (the family's
    summed kernel durations from the instrumented single-stream pass instead of wall time), `mfma_busy` (committed PMC
    pass).  `traffic` arrives as the committed PMC figure per LAUNCH; the per-step total is spelled out beside it."""
    flops_step = gt.flops / n_steps
    split_step = getattr(gt, "split_flops", 0.0) / n_steps
    bound_ms, peak = mixed_bound(flops_step, split_step)
    k_ms = ms / n_steps
    wall = wall_ms_per_step if wall_ms_per_step else k_ms
    ach = (flops_step / (wall * 1e-3)) / 1e12 if wall > 0 else 0.0
    k_ach = (flops_step / (k_ms * 1e-3)) / 1e12 if k_ms > 0 else 0.0
    per_launch, per_step, profile = (tuple(traffic) + (None,))[:3] if isinstance(traffic, tuple) else (traffic, None, None)
    return {
        "bound": "mfma", "achieved": round(ach, 3), "peak": round(peak, 2), "unit": "TFLOP/s",
        "frac": round(bound_ms / wall, 5) if wall > 0 else None, "frac_exceeds_1": bool(wall > 0 and bound_ms / wall > 1.0),
        "traffic": per_launch,
        "peaks": PEAKS, "bound_ms_per_" + per: round(bound_ms, 4),
        "traffic_unit": "HBM bytes per launch of the family (rocprofv3 PMC passes committed under profiles/)",
        "traffic_bytes_per_" + per: per_step,
        "traffic_profile": profile,  # the committed PMC summary the two figures come from, with its per-kernel launches / bytes
        "kernel": MFMA_FAMILY + ": " + what,
        "flops_per_" + per: int(flops_step), "launches_per_" + per: round(gt.launches / n_steps, 1),
        "basis": ("algorithmic FLOPs of one " + per + " / WALL time of the timed " + per + " (median repetition; every other "
                  "kernel and every gap of the " + per + " is inside the denominator), against the peak of its fp32 / "
                  "split-bf16 FLOP mix" if wall_ms_per_step else
                  "algorithmic FLOPs / summed kernel durations of the family, against the peak of its FLOP mix"),
        "split_bf16": split_bf16_part(gt.flops, getattr(gt, "split_flops", 0.0), getattr(gt, "split_launches", 0), n_steps),
        "mfma_busy": busy,
        "fp32_peak_basis": {"peak": PEAK_F32_MFMA_TFLOPS, "ratio_to_fp32_mfma_peak": round(ach / PEAK_F32_MFMA_TFLOPS, 5),
                            "note": "the same algorithmic FLOPs priced on the fp32 MFMA peak alone (what rounds 1-4 printed "
                                    "as `frac`): a ratio, not a fraction of a bound - the split-bf16 convs run on a pipe "
                                    "with 16x the fp32 pipe's rate at 6x the work"},
        "kernel_time": {
            "achieved": round(k_ach, 3), "frac": round(bound_ms / k_ms, 5) if k_ms > 0 else None,
            "kernel_ms_per_" + per: round(k_ms, 4),
            "by_kernel": {k: {"launches_per_" + per: round(c / n_steps, 3), "ms_per_" + per: round(t / n_steps, 4)}
                          for k, (c, t) in sorted(getattr(gt, "by_kernel", {}).items(), key=lambda kv: -kv[1][1])},
            "time_basis": "sum of the family's kernel durations - a start / stop HIP event on every dispatch "
                          "(hipExtLaunchKernelGGL through ivln_family_timing_begin / _end), the per-kernel figure rocprofv3 "
                          "reports - in an instrumented EAGER single-stream pass outside the timed region",
        },
    }


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


def cpu_baseline(obs_cpu, B, budget_s=12.0, pred=False):
    """Torch-CPU policy port + C mapper oracle on the host cores (kind = "port"); with `pred` the labels come
    from the torch-CPU RedNet port (configs[2])."""
    from oracle.mapper_ref import MapperRef
    from oracle.policy_ref import MapCMAPolicyRef

    torch.manual_seed(0)
    ncores = usable_cores()
    torch.set_num_threads(ncores)
    pol = MapCMAPolicyRef().eval()
    mapper = MapperRef(256, 256)
    rnn = torch.zeros(B, 2, 512)
    prev = torch.zeros(B, 1, dtype=torch.long)
    rednet = None
    if pred:
        from oracle.rednet_ref import RedNetRef, predict_semantics_ref

        rednet = RedNetRef().eval()

    def step(o):
        nonlocal rnn, prev
        if pred:
            labels = predict_semantics_ref(rednet, o["rgb"], o["depth"])[1].numpy()
        else:
            labels = o["semantic12"].numpy()
        occ, sem = mapper.step(o["depth"].numpy(), labels, o["world_robot_pose"].numpy(),
                               o["world_robot_orientation"].numpy(), o["not_done_masks"].numpy())
        ob = {"depth": o["depth"], "instruction": o["instruction"], "occupancy_map": torch.from_numpy(occ),
              "semantic_map": torch.from_numpy(sem)}
        with torch.no_grad():
            a, rnn, _ = pol.act(ob, rnn, prev, o["not_done_masks"])
        prev = a

    n_warm, n_min, n_max = (1, 6, 40) if pred else (3, 20, 400)
    for o in obs_cpu[:n_warm]:
        step(o)
    n, t0 = 0, time.perf_counter()
    i = n_warm
    while True:
        step(obs_cpu[i % len(obs_cpu)])
        i += 1
        n += 1
        el = time.perf_counter() - t0
        if (n >= n_min and el > budget_s) or n >= n_max or el > 3 * budget_s:
            break
    what = ("torch-CPU RedNet port + C mapper oracle + torch-CPU MapCMA port" if pred
            else "C mapper oracle + torch-CPU MapCMA port")
    return {
        "value": round(B * n / el, 2), "unit": "env-steps/s", "cores": ncores, "kind": "port",
        "sample": f"{n} steps of {B} envs (256x256 depth, " + ("224x224 rgb, " if pred else "gt semantics, ")
                  + f"80-token instruction): {what}, {ncores} threads, after {n_warm} warm-up step(s)",
    }


def pmc_traffic_pair(name, family="mfma_family"):
    """(bytes per launch, bytes per step, {source, by_kernel}) of a kernel family from a committed PMC summary, or None."""
    a = pmc_traffic(name, (family, "hbm_bytes_per_launch_corrected"))
    b = pmc_traffic(name, (family, "hbm_bytes_per_step_corrected"))
    if a is None:
        return None
    src = next((f"profiles/{r}_{name}" for r in PROFILE_ROUNDS if os.path.exists(os.path.join(ROOT, "profiles", f"{r}_{name}"))), None)
    return (a, b, {"source": src, "by_kernel": pmc_traffic(name, (family, "by_kernel")),
                   "family_launches_per_step": pmc_traffic(name, (family, "launches_per_step"))})


def pmc_traffic(name, key):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE in separate runs,
    gfx950 FETCH correction applied; profiles/<name>).  A counter pass cannot run inside the timed bench, so the
    figure is the committed one and only for its workload; None when no such profile is committed."""
    for rnd in PROFILE_ROUNDS:
        path = os.path.join(ROOT, "profiles", f"{rnd}_{name}")
        if os.path.exists(path):
            try:
                d = json.load(open(path))
                for k in key:
                    d = d[k]
                return d
            except Exception:  # noqa: BLE001
                return None
    return None


def inflection_weights(targets_TN, coef=3.2):
    """dagger_trainer.py:193-214: weight `coef` where the expert action differs from the previous step's (the first
    step counts as an inflection), 1 elsewhere."""
    infl = torch.ones_like(targets_TN, dtype=torch.bool)
    infl[1:] = targets_TN[1:] != targets_TN[:-1]
    return torch.where(infl, torch.tensor(coef), torch.tensor(1.0))


def update_batch(T, N, seed=7):
    """SURVEY section 8d's synthetic update batch on the host: (obs incl. de-duplicated, trimmed instructions, prev, nd,
    targets, weights) plus the un-trimmed per-row instruction tensor the oracle takes."""
    from ivln_ce_amd.utils import dedupe_instructions, trim_instruction_padding

    g = torch.Generator().manual_seed(seed)
    TN = T * N
    instr = torch.zeros(N, 200)
    instr[:, :80] = torch.randint(2, 2504, (N, 80), generator=g).float()
    # the trainer's loader drops the all-padding tail of the token batch on the host (trainers.PrefetchLoader):
    # like the reference's packed LSTM, the update only ever sees the batch's longest instruction (80 of 200)
    # ... and hands the policy the batch's UNIQUE token rows + each row's index (one encoding per trajectory)
    host = dedupe_instructions(trim_instruction_padding({"instruction": instr.repeat(T, 1)}, first_rows=N))
    obs = {"depth_features": torch.randn(TN, 128, 4, 4, generator=g),
           "occupancy_map": (torch.rand(TN, 64, 64, generator=g) < 0.3).float(),
           "semantic_map": torch.randint(0, 13, (TN, 64, 64), generator=g).float(),
           "progress": torch.rand(TN, 1, generator=g)}
    obs.update({k: v.float() for k, v in host.items()})  # batch_to casts every observation to float32
    prev = torch.randint(0, 4, (TN, 1), generator=g)
    nd = torch.ones(T, N, dtype=torch.uint8)
    nd[0] = 0
    nd = nd.view(-1, 1)
    tgt = torch.randint(0, 4, (T, N), generator=g)
    return obs, prev, nd, tgt, inflection_weights(tgt), instr.repeat(T, 1)


class UpdateLeg:
    """DAgger update step (base_il_trainer.py:173-219) on SURVEY section 8d's synthetic batch: forward over T*N rows
    with BPTT, inflection-weighted CE + progress-monitor aux loss (quirk Q7), hand-written HIP backward, one
    flat-bucket RCCL all-reduce (world > 1), Adam."""

    def __init__(self, policy, dev, world, T=64, N=8):
        from ivln_ce_amd.trainers import FlatAdam

        self.policy, self.dev, self.world, self.T, self.N = policy, dev, world, T, N
        policy.train()
        self.opt = FlatAdam(policy, lr=2.5e-4)
        obs, prev, nd, tgt, w, self.instr_rows = update_batch(T, N)
        self.host = (obs, prev, nd, tgt, w)
        self.args = ({k: v.to(dev) for k, v in obs.items()}, prev.to(dev), nd.to(dev), tgt.to(dev), w.to(dev))

    def once(self):
        from ivln_ce_amd.trainers import update_agent

        return update_agent(self.policy, self.opt, *self.args, world=self.world)

    def timed(self, barrier, max_over_ranks, iters=5, warm=2, reps=5):
        """`reps` repetitions of `iters` updates, each bracketed by the barrier; -> list of ms per update."""
        from ivln_ce_amd import dist as D
        from ivln_ce_amd.aux_losses import AuxLosses

        AuxLosses.activate()
        try:
            for _ in range(warm):
                self.once()
            out = []
            D.ALLREDUCE_EVENTS = [] if self.world > 1 else None  # (event pairs around the collective of the timed updates)
            for _ in range(reps):
                barrier()
                t0 = time.perf_counter()
                for _ in range(iters):
                    self.once()
                barrier()
                out.append(1e3 * max_over_ranks(time.perf_counter() - t0) / iters)
        finally:
            AuxLosses.deactivate()
            self._allreduce = self.allreduce_object()
            D.ALLREDUCE_EVENTS = None
        return out

    def roofline(self, wall_ms_per_update=None):
        """MFMA kernel family of one update: `frac` on the update's WALL time (the timed median repetition) against the
        peak of its fp32 / split-bf16 FLOP mix; the summed kernel durations of the instrumented pass are the secondary
        `kernel_time`.  The durations are summed, so that pass runs everything on one stream: with the instruction
        branch on its side stream (the timed configuration) concurrent launches would be counted twice."""
        from ivln_ce_amd import train as _train
        from ivln_ce_amd.aux_losses import AuxLosses

        overlap, _train.OVERLAP_INSTRUCTION = _train.OVERLAP_INSTRUCTION, False
        AuxLosses.activate()
        try:
            with GemmTimer() as gt:
                self.once()
                ms = gt.total_ms()
        finally:
            _train.OVERLAP_INSTRUCTION = overlap
            AuxLosses.deactivate()
        return mfma_roofline(gt, ms, 1, pmc_traffic_pair("update_pmc_traffic.json"),
                             "k_conv_bf3 / k_wgrad_bf3 (split-bf16: the map CNN's forward convs, input gradients and weight "
                             "gradients), k_gemm_vec / k_gemm (fp32 MFMA: linears, Conv1d, their gradients) of one update",
                             wall_ms_per_step=wall_ms_per_update, busy=pmc_mfma_busy("update_pmc_mfma_util.csv"), per="update")

    def allreduce_object(self):
        """The update's ONE collective (flat fp32 gradient bucket, sum) by itself: event pair around the call inside
        the timed updates (ivln_ce_amd.dist.ALLREDUCE_EVENTS).  Populated when world > 1."""
        from ivln_ce_amd import dist as D

        nbytes = int(self.opt.grad.numel()) * 4
        obj = {"bytes": nbytes, "calls_per_update": 1, "world": self.world, "ms": None, "bus_GBps": None, "algo_GBps": None,
               "what": "torch.distributed.all_reduce(SUM) of the flat gradient bucket between backward and Adam "
                       "(backend nccl = RCCL over xGMI; 1/world folded into the Adam kernel); ms = median over the timed "
                       "updates of a HIP event pair around the call on the update's stream; bus = 2 (n-1)/n x bytes / time"}
        ev = D.ALLREDUCE_EVENTS
        if self.world > 1 and ev:
            torch.cuda.synchronize()
            t = sorted(a.elapsed_time(b) for a, b in ev)
            med = t[len(t) // 2]
            obj["ms"] = round(med, 4)
            obj["samples"] = len(t)
            if med > 0:
                obj["algo_GBps"] = round(nbytes / (med * 1e-3) / 1e9, 2)
                obj["bus_GBps"] = round(2.0 * (self.world - 1) / self.world * nbytes / (med * 1e-3) / 1e9, 2)
        return obj

    def cpu_baseline(self, budget_s=25.0):
        """The oracle's update on the host cores: `MapCMAPolicyRef.update_loss` (base_il_trainer.py:173-219 restated
        in torch CPU) + autograd backward + torch.optim.Adam on the same T x N batch."""
        from oracle.policy_ref import MapCMAPolicyRef

        ncores = usable_cores()
        torch.set_num_threads(ncores)
        torch.manual_seed(0)
        ref = MapCMAPolicyRef(use_pm=True).train()
        opt = torch.optim.Adam(ref.parameters(), lr=2.5e-4)
        obs, prev, nd, tgt, w = self.host
        obs = {k: v for k, v in obs.items() if not k.startswith("instruction_")}
        obs["instruction"] = self.instr_rows  # (the oracle encodes every row's own 200-token instruction tensor)

        def one():
            opt.zero_grad()
            loss = ref.update_loss(obs, prev, nd, tgt, w)[0]
            loss.backward()
            opt.step()

        one()  # warm-up
        n, t0 = 0, time.perf_counter()
        while True:
            one()
            n += 1
            el = time.perf_counter() - t0
            if el > budget_s or n >= 8:
                break
        rows = self.T * self.N
        return {"value": round(rows * n / el, 2), "unit": "rows/s", "cores": ncores, "kind": "port",
                "ms_per_update": round(1e3 * el / n, 1),
                "sample": f"{n} update(s) of T={self.T} x N={self.N} rows after 1 warm-up: oracle/policy_ref.py update_loss + "
                          f"autograd backward + torch.optim.Adam, {ncores} threads"}


class CollectLeg:
    """One step of a DAgger COLLECTION (dagger_trainer.py:416-494; configs[3]'s per-GPU shard of 8 envs): mapper +
    `policy.act(deterministic=False)` + beta-mixing with the expert + the -1 rule, then what the loop keeps of the step
    on the host - actions, the two maps, the frozen depth encoder's features - through trainers._RolloutStepper,
    exactly as `_update_dataset` drives it (policy in train mode: quirk Q6).  Env stepping and trajectory storage
    are host work outside the hot path.  Observations are resident in HBM when the clock starts."""

    def __init__(self, cfg, policy, dev, rank, B=8, n_pool=110, graph=True, seed=99):
        from ivln_ce_amd import trainers
        from ivln_ce_amd.obs_transforms import GTSemanticsIterativeMapper

        cfg = cfg.clone()
        cfg.defrost()
        cfg.IL.DAGGER.USE_HIP_GRAPH = bool(graph)
        cfg.freeze()
        self.trainers = trainers
        tr = trainers.DaggerTrainer.__new__(trainers.DaggerTrainer)
        tr.config, tr.device, tr.policy = cfg, dev, policy
        tr.rank, tr.local_rank, tr.world = rank, dev.index or 0, 1
        tr.obs_transforms = [GTSemanticsIterativeMapper.from_config(cfg)]
        self.tr, self.B, self.dev, self.policy, self.n_pool = tr, B, dev, policy, n_pool
        self.uuid = cfg.IL.DAGGER.expert_policy_sensor_uuid
        g = torch.Generator().manual_seed(seed + rank)
        self.obs = to_dev(gen_observations(B, n_pool, seed=seed + rank), dev)
        for o in self.obs:
            o[self.uuid] = torch.randint(0, 4, (B, 1), generator=g).double().to(dev)
        self.stepper = None
        self.i = 0

    def open(self):
        """A collection phase starts (`_update_dataset`): policy in train mode, a fresh stepper (fresh capture)."""
        self.was_training = self.policy.training
        self.policy.train()
        self.stepper = self.trainers._RolloutStepper(self.tr, beta=0.75, expert_uuid=self.uuid, iterative=False)
        self.rnn = torch.zeros(self.B, 2, 512, device=self.dev)
        self.prev = torch.zeros(self.B, 1, dtype=torch.long, device=self.dev)

    def step(self):
        batch = dict(self.obs[self.i % self.n_pool])
        self.i += 1
        if not self.stepper.use_graph:
            batch = self.tr.obs_transforms[0](batch)
        with torch.no_grad():
            self.prev, self.rnn, host = self.stepper.step(batch, self.rnn, self.prev, (batch["not_done_masks"],))
        return host

    def close(self):
        used = self.stepper.use_graph
        self.tr.obs_transforms[0].mapping_module.check_status()
        self.stepper.close()
        self.stepper = None
        self.policy.train(self.was_training)
        return used

    def timed(self, barrier, max_over_ranks, K, W, reps):
        """-> (list of ms per step over `reps` repetitions of K steps, replayed as graphs?)"""
        self.open()
        try:
            for _ in range(W):
                self.step()
            out = []
            for _ in range(reps):
                barrier()
                t0 = time.perf_counter()
                for _ in range(K):
                    host = self.step()
                barrier()
                out.append(1e3 * max_over_ranks(time.perf_counter() - t0) / K)
            assert host["depth"].shape == (self.B, 128, 4, 4) and host["occ"].shape == (self.B, 64, 64) and len(host["actions"]) == self.B
        finally:
            used = self.close()
        return out, used


def dagger_iteration_leg(collect, update, barrier, max_over_ranks, n_collect=64, n_update=4, iters=5):
    """What a DAgger iteration alternates, in ONE process: a collection phase replayed as hipGraphs (`n_collect` steps
    after a fresh capture, like every `_update_dataset`) and `n_update` eager updates (ops.eager_work_stream).  Round 3
    saw updates run 25 % slower on a stream that had replayed graphs; this leg is the number of record for the
    alternation.  The capture itself (once per collection phase of thousands of steps in a real run) and its two
    warm-up steps are outside the clock."""
    from ivln_ce_amd.aux_losses import AuxLosses

    c_ms, u_ms = [], []
    for it in range(iters + 1):  # iteration 0 = warm-up
        collect.open()
        try:
            for _ in range(2):
                collect.step()
            barrier()
            t0 = time.perf_counter()
            for _ in range(n_collect):
                collect.step()
            barrier()
            t1 = time.perf_counter()
        finally:
            collect.close()
        AuxLosses.activate()  # (the trainer activates the progress-monitor loss around its updates only)
        try:
            barrier()
            t2 = time.perf_counter()
            for _ in range(n_update):
                update.once()
            barrier()
            t3 = time.perf_counter()
        finally:
            AuxLosses.deactivate()
        if it:
            c_ms.append(1e3 * max_over_ranks(t1 - t0) / n_collect)
            u_ms.append(1e3 * max_over_ranks(t3 - t2) / n_update)
    return c_ms, u_ms


def mapper_roofline(mapper_tr, obs_dev, B, n_steps=20):
    """HBM roofline of the egocentric mapper (north_star: "achieved HBM GB/s for the scatter against gfx950
    peak").  Event pair around the mapper's launches of one step, the GPU parked on a spin kernel while the host
    enqueues so that the pair times back-to-back kernels.  Algorithmic bytes per env-step as SURVEY section 8d
    defines them: 65 536 px x (4 B depth + 1 B label) in, the world cloud (x, y, z, batch, label = 17 B per point)
    read and written once, two 64x64 u8 maps out."""
    mm = mapper_tr.mapping_module

    def timed():
        evs = []
        for i in range(n_steps):
            o = dict(obs_dev[i % len(obs_dev)])
            torch.cuda._sleep(4_000_000)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            mapper_tr(o)
            b.record()
            evs.append((a, b))
        torch.cuda.synchronize()
        return 1e3 * sum(a.elapsed_time(b) for a, b in evs) / n_steps

    # the split replay launches the gt-semantics mapper narrow (it runs beside the depth-ResNet chain with time to spare,
    # graphed.py); the roofline is that of the kernels at full width, the in-step figure is reported next to it
    width = getattr(mm, "_width", (0, 0))
    us_in_step = timed() if width != (0, 0) else None
    mm.set_launch_width(0, 0)
    us = timed()
    mm.set_launch_width(*width)
    world_pts = mm.check_status()
    H, W = mm._hw
    bytes_step = B * (H * W * 5 + 2 * 64 * 64) + 2 * 17 * world_pts
    ach = bytes_step / (us * 1e-6) / 1e9 if us > 0 else 0.0
    traffic = pmc_traffic("rollout_pmc_traffic.json", ("mapper", "hbm_bytes_per_step_corrected")) if B == 4 else None
    return {
        "bound": "hbm", "achieved": round(ach, 2), "peak": PEAK_HBM_GBS, "unit": "GB/s",
        "frac": round(ach / PEAK_HBM_GBS, 5), "traffic": traffic,
        "kernel": "egocentric mapper (csrc/mapper.hip: unproject, keep-highest scatter-max, world merge, raster), "
                  "all launches of one step",
        "bytes_per_step": int(bytes_step), "us_per_step": round(us, 2), "world_points": int(world_pts), "envs": B,
        "us_per_step_as_launched_in_the_step": None if us_in_step is None else round(us_in_step, 2),
        "launch_width_in_the_step": list(width),
        "note": "launch-latency class: ~1 MB of algorithmic traffic per env-step (SURVEY section 8d)",
    }


def capture(policy, transforms, example, mode, rank):
    """GraphedRollout in the wanted mode, falling back to one stream if the split capture fails."""
    from ivln_ce_amd.graphed import GraphedRollout

    for attempt in ([mode, False] if mode else [False]):
        try:
            runner = GraphedRollout(policy, transforms, example, deterministic=True, streams=attempt)
            note = ("3 forked streams" if attempt is True else
                    "3 graphs on 2 streams" if attempt == "split" else "1 stream")
            return runner, note
        except Exception as e:  # noqa: BLE001 - a capture problem must not cost the measurement
            log(f"rank {rank}: graph capture ({attempt!r}) failed: {type(e).__name__}: {e}")
            torch.cuda.synchronize()
    return None, None


def kfd_gpu_nodes():
    """GPUs the kernel driver exposes, counted from sysfs (KFD topology nodes with SIMDs; CPU nodes have none).  None
    when the topology is not readable - the children then find out themselves."""
    base = "/sys/class/kfd/kfd/topology/nodes"
    try:
        n = 0
        for node in os.listdir(base):
            with open(os.path.join(base, node, "properties")) as f:
                props = dict(line.split()[:2] for line in f if len(line.split()) >= 2)
            n += int(props.get("simd_count", "0")) > 0
        return n
    except Exception:  # noqa: BLE001
        return None


def spawn_ranks(n):
    """--gpus N without a launcher: start N ranks (one per GPU) through torch.distributed.run BEFORE this process
    initialises HIP, stream their output through and exit with their status."""
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    if not env.get("IVLN_BENCH_ONE_DEVICE") and "--plumbing-only" not in sys.argv:
        have = kfd_gpu_nodes()  # a sysfs read: the parent never touches the HIP runtime
        if have is not None and have < n:
            log(f"--gpus {n} but only {have} GPU node(s) under /sys/class/kfd")
            return 2
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log("spawning", n, "ranks:", " ".join(cmd[1:]))
    return subprocess.call(cmd, env=env)


def plumbing_only(args, rank, world):
    """Launch-path check without a GPU (tests/test_host_logic.py): rendezvous over gloo, barrier, max-over-ranks of
    a dummy clock, rank 0 prints a line shaped like the real one but marked as NOT a measurement."""
    import torch.distributed as dist

    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        dist.barrier()
        t = torch.tensor([float(rank + 1)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        assert int(t.item()) == world
    if rank == 0:
        print(json.dumps({"metric": METRIC, "value": None, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps,
                          "warmup": args.warmup, "plumbing_only": True, "data": "none (launch-path check, no GPU work)"}),
              flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--reps", type=int, default=5, help="repetitions of the K-step timed region per leg (median reported)")
    ap.add_argument("--envs", type=int, default=4, help="parallel envs per GPU of the gt-semantics leg (configs[1]: 4)")
    ap.add_argument("--pred-envs", type=int, default=8, help="envs per GPU of the pred-semantics headline (configs[2]: 8)")
    ap.add_argument("--pred-semantics", action="store_true", help="(default since round 4) configs[2] is the headline")
    ap.add_argument("--gt-semantics", action="store_true",
                    help="profiling aid: make BASELINE configs[1] (gt semantics, --envs envs) the headline `value`")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pred-leg", action="store_true", help="skip the OTHER rollout leg (the one that is not the headline)")
    ap.add_argument("--no-gt-leg", action="store_true", help="same as --no-pred-leg")
    ap.add_argument("--no-update", action="store_true", help="skip the DAgger update-step leg (extra JSON object)")
    ap.add_argument("--no-collect", action="store_true", help="skip the DAgger collection-step and iteration legs")
    ap.add_argument("--collect-envs", type=int, default=8, help="envs per GPU of the collection leg (configs[3]: 64 / 8)")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay")
    ap.add_argument("--single-stream", action="store_true",
                    help="one graph on one stream instead of the default three graphs on two streams (depth ResNet || "
                         "mapper + map CNN + instruction encoder, then the head)")
    ap.add_argument("--streams", action="store_true",
                    help="fork the three encoder branches onto side streams inside the graph (measured SLOWER on "
                         "ROCm 7.2: cross-queue dependencies cost more than the overlap wins)")
    ap.add_argument("--only-update", action="store_true",
                    help="profiling aid: run the DAgger update leg alone and print its object (not the driver's line)")
    ap.add_argument("--plumbing-only", action="store_true",
                    help="exercise the launch / rendezvous path only (no GPU work, not a measurement)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args.gpus))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.plumbing_only:
        return plumbing_only(args, rank, world)
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

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    def max_over_ranks(seconds):
        if world > 1:
            t = torch.tensor([seconds], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item())
        return seconds

    K, W, R = args.steps, args.warmup, max(1, args.reps)
    cfg, policy = make_policy(dev)

    def rep_obj(ms_list, per="step"):
        med, lo, hi = stats_of(ms_list)
        return {"n": len(ms_list), f"ms_per_{per}_median": round(med, 4), f"ms_per_{per}_min": round(lo, 4),
                f"ms_per_{per}_max": round(hi, 4),
                "note": f"each repetition = barrier + synchronize, EXACTLY the stated number of {per}s, barrier + synchronize, "
                        "MAX over ranks; the leg's value is the median repetition"}

    if args.only_update:
        ul = UpdateLeg(policy, dev, world)
        ums = ul.timed(barrier, max_over_ranks, iters=max(5, min(K, 20)), reps=1)  # (a collective: every rank)
        roof = ul.roofline(ums[0])  # (EVERY rank: the instrumented update contains the gradient all-reduce)
        if rank == 0:
            print(json.dumps({"update_step": {"ms_per_update": round(ums[0], 3), "roofline": roof,
                                              "allreduce": ul._allreduce}}), flush=True)
        return
    mode = False if args.no_graph else (True if args.streams else (False if args.single_stream else "split"))

    def run_leg(pred, B, K, W, seed):
        """R repetitions of: K steps of one workload, barrier + synchronize on both sides, max over ranks (W warm-up
        steps once, after the capture).  Returns the leg's dict of results and what the instrumented passes need."""
        cls = PredictedSemanticsIterativeMapper if pred else GTSemanticsIterativeMapper
        tr = cls.from_config(cfg)
        n_pool = min(W + K * R, 240 if not pred else 32)
        obs_cpu = gen_observations(B, n_pool, seed=seed + rank, with_rgb=pred)
        obs_dev = to_dev(obs_cpu, dev)
        state = {"rnn": torch.zeros(B, 2, 512, device=dev), "prev": torch.zeros(B, 1, dtype=torch.long, device=dev)}
        runner, note = (None, None)
        if not args.no_graph:
            log(f"rank {rank}: capturing the {'pred-semantics ' if pred else ''}step graph ({B} envs)")
            runner, note = capture(policy, [tr], obs_dev[0], mode, rank)
        if runner is not None:
            def do_step(i):
                runner.step(obs_dev[i % n_pool])
        else:
            # eager launches (--no-graph: the PMC passes) must be the kernels of the TIMED step: the capture runs the policy's
            # depth encoder as conv + GroupNorm pairs beside RedNet and as the persistent launch otherwise (graphed.py);
            # left at its eager default the pred-semantics pass priced k_depth_net, which the replayed step never launches
            venc = getattr(getattr(policy.net, "depth_encoder", None), "visual_encoder", None)
            if venc is not None:
                venc.latency_bound = not pred

            def do_step(i):
                rollout_step(tr, policy, obs_dev[i % n_pool], state)
        for i in range(W):
            do_step(i)
        ms, i = [], W
        for _ in range(R):
            barrier()
            t0 = time.perf_counter()
            for _ in range(K):
                do_step(i)
                i += 1
            barrier()
            ms.append(1e3 * max_over_ranks(time.perf_counter() - t0) / K)
        tr.mapping_module.check_status()
        del runner  # graphs hold the activation pools
        return {"ms": ms, "tr": tr, "obs_cpu": obs_cpu, "obs_dev": obs_dev, "state": state, "n_pool": n_pool, "B": B,
                "launch": ("hipGraph replay, " + note) if note else "eager", "overlapped": note is not None and "2 streams" in note}

    def instrumented_mfma(leg, n_inst, what, traffic, busy=None):
        """MFMA-family figures of a leg from an eager pass with a start / stop event on every GEMM-family dispatch, the GPU
        parked on a spin kernel while the host enqueues each step so that the durations are those of back-to-back kernels
        (what rocprofv3's per-kernel durations show) and not the host's launch gaps.  Not part of any `value`."""
        from ivln_ce_amd.rednet import PredictSemantics

        # the timed configuration replays RedNet's recorded launch table through ONE C call (ivln_rednet_fwd), which
        # the per-launch event pairs cannot see: the instrumented pass walks the same launches from Python instead
        plan, PredictSemantics.USE_PLAN = PredictSemantics.USE_PLAN, False
        from ivln_ce_amd import ops

        try:
            # (on a stream that has not launched graphs: eager launches behind graph replays carry extra host time
            #  per launch, ops.eager_work_stream, which the event pairs would count as kernel time)
            with ops.eager_work_stream(), GemmTimer() as gt:
                for i in range(n_inst):
                    torch.cuda._sleep(12_000_000)
                    rollout_step(leg["tr"], policy, leg["obs_dev"][i % leg["n_pool"]], leg["state"])
                ms = gt.total_ms()
        finally:
            PredictSemantics.USE_PLAN = plan
        med = stats_of(leg["ms"])[0]
        return mfma_roofline(gt, ms, n_inst, traffic, what, wall_ms_per_step=med, busy=busy)

    def leg_object(leg, pred, what_cfg):
        med = stats_of(leg["ms"])[0]
        Bl = leg["B"]
        obj = {"value": round(world * Bl / (med * 1e-3), 2), "unit": "env-steps/s", "ms_per_step": round(med, 4),
               "steps": K, "warmup": W, "repetitions": rep_obj(leg["ms"]), "envs_per_gpu": Bl,
               "config": {"workload": what_cfg, "envs_per_gpu": Bl, "launch": leg["launch"]}}
        if rank == 0:
            if pred:
                obj["roofline"] = instrumented_mfma(leg, min(6, K), "RedNet + depth ResNet + map CNN launches of one step",
                                                    pmc_traffic_pair(f"predsem_B{Bl}_pmc_traffic.json"),
                                                    busy=pmc_mfma_busy(f"predsem_B{Bl}_pmc_mfma_util.csv"))
            else:
                obj["roofline"] = instrumented_mfma(leg, min(20, K), "all conv/linear launches of one step",
                                                    pmc_traffic_pair("rollout_pmc_traffic.json") if Bl == 4 else None,
                                                    busy=pmc_mfma_busy("rollout_pmc_mfma_util.csv") if Bl == 4 else None)
                obj["mapper_roofline"] = mapper_roofline(leg["tr"], leg["obs_dev"], Bl)
            if world == 1 and not args.no_cpu_baseline:
                log(f"cpu baseline ({'pred' if pred else 'gt'}-semantics) ...")
                obj["cpu_baseline"] = cpu_baseline(leg["obs_cpu"], Bl, budget_s=12.0, pred=pred)
        return obj

    W_PRED = ("BASELINE configs[2] (the largest single-GPU configuration): MapCMA pred-semantics eval step = RedNet(rgb "
              "224x224 -> 256x256, depth) -> arg-max labels -> egocentric mapper -> MapCMAPolicy.act, {B} parallel envs per GPU, "
              "256x256 depth + 224x224 rgb, 80-token instruction, random-init weights of the reference architecture")
    W_GT = ("BASELINE configs[1]: MapCMA gt-semantics eval step = egocentric mapper + MapCMAPolicy.act, {B} parallel envs per "
            "GPU, 256x256 depth + semantic12, 80-token instruction, random-init weights of the reference architecture")
    head_pred = not args.gt_semantics
    other = not (args.no_pred_leg or args.no_gt_leg)
    legs = {}
    for pred in ((True, False) if head_pred else (False, True)):
        if pred != head_pred and not other:
            continue
        Bl = args.pred_envs if pred else args.envs
        log(f"rank {rank}: {'configs[2] pred-semantics' if pred else 'configs[1] gt-semantics'} leg, {Bl} envs"
            + (" (headline)" if pred == head_pred else ""))
        leg = run_leg(pred, Bl, K, W, seed=4321 if pred else 1234)
        log(f"rank {rank}: ms per step over {R} repetitions of {K} steps: " + ", ".join(f"{m:.4f}" for m in leg["ms"]))
        legs[pred] = leg_object(leg, pred, (W_PRED if pred else W_GT).format(B=Bl))
        del leg

    # ---- DAgger update step (fwd + bwd + all-reduce + Adam): reported beside the headline ----
    update = ul = None
    if not args.no_update:
        log(f"rank {rank}: update-step leg")
        ul = UpdateLeg(policy, dev, world)
        ums = ul.timed(barrier, max_over_ranks, iters=5, warm=2, reps=R)
        umed = stats_of(ums)[0]
        rows = ul.T * ul.N
        update = {"value": round(world * rows / (umed * 1e-3), 1), "unit": "rows/s", "ms_per_update": round(umed, 3),
                  "repetitions": rep_obj(ums, per="update"), "updates_per_repetition": 5, "rows_per_update_per_gpu": rows,
                  "what": f"DAgger update T={ul.T} x N={ul.N} per GPU from cached depth features (SURVEY 8d "
                          "batch: inflection weights 3.2 from the targets, progress U(0,1)): MapCMA forward with BPTT, "
                          "inflection-weighted CE + progress-monitor aux loss, HIP backward, "
                          + ("one flat RCCL all-reduce, " if world > 1 else "") + "Adam"}
        update["allreduce"] = ul._allreduce
        # the instrumented pass is one more update_agent call, gradient all-reduce included: every rank runs it (rank 0 alone
        # left its collective unmatched - found by the 8-rank one-device test of round 6; over RCCL that is a hang)
        roof = ul.roofline(umed)
        if rank == 0:
            update["roofline"] = roof
            if world == 1 and not args.no_cpu_baseline:
                log("cpu baseline (update) ...")
                update["cpu_baseline"] = ul.cpu_baseline()

    # ---- DAgger collection step (sampled action + beta-mix + host copies): the rollout half of configs[3] ----
    collect = iteration = None
    if not args.no_collect:
        Bc, ck, cw = args.collect_envs, K, W
        log(f"rank {rank}: collection-step leg ({Bc} envs)")
        cl = CollectLeg(cfg, policy, dev, rank, B=Bc, graph=not args.no_graph)
        cms, used_graph = cl.timed(barrier, max_over_ranks, ck, cw, R)
        cmed = stats_of(cms)[0]
        collect = {"value": round(world * Bc / (cmed * 1e-3), 1), "unit": "env-steps/s", "ms_per_step": round(cmed, 4),
                   "steps": ck, "warmup": cw, "repetitions": rep_obj(cms), "envs_per_gpu": Bc,
                   "launch": "hipGraph replay, 3 graphs on 2 streams" if used_graph else "eager",
                   "what": "DAgger collection step of configs[3]'s per-GPU shard: gt-semantics mapper + MapCMAPolicy.act "
                           "SAMPLED on the device from host uniforms, beta-mixed with the expert action (beta = 0.75) "
                           "and zeroed where the expert says -1 in the action head's launch, policy in train mode "
                           "(BatchNorm batch statistics, quirk Q6); actions, both maps and the cached depth features "
                           "land in pinned host memory behind ONE synchronisation per step"}
        if used_graph and rank == 0 and world == 1:  # the same step as eager launches, for the record
            ce = CollectLeg(cfg, policy, dev, rank, B=Bc, graph=False)
            ems, _ = ce.timed(barrier, max_over_ranks, min(30, ck), 5, 1)
            collect["eager_ms_per_step"] = round(ems[0], 4)
            del ce
        if ul is not None and used_graph:
            log(f"rank {rank}: DAgger-iteration leg (64 replayed collection steps <-> 4 eager updates)")
            c_ms, u_ms = dagger_iteration_leg(cl, ul, barrier, max_over_ranks)
            cm, um = stats_of(c_ms)[0], stats_of(u_ms)[0]
            iteration = {
                "value": round(1e3 / (64 * cm + 4 * um), 3), "unit": "iterations/s (64 collection steps + 4 updates)",
                "ms_per_iteration": round(64 * cm + 4 * um, 3), "iterations": len(c_ms),
                "collect_ms_per_step": round(cm, 4), "collect_ms_per_step_min_max": [round(min(c_ms), 4), round(max(c_ms), 4)],
                "update_ms_per_update": round(um, 3), "update_ms_per_update_min_max": [round(min(u_ms), 3), round(max(u_ms), 3)],
                "legs_alone": {"collect_ms_per_step": collect["ms_per_step"], "update_ms_per_update": update["ms_per_update"],
                               "ms_per_iteration": round(64 * collect["ms_per_step"] + 4 * update["ms_per_update"], 3)},
                "what": "one process alternating a collection phase replayed as hipGraphs (fresh capture per phase, outside "
                        f"the clock; {Bc} envs, sampled actions, host copies) with 4 eager DAgger updates (T=64 x N=8) on "
                        "ops.eager_work_stream - the graph-replay / eager-update interaction of a real DAgger iteration; "
                        "`legs_alone` = the same two legs timed on their own earlier in this run"}
        del cl

    head = legs[head_pred]
    out = {
        "metric": METRIC, "value": head["value"], "unit": "env-steps/s", "n_gpus": world,
        "steps": K, "warmup": W, "ms_per_step": head["ms_per_step"], "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": DTYPE, "data": "synthetic",
        "dtype_note": "fp32 in, fp32 out, fp32 accumulation everywhere.  The kernels named k_*_bf3* (3x3 / 7x7 convs, the deep and "
                      "fused 1x1 convs, the stride-2 / transposed convs of RedNet, the map CNN's weight gradients) form their "
                      "fp32 products from three bf16 pieces per operand on the bf16 MFMA pipe (six piece products, the dropped "
                      "ones < 2^-23 of a product: error vs float64 at or below the fp32 MFMA kernels', tests/test_gpu_kernels.py); "
                      "every other kernel multiplies in fp32.  IVLN_SPLIT_BF16=0 keeps every conv on the fp32 MFMA kernels",
        "repetitions": head["repetitions"],
        "config": dict(head["config"], parallelism=f"dp{world} (envs sharded, no data-path collective)",
                       headline=("configs[2], the largest single-GPU configuration of BASELINE.json (configs[0] is the CPU "
                                 "plumbing case, configs[1] = `gt_semantics_step`, the DAgger update of configs[3]'s per-GPU "
                                 "shard = `update_step`)" if head_pred else "configs[1] by request (--gt-semantics)")),
        "roofline": head.get("roofline"),
    }
    if "cpu_baseline" in head:
        out["cpu_baseline"] = head["cpu_baseline"]
    if (not head_pred) and "mapper_roofline" in head:
        out["mapper_roofline"] = head["mapper_roofline"]
    if (not head_pred) in legs:
        out["gt_semantics_step" if head_pred else "pred_semantics_step"] = legs[not head_pred]
    if update is not None:
        out["update_step"] = update
    if collect is not None:
        out["dagger_collect_step"] = collect
    if iteration is not None:
        out["dagger_iteration"] = iteration
    out.update(ranks_observed(world, dev))
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def ranks_observed(world, dev):
    """What the collective itself saw, not what the environment said: `ranks_seen` = a sum all-reduce of
    ones over the process group on the bench's device, `devices_seen` = distinct (host, device uuid) pairs gathered from
    the ranks (1 under IVLN_BENCH_ONE_DEVICE, where every rank sits on cuda:0 over gloo), `backend` = the group's."""
    import socket

    props = torch.cuda.get_device_properties(dev)
    ident = f"{socket.gethostname()}:{getattr(props, 'uuid', None) or getattr(props, 'pci_bus_id', dev.index)}"
    if world <= 1:
        return {"ranks_seen": 1, "devices_seen": 1, "collective_backend": None}
    import torch.distributed as dist

    ones = torch.ones(1, device=dev, dtype=torch.float32)
    dist.all_reduce(ones, op=dist.ReduceOp.SUM)
    idents = [None] * world
    dist.all_gather_object(idents, ident)
    return {"ranks_seen": int(round(float(ones.item()))), "devices_seen": len(set(idents)),
            "collective_backend": dist.get_backend()}


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
