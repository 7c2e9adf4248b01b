"""The driver's bench line: shape of the JSON object at N = 1, and the N = 2 launch path on
one device (gloo, both ranks on cuda:0 - control flow only, not a measurement)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(args, env=None, timeout=900):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], cwd=ROOT, capture_output=True, text=True,
                       timeout=timeout, env=dict(os.environ, **(env or {})))
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
    assert len(lines) == 1, r.stdout[-1500:]
    return json.loads(lines[0])


def test_bench_line_carries_the_contract_at_one_gpu():
    d = _bench(["--steps", "4", "--warmup", "2", "--reps", "3", "--no-cpu-baseline"])
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 2 and d["unit"] == "env-steps/s"
    assert d["dtype"].startswith("f32") and "bf16x3" in d["dtype"]  # (the split-bf16 qualifier is part of the value)
    assert "configs[2]" in d["config"]["workload"] and d["config"]["envs_per_gpu"] == 8
    assert d["repetitions"]["n"] == 3
    assert d["repetitions"]["ms_per_step_min"] <= d["ms_per_step"] <= d["repetitions"]["ms_per_step_max"]
    assert abs(d["value"] - 8 / (d["ms_per_step"] * 1e-3)) < 0.01 * d["value"]
    roof = d["roofline"]

    def check_mixed(roof, flops_key, wall_ms):
        """`frac` = achieved / peak on the step's WALL time, where `peak` is the rate of the leg's fp32 / split-bf16 FLOP mix
        with each pipe at its dense peak - (fp32 FLOPs / 157.3 T + 6 x split FLOPs / 2500 T) / wall: never above 1."""
        assert roof["bound"] == "mfma" and roof["unit"] == "TFLOP/s" and 0.0 < roof["frac"] < 1.0
        pk = roof["peaks"]
        assert pk["f32_mfma_tflops"] == 157.3 and pk["bf16_mfma_tflops"] == 2500.0 and pk["bf16_flops_executed_per_split_flop"] == 6
        flops = roof[flops_key]
        sp = roof["split_bf16"]
        split = sp["algorithmic_flops_per_step"] if sp else 0
        bound_ms = ((flops - split) / 157.3e12 + 6 * split / 2500e12) * 1e3
        assert abs(roof["achieved"] - flops / (wall_ms * 1e-3) / 1e12) < 0.02 * roof["achieved"]
        assert abs(roof["frac"] - bound_ms / wall_ms) < 0.02 * roof["frac"]
        assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 0.02 * roof["frac"]
        assert 157.3 - 0.01 <= roof["peak"] <= 2500.0 / 6 + 0.01
        assert 0.0 < roof["kernel_time"]["frac"] <= 1.0
        assert roof["fp32_peak_basis"]["peak"] == 157.3 and roof["fp32_peak_basis"]["ratio_to_fp32_mfma_peak"] > 0
        return sp

    sp = check_mixed(roof, "flops_per_step", d["ms_per_step"])
    assert roof["traffic"] is not None
    # the convs that ran on the split-bf16 kernel are spelled out
    assert sp is not None and 0.0 < sp["share_of_family_flops"] < 1.0
    assert sp["executed_bf16_flops_per_step"] == 6 * sp["algorithmic_flops_per_step"] and "dtype_note" in d
    # MFMA-pipe busy of the dominant kernel from the committed PMC pass
    mb = roof["mfma_busy"]
    assert mb is not None and 0.0 < mb["mfma_busy"] <= 1.0 and mb["kernel"].startswith("k_") and mb["source"].startswith("profiles/")
    gt = d["gt_semantics_step"]
    assert "configs[1]" in gt["config"]["workload"] and gt["envs_per_gpu"] == 4
    check_mixed(gt["roofline"], "flops_per_step", gt["ms_per_step"])
    assert gt["mapper_roofline"]["bound"] == "hbm"
    up = d["update_step"]
    assert up["unit"] == "rows/s" and up["repetitions"]["n"] == 3 and up["roofline"]["traffic"] is not None
    # the update's fraction is on its WALL time too (the kernel-time figure is the secondary), and below 1
    check_mixed(up["roofline"], "flops_per_update", up["ms_per_update"])
    assert up["roofline"]["kernel_time"]["kernel_ms_per_update"] <= up["ms_per_update"] * 1.05
    ar = up["allreduce"]
    assert ar["bytes"] > 20e6 and ar["world"] == 1 and ar["ms"] is None  # (populated when world > 1)

    def no_frac_above_one(o):
        if isinstance(o, dict):
            for k, v in o.items():
                if k == "frac" and v is not None:
                    assert 0.0 <= v <= 1.0, (k, v)
                no_frac_above_one(v)

    no_frac_above_one(d)
    assert d["dagger_collect_step"]["envs_per_gpu"] == 8
    it = d["dagger_iteration"]
    assert it["iterations"] >= 3 and it["ms_per_iteration"] > 0 and it["legs_alone"]["ms_per_iteration"] > 0


def test_bench_gpus_2_on_one_device_reports_the_sum_over_ranks():
    d = _bench(["--gpus", "2", "--steps", "4", "--warmup", "2", "--reps", "2", "--gt-semantics", "--no-pred-leg", "--no-update",
                "--no-collect", "--no-cpu-baseline"], env={"IVLN_BENCH_ONE_DEVICE": "1"})
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["envs_per_gpu"] == 4
    # whole-job value = envs of BOTH ranks per (max-over-ranks) step time
    assert abs(d["value"] - 2 * 4 / (d["ms_per_step"] * 1e-3)) < 0.01 * d["value"]
