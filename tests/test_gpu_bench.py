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
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 2 and d["unit"] == "env-steps/s" and d["dtype"] == "f32"
    assert "configs[2]" in d["config"]["workload"] and d["config"]["envs_per_gpu"] == 8
    assert d["repetitions"]["n"] == 3
    assert d["repetitions"]["ms_per_step_min"] <= d["ms_per_step"] <= d["repetitions"]["ms_per_step_max"]
    assert abs(d["value"] - 8 / (d["ms_per_step"] * 1e-3)) < 0.01 * d["value"]
    roof = d["roofline"]
    assert roof["bound"] == "mfma" and roof["peak"] == 157.3 and 0.0 < roof["frac"] < 1.0
    # wall-clock basis: achieved = flops per step / ms per step of the timed (median) repetition
    assert abs(roof["achieved"] - roof["flops_per_step"] / (d["ms_per_step"] * 1e-3) / 1e12) < 0.02 * roof["achieved"]
    assert roof["kernel_time"]["frac"] > 0 and roof["traffic"] is not None
    # the convs that ran on the split-bf16 kernel are spelled out, and priced on their own pipe rate the step stays below 1
    sp = roof["split_bf16"]
    assert sp is not None and 0.0 < sp["share_of_family_flops"] < 1.0 and 0.0 < sp["frac_of_mixed_bound"] < 1.0
    assert sp["executed_bf16_flops_per_step"] == 6 * sp["algorithmic_flops_per_step"] and "dtype_note" in d
    gt = d["gt_semantics_step"]
    assert "configs[1]" in gt["config"]["workload"] and gt["envs_per_gpu"] == 4 and gt["roofline"]["frac"] > 0
    assert gt["mapper_roofline"]["bound"] == "hbm"
    up = d["update_step"]
    assert up["unit"] == "rows/s" and up["repetitions"]["n"] == 3 and up["roofline"]["traffic"] is not None
    assert 0.0 < up["roofline"]["split_bf16"]["frac_of_mixed_bound"] < 1.0
    assert d["dagger_collect_step"]["envs_per_gpu"] == 8
    it = d["dagger_iteration"]
    assert it["iterations"] >= 3 and it["ms_per_iteration"] > 0 and it["legs_alone"]["ms_per_iteration"] > 0


def test_bench_gpus_2_on_one_device_reports_the_sum_over_ranks():
    d = _bench(["--gpus", "2", "--steps", "4", "--warmup", "2", "--reps", "2", "--gt-semantics", "--no-pred-leg", "--no-update",
                "--no-collect", "--no-cpu-baseline"], env={"IVLN_BENCH_ONE_DEVICE": "1"})
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["envs_per_gpu"] == 4
    # whole-job value = envs of BOTH ranks per (max-over-ranks) step time
    assert abs(d["value"] - 2 * 4 / (d["ms_per_step"] * 1e-3)) < 0.01 * d["value"]
