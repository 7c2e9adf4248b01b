"""Every non-default compute path an IVLN_* switch selects has to give the reference's answer too ("a place
where a later change can break bits silently").  The switches are read once per process - by the library (getenv in a
static) or at import (ivln_ce_amd.ops) - so each case runs the golden tests in a FRESH process with the switch in its
environment: the policy step against the reference's own `MapCMAPolicy` outputs (tests/golden/policy_act.npz), RedNet
against the reference's own network (rednet.npz), and - for the switches that touch training - one update against the
reference's loss and gradients (policy_update.npz); the bars are the ones written in those tests."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

ROLLOUT = ["tests/test_gpu_policy.py::test_act_matches_reference_golden", "tests/test_gpu_rednet.py::test_rednet_matches_reference_golden",
           "tests/test_gpu_policy.py::test_act_matches_oracle_other_batches"]
# (the split-bf16 kernel forms only engage at RedNet's real size: their switches also run the full-size comparison with the oracle)
FULLSIZE = ROLLOUT + ["tests/test_gpu_rednet.py::test_rednet_fullsize_matches_oracle_and_pred_mapper_is_exact"]
UPDATE = ["tests/test_gpu_train.py::test_reference_style_update_matches_golden",
          "tests/test_gpu_train.py::test_hip_update_agent_matches_reference_loss_and_moves_params"]

# (switch, value, which goldens)
CASES = [
    ("IVLN_SPLIT_BF16", "0", ROLLOUT + UPDATE),          # every conv on the fp32 MFMA kernels
    ("IVLN_SPLIT_BF16_1X1", "1", ROLLOUT),               # the split-bf16 1x1 form wherever eligible
    ("IVLN_SPLIT_BF16_1X1", "0", ROLLOUT),
    ("IVLN_NO_SPLIT_BF16_WGRAD", "1", UPDATE),           # fp32 MFMA weight gradients
    ("IVLN_BF3_KS", "0", FULLSIZE),                       # deep 3x3 convs on the tiled split-bf16 kernel + split-K slabs
    ("IVLN_BF3_KS", "1", FULLSIZE),                       # ... on the K-split-over-waves kernel wherever eligible
    ("IVLN_BF3_KS_TN", "1", FULLSIZE),                    # ... with 32-pixel tiles everywhere
    ("IVLN_BF3_KS_TN", "2", FULLSIZE),                    # ... with 64-pixel tiles everywhere
    ("IVLN_BF3_1X1_KS", "0", FULLSIZE),                   # stride-1 1x1 convs without the register-built forms
    ("IVLN_BF3_1X1_FORM", "ks", FULLSIZE),                # ... K split over waves wherever a form is taken
    ("IVLN_BF3_1X1_FORM", "wt", FULLSIZE),                # ... wave tiles wherever a form is taken
    ("IVLN_BF3_FUSE", "0", FULLSIZE),                     # bottleneck tails as two launches
    ("IVLN_BF3_FUSE_PX", "128", FULLSIZE),                # ... fused, 128 mid channels on the 128 x 128-pixel tile
    ("IVLN_BF3_FUSE_PX", "64", FULLSIZE),                 # ... on the 128 x 64-pixel tile
    ("IVLN_REDNET_SKIP_ADD", "0", FULLSIZE),              # the decoder's skip adds as launches of their own
    ("IVLN_BF3_NOSPLIT4", "0", FULLSIZE + UPDATE),        # the 64 x 128 tile split over the channel chunks as in round 4
    ("IVLN_DEPTH_NET", "0", ROLLOUT),                    # depth encoder: launch chain
    ("IVLN_DEPTH_NET", "2", ROLLOUT),                    # ... persistent launch everywhere
    ("IVLN_GN_CONV", "0", ROLLOUT),                      # ... conv + GroupNorm pairs (with IVLN_DEPTH_NET=0 below)
    ("IVLN_NCONV_FRONT", "0", ROLLOUT),
    ("IVLN_NO_VEC_GEMM", "1", ROLLOUT + UPDATE),         # scalar-gather implicit GEMM instead of the float4-staged one
    ("IVLN_NO_DIRECT_CONV", "1", ROLLOUT + UPDATE),      # no LDS-patch direct conv
    ("IVLN_NO_CONV1X1_STREAM", "1", ROLLOUT),
    ("IVLN_NO_WIDE_EPILOGUE", "1", ROLLOUT),
    ("IVLN_NO_XCD_REMAP", "1", ROLLOUT),
    ("IVLN_CONVT_STACK", "0", ROLLOUT),
    ("IVLN_REDNET_NO_GROUP", "1", ROLLOUT),
    ("IVLN_REDNET_PLAN", "0", ROLLOUT),
    ("IVLN_CMA_STEP_MODE", "-1", ROLLOUT),               # unfused recurrent head
    ("IVLN_KV_LINEAR", "0", ROLLOUT),
    ("IVLN_FOLD_GATES", "0", ROLLOUT),
    ("IVLN_CACHE_INSTRUCTION", "0", ROLLOUT),            # instruction re-encoded at every step
    ("IVLN_SEQ_PERSISTENT", "0", UPDATE),                # per-timestep GRU launches
    ("IVLN_NO_TRAIN_OVERLAP", "1", UPDATE),
    ("IVLN_WGRAD_OVERLAP", "1", UPDATE),                 # weight gradients on a side stream
    ("IVLN_LINEAR_BWD_NO_SPLIT", "1", UPDATE),
    ("IVLN_DIRECT_GRADS", "0", UPDATE),
    ("IVLN_COLSUM_MULTI", "0", UPDATE),
    ("IVLN_CONV_STATS", "0", UPDATE),
]


@pytest.mark.parametrize("switch,value,tests", CASES, ids=[f"{s}={v}" for s, v, _ in CASES])
def test_non_default_switch_keeps_the_goldens(switch, value, tests):
    env = dict(os.environ, **{switch: value})
    if switch == "IVLN_GN_CONV":  # (the pairs only run when the persistent launch does not take the encoder)
        env["IVLN_DEPTH_NET"] = "0"
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-p", "no:cacheprovider", *tests], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, f"{switch}={value}:\n" + r.stdout[-3000:] + r.stderr[-1500:]
    assert " passed" in r.stdout and "failed" not in r.stdout, r.stdout[-800:]
