"""GPU-box run of the host-logic suites.  The rollout / eval control flow (A18), collate / IW dataset / block shuffle
(A19), the t-nDTW / nDTW / SDTW evaluators (8f-1) and the trajectory store + loader (8f-2) are pinned by CPU tests
(`-m "not gpu"`: tests/test_host_logic.py, test_loaders.py, test_oracle_*.py, test_dist_cpu.py, test_cabi_exports.py)
against goldens produced by the reference's own loops.  The round-end GPU run only selects `-m gpu`, so a regression
there would not show in it: this wrapper runs those suites as a child pytest on the GPU box and fails with their
output."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SUITES = ["test_host_logic.py", "test_loaders.py", "test_oracle_mapper.py", "test_oracle_policy.py", "test_oracle_twin.py", "test_oracle_rednet.py",
          "test_dist_cpu.py", "test_cabi_exports.py"]


@pytest.mark.gpu
def test_cpu_host_logic_suites_pass_on_the_gpu_box():
    cmd = [sys.executable, "-m", "pytest", "-x", "-q", "-m", "not gpu", "-p", "no:cacheprovider",
           *[os.path.join(ROOT, "tests", f) for f in SUITES]]
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=1500)
    tail = r.stdout[-3000:] + r.stderr[-2000:]
    assert r.returncode == 0, tail
    assert " passed" in r.stdout and " failed" not in r.stdout, tail
    print(r.stdout.strip().splitlines()[-1])
