"""The wide-store hazard behind BF3_STORE_GUARD (csrc/conv_bf3.hip).

CPU: every 12- / 16-byte buffer store with a REGISTER soffset in the built gfx950 code - the form the compiler's hazard
recognizer does not pad - has at least two wait states before anything writes its data registers
(tools/check_store_hazard.py walks the disassembly), and the checker does find such sites when the guard is compiled out.
GPU: tools/store_hazard.hip, the minimal reproduction - store, VALU overwrite of the data registers, in inline assembly -
leaves no stale store with the guard's `s_nop 3` behind the store."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def test_no_wide_store_with_a_register_soffset_is_followed_by_a_write_of_its_data_registers(tmp_path):
    import __graft_entry__ as ge

    ge.build()
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_store_hazard as chk

    csrc = os.path.join(ROOT, "ivln-ce_amd", "csrc")
    sites_total = 0
    for f in sorted(os.listdir(csrc)):
        if f.endswith(".o") and os.path.exists(os.path.join(csrc, f[:-2] + ".hip")):
            sites, bad = chk.check(chk.disassemble(os.path.join(csrc, f)), need=2)
            assert not bad, (f, bad[:3])
            sites_total += sites
    assert sites_total >= 20  # (the straight-from-the-accumulators epilogues of conv_bf3.hip)
    # the checker is not blind: with the guard compiled out the same source leaves VALU writes right behind such stores
    noguard = str(tmp_path / "conv_bf3_noguard.o")
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-DBF3_NO_STORE_GUARD", "-c",
                           os.path.join(csrc, "conv_bf3.hip"), "-o", noguard])
    sites, bad = chk.check(chk.disassemble(noguard), need=2)
    assert sites >= 20 and len(bad) >= 1


@pytest.mark.gpu
def test_guarded_wide_store_leaves_no_stale_lanes_on_the_hardware():
    exe = os.path.join(ROOT, "tools", "store_hazard")
    src = exe + ".hip"
    if not os.path.exists(exe) or os.path.getmtime(exe) < os.path.getmtime(src):
        subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-o", exe, src])
    out = subprocess.run([exe, "64"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    rows = {}
    for ln in out.stdout.splitlines():
        m = re.search(r"^(.*?)\s+workgroups\s+(\d+).*?stale stores\s+(\d+).*?other mismatches (\d+)", ln)
        if m:
            rows[m.group(1).strip()] = (int(m.group(2)), int(m.group(3)), int(m.group(4)))
    guard = next(v for k, v in rows.items() if "BF3_STORE_GUARD" in k)
    assert guard[0] >= 250_000 and guard[1] == 0 and guard[2] == 0, rows
    assert all(v[2] == 0 for v in rows.values()), rows  # (nothing but the A-or-B question is ever wrong)
