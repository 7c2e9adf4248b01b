"""CPU: the C-ABI library loads and exports every symbol include/ivln_hip.h declares."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "ivln_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ivln_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as ge

    so = ge.build()
    L = ctypes.CDLL(so)
    names = _declared()
    assert len(names) >= 10
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, f"declared in include/ivln_hip.h but not exported: {missing}"
    L.ivln_strerror.restype = ctypes.c_char_p
    assert L.ivln_strerror(0) == b"ok"
    assert L.ivln_version() >= 1


def test_product_path_never_imports_oracle():
    pkg = os.path.join(ROOT, "ivln-ce_amd")
    bad = []
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dp, f)).read()
                if re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M) or "libmapper_ref" in txt:
                    bad.append(f)
    assert not bad, f"product files reference the oracle: {bad}"


def test_mfma_family_has_one_definition():
    """The family behind `roofline.traffic` / `mfma_busy` / the FLOP hooks: tools/kernel_family.py mirrors
    ivln_family_kernel_names(), and every IVLN_LAUNCH_FAMILY site in csrc/ launches a kernel of that list."""
    import sys

    import __graft_entry__ as ge

    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from kernel_family import FAMILY_KERNELS, base_name, is_family

    L = ctypes.CDLL(ge.build())
    L.ivln_family_kernel_names.restype = ctypes.c_char_p
    assert tuple(L.ivln_family_kernel_names().decode().split(",")) == FAMILY_KERNELS
    csrc = os.path.join(ROOT, "ivln-ce_amd", "csrc")
    seen = set()
    for f in sorted(os.listdir(csrc)):
        if not f.endswith(".hip"):
            continue
        txt = open(os.path.join(csrc, f)).read()
        for m in re.finditer(r"IVLN_LAUNCH_FAMILY\(\s*\(?\s*([A-Za-z_0-9]+)", txt):
            assert m.group(1) in FAMILY_KERNELS, f"{f}: IVLN_LAUNCH_FAMILY({m.group(1)} ...) is not in the family list"
            seen.add(m.group(1))
        for m in re.finditer(r'IVLN_LAUNCH_FAMILY_NAMED\(\s*"([A-Za-z_0-9]+)"', txt):
            assert m.group(1) in FAMILY_KERNELS, f"{f}: IVLN_LAUNCH_FAMILY_NAMED(\"{m.group(1)}\" ...) is not in the family list"
            seen.add(m.group(1))
    assert seen == set(FAMILY_KERNELS), (sorted(seen), FAMILY_KERNELS)
    # rocprofv3's kernel names reduce to the same base names
    assert base_name('void (anonymous namespace)::k_conv1x1_bf3_ks<2, 4, true>(ivln_gemm_desc, int)') == "k_conv1x1_bf3_ks"
    assert base_name('"k_conv_bf3<3, 2, 1, 8, 16, 32, 1, 3, false>(ivln_gemm_desc, unsigned int const*, lo"') == "k_conv_bf3"
    assert is_family("k_depth_net(DnArgs)") and not is_family("k_conv_bf3_pack<3>(float const*)") and not is_family("k_add(float*)")
