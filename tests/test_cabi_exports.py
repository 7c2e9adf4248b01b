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
