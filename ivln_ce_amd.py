"""Import alias: the package directory is `ivln-ce_amd/` (not a valid Python identifier), so
`import ivln_ce_amd` loads it from there."""
import importlib.util
import os
import sys

_d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ivln-ce_amd")
_spec = importlib.util.spec_from_file_location(
    "ivln_ce_amd", os.path.join(_d, "__init__.py"), submodule_search_locations=[_d]
)
_mod = importlib.util.module_from_spec(_spec)
sys.modules["ivln_ce_amd"] = _mod
_spec.loader.exec_module(_mod)
