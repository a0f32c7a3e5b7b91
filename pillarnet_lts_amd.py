"""Import shim: the package directory is `pillarnet-lts_amd/` (not a valid
Python identifier), so `import pillarnet_lts_amd` loads it from there."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "pillarnet-lts_amd")
_spec = importlib.util.spec_from_file_location(
    "pillarnet_lts_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir]
)
_mod = importlib.util.module_from_spec(_spec)
sys.modules["pillarnet_lts_amd"] = _mod
_spec.loader.exec_module(_mod)
