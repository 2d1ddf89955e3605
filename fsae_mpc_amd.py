"""Import shim: the product package lives in the directory `fsae-mpc_amd/` (not a valid Python
identifier); `import fsae_mpc_amd` loads it from there."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fsae-mpc_amd")
_spec = importlib.util.spec_from_file_location("fsae_mpc_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["fsae_mpc_amd"] = _mod
_spec.loader.exec_module(_mod)
