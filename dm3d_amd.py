"""Import shim: the package directory is named after the reference repository
(``3d-condtional-stable-diffusion_amd``), which is not a Python identifier; ``import dm3d_amd`` loads it."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "3d-condtional-stable-diffusion_amd")
_spec = importlib.util.spec_from_file_location("dm3d_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["dm3d_amd"] = _mod
_spec.loader.exec_module(_mod)
