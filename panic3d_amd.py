"""Import shim: `import panic3d_amd` loads the package in ./panic3d-anime-reconstruction_amd/ (a hyphen cannot be imported)."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "panic3d-anime-reconstruction_amd")
_spec = importlib.util.spec_from_file_location("panic3d_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["panic3d_amd"] = _mod
_spec.loader.exec_module(_mod)
