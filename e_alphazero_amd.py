"""Import alias: the package directory is `e-alphazero_amd/` (not a valid identifier), so this module loads it under
the importable name `e_alphazero_amd` and replaces itself in sys.modules."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "e-alphazero_amd")
_spec = importlib.util.spec_from_file_location(
    "e_alphazero_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir]
)
_mod = importlib.util.module_from_spec(_spec)
sys.modules["e_alphazero_amd"] = _mod
_spec.loader.exec_module(_mod)
