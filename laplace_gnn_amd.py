"""Import shim: the package lives in the directory ``laplace-gnn_amd/`` (the name the build
contract prescribes); a hyphen is not importable, so ``import laplace_gnn_amd`` loads that
directory under this module name."""
import importlib.util
import os
import sys

_here = os.path.dirname(os.path.abspath(__file__))
_pkg_dir = os.path.join(_here, "laplace-gnn_amd")
_spec = importlib.util.spec_from_file_location(
    "laplace_gnn_amd", os.path.join(_pkg_dir, "__init__.py"), submodule_search_locations=[_pkg_dir]
)
_mod = importlib.util.module_from_spec(_spec)
sys.modules["laplace_gnn_amd"] = _mod
_spec.loader.exec_module(_mod)
