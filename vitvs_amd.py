"""Import shim: the package directory is named ``vit-vs_amd`` (not a valid Python
identifier), so ``import vitvs_amd`` loads it from there under this name."""
import importlib.util
import os
import sys

_here = os.path.dirname(os.path.abspath(__file__))
_pkg_dir = os.path.join(_here, "vit-vs_amd")
_spec = importlib.util.spec_from_file_location(
    "vitvs_amd", os.path.join(_pkg_dir, "__init__.py"),
    submodule_search_locations=[_pkg_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["vitvs_amd"] = _mod
_spec.loader.exec_module(_mod)
