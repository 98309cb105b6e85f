"""Import shim: the package directory is named ``foveated-instance-segmentation_amd`` (not a
valid Python identifier), so ``import fovealseg`` loads it under that alias."""
import importlib.util
import os
import sys

_PKG_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "foveated-instance-segmentation_amd")
_spec = importlib.util.spec_from_file_location(
    "fovealseg", os.path.join(_PKG_DIR, "__init__.py"), submodule_search_locations=[_PKG_DIR])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["fovealseg"] = _mod
_spec.loader.exec_module(_mod)
