"""Import shim: the package directory is named ``saspa-aug_amd`` (not a valid Python
identifier), so ``import saspa_aug_amd`` resolves to this file, which loads the real
package from that directory under the importable name and replaces itself in
``sys.modules``."""
import importlib.util
import os
import sys

_root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "saspa-aug_amd")
_spec = importlib.util.spec_from_file_location(
    "saspa_aug_amd", os.path.join(_root, "__init__.py"), submodule_search_locations=[_root])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["saspa_aug_amd"] = _mod
_spec.loader.exec_module(_mod)
