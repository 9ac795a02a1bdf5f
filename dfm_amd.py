"""Import alias: ``import dfm_amd`` == the package in ``depth-from-motion_amd/``
(a hyphenated directory name is not a Python identifier)."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_pkg = importlib.import_module('depth-from-motion_amd')
sys.modules[__name__] = _pkg
