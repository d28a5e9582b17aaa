"""Importable alias of the package directory `pde-based-image-processing_amd/` (whose name is
not a Python identifier):  `import pdeip_amd; pdeip_amd.mex_api.Oflow_sor_elin4_2d(...)`."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_pkg = importlib.import_module("pde-based-image-processing_amd")
sys.modules[__name__] = _pkg
