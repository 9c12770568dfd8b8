"""Importable alias for the package directory
`physics-informed-neural-network-for-explainable-fault-diagnosis-in-fuel-cells_amd/`
(its mandated name contains hyphens, which `import` cannot spell).

`import pinn_amd` loads that directory as the package `pinn_amd`.
"""
import importlib.util
import os
import sys

_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)),
                    "physics-informed-neural-network-for-explainable-fault-diagnosis-in-fuel-cells_amd")
_spec = importlib.util.spec_from_file_location(
    "pinn_amd", os.path.join(_DIR, "__init__.py"), submodule_search_locations=[_DIR])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["pinn_amd"] = _mod
_spec.loader.exec_module(_mod)
