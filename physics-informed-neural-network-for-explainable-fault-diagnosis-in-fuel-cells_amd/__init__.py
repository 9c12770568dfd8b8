"""MI355X-native PINN training + MC-dropout hot path (gfx950 HIP kernels behind a C ABI).

Public surface mirrors the reference's `01_train_pinn_multiphysics_model.py`:
`PhysicsInformedNN`, `DNN`, `get_MC_samples`, `create_comprehensive_results_array_v2`,
`create_fault_labels`, `smooth_by_segments`, `_moving_average_centered`.
Submodules are imported lazily so that `pinn_amd.synth` (numpy only) works without torch/HIP.
"""
import importlib

_LAZY = {
    "DNN": "model", "PhysicsInformedNN": "model",
    "get_MC_samples": "mc",
    "create_comprehensive_results_array_v2": "results", "create_fault_labels": "results",
    "smooth_by_segments": "results", "_moving_average_centered": "results",
    "DataParallelTrainer": "dp",
}


def __getattr__(name):
    if name in _LAZY:
        return getattr(importlib.import_module("." + _LAZY[name], __name__), name)
    try:
        return importlib.import_module("." + name, __name__)
    except ModuleNotFoundError as e:
        raise AttributeError(name) from e
