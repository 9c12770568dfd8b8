"""MI355X-native PINN training + MC-dropout hot path (gfx950 HIP kernels behind a C ABI).

Public surface mirrors the reference's `01_train_pinn_multiphysics_model.py`:
`PhysicsInformedNN`, `DNN`, `get_MC_samples`, `create_comprehensive_results_array_v2`,
`create_fault_labels`, `smooth_by_segments`, `_moving_average_centered`; the steps either side of the hot path:
`load_data_normal_raw`, `load_data_fault_raw`, `combine_and_normalize_datasets`, `add_noise_to_combined_data` (ingest),
`plot_model_results_detailed_split` (its statistics; no figure), and `save_checkpoint` / `load_checkpoint`.
Submodules are imported lazily so that `pinn_amd.synth` (numpy only) works without torch/HIP.
"""
import importlib

_LAZY = {
    "DNN": "model", "PhysicsInformedNN": "model",
    "get_MC_samples": "mc",
    "create_comprehensive_results_array_v2": "results", "create_fault_labels": "results",
    "smooth_by_segments": "results", "_moving_average_centered": "results",
    "add_noise_to_combined_data": "ingest", "load_data_normal_raw": "ingest", "load_data_fault_raw": "ingest",
    "combine_and_normalize_datasets": "ingest",
    "model_statistics": "report", "plot_model_results_detailed_split": "report",
    "save_checkpoint": "report", "load_checkpoint": "report",
}


def __getattr__(name):
    if name in _LAZY:
        return getattr(importlib.import_module("." + _LAZY[name], __name__), name)
    try:
        return importlib.import_module("." + name, __name__)
    except ModuleNotFoundError as e:
        raise AttributeError(name) from e
