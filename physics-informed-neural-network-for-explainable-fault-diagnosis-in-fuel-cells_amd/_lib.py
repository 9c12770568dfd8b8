"""ctypes binding of libpinn_hip.so -- the only route from Python to the HIP kernels.

There is NO CPU fallback: if the library is missing or a call fails, an exception is raised.
All pointers handed over are raw device pointers (torch tensors' data_ptr()); torch only
provides device memory and the current stream.
"""
import ctypes
import os

from . import _build

c_void_p, c_int, c_uint, c_ll, c_float, c_size_t = (ctypes.c_void_p, ctypes.c_int, ctypes.c_uint, ctypes.c_longlong,
                                                    ctypes.c_float, ctypes.c_size_t)

NLAMBDA, NCOLS, NSUMS = 17, 20, 32
RES_V, RES_T, RES_H, RES_O, RES_ALL = 1, 2, 4, 8, 15
DROP_NONE, DROP_PHILOX, DROP_BITS = 0, 1, 2
STAGE_LAMBDA_PM, STAGE_LAMBDA_F, STAGE_THERMAL, STAGE_HYDROGEN, STAGE_OXYGEN = 0, 1, 2, 3, 4

# column / sum indices (include/pinn_hip.h)
C = {n: i for i, n in enumerate(
    ["FV", "VACT", "VOHM", "VCONC", "ENERNST", "VEST5", "I", "VOUT5", "FT", "TPRED", "TOUT",
     "FH", "ACTH", "TGTH", "ITOT", "FO", "ACTO", "TGTO", "QO2", "O2FLOW"])}
S = {n: i for i, n in enumerate(
    ["FV2", "FV_D1", "FV_D2", "FV_D3", "YV2", "YV_D1", "YV_D2", "YV_D3", "YU2",
     "FT2", "FT_D1", "FT_D3", "FT_D5", "FT_ABS", "FH2", "FH_D1", "FH_D2", "FH_D3", "ACTH", "TGTH",
     "FO2", "FO_D1", "FO_D2", "FO_D3", "ACTO", "TGTO"])}


class Affine(ctypes.Structure):
    _fields_ = [("x_min", ctypes.c_double * 8), ("x_scale", ctypes.c_double * 8), ("y_min", ctypes.c_double),
                ("y_scale", ctypes.c_double), ("vn_scale", c_float), ("vn_min", c_float)]


PREC_FP32, PREC_BF16, PREC_F32X6, PREC_F32X6_G6 = 0, 1, 2, 3
PHASE_CHAIN, PHASE_WGRAD, PHASE_REDUCE, PHASE_ALL = 1, 2, 4, 7
PHASE_WGRAD_TAIL, PHASE_WGRAD_HEAD, PHASE_REDUCE_TAIL, PHASE_REDUCE_HEAD = 32, 64, 128, 256
STAGE_RUN_MAX_ROWS, STAGE_LOG_FLOATS = 65536, 64


class Net(ctypes.Structure):
    _fields_ = [("n_in", c_int), ("hidden", c_int), ("n_hidden", c_int), ("precision", c_int), ("d_packed", c_void_p)]

    def __init__(self, n_in=8, hidden=256, n_hidden=3, precision=0, d_packed=None):
        super().__init__(n_in, hidden, n_hidden, precision, d_packed)


class Dropout(ctypes.Structure):
    _fields_ = [("mode", c_int), ("p", c_float * 9), ("seed", ctypes.c_ulonglong), ("stream", c_uint),
                ("row_offset", c_ll), ("d_bits", c_void_p), ("d_step_counter", c_void_p)]


class PinnError(RuntimeError):
    pass


class PinnRangeError(PinnError):
    """A weight or gradient left the domain of the split-operand precisions (include/pinn_hip.h: PINN_E_RANGE)."""


E_RANGE = -4


_SIGS = {
    "pinn_abi_version": (c_int, []),
    "pinn_residuals_workspace_bytes": (c_size_t, []),
    "pinn_residuals": (c_int, [c_void_p, c_void_p, c_void_p, ctypes.POINTER(Affine), c_void_p, c_uint, c_ll, c_void_p, c_ll,
                               c_void_p, c_void_p, c_size_t, c_void_p]),
    "pinn_lambda_step": (c_int, [c_int, c_void_p, c_ll, c_float, c_float, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "pinn_lambda_stage_run": (c_int, [c_int, ctypes.c_uint, c_void_p, c_void_p, c_void_p, ctypes.POINTER(Affine), c_ll, ctypes.c_double,
                                      ctypes.c_double, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "pinn_lambda_stage_workspace_bytes": (c_size_t, [c_ll]),
    "pinn_param_count": (c_ll, [ctypes.POINTER(Net)]),
    "pinn_packed_bytes": (c_size_t, [ctypes.POINTER(Net)]),
    "pinn_net_range_status": (c_int, [ctypes.POINTER(Net), c_void_p]),
    "pinn_mlp_forward": (c_int, [ctypes.POINTER(Net), c_void_p, c_void_p, c_ll, ctypes.POINTER(Dropout), c_void_p, c_void_p,
                                 c_void_p]),
    "pinn_mc_dropout": (c_int, [ctypes.POINTER(Net), c_void_p, c_void_p, c_ll, ctypes.POINTER(Dropout), c_int, c_void_p,
                                c_void_p, c_void_p, c_void_p]),
    "pinn_train_workspace_bytes": (c_size_t, [ctypes.POINTER(Net), c_ll]),
    "pinn_grad_split": (c_ll, [ctypes.POINTER(Net)]),
    "pinn_mlp_train_grads": (c_int, [ctypes.POINTER(Net), c_void_p, c_void_p, c_void_p, c_ll, c_ll, ctypes.POINTER(Dropout),
                                     c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "pinn_mlp_train_grads_phases": (c_int, [ctypes.POINTER(Net), c_void_p, c_void_p, c_void_p, c_ll, c_ll, ctypes.POINTER(Dropout),
                                            c_void_p, c_void_p, c_void_p, c_size_t, c_void_p, c_uint]),
    "pinn_adam_step": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_ll, c_float, c_int, c_void_p]),
    "pinn_adam_coeffs": (None, [c_float, c_int, ctypes.POINTER(c_float), ctypes.POINTER(c_float)]),
    "pinn_adam_step_dev": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_ll, c_void_p, c_void_p, c_void_p]),
    "pinn_mlp_train_step_dev": (c_int, [ctypes.POINTER(Net), c_void_p, c_void_p, c_void_p, c_ll, c_ll, ctypes.POINTER(Dropout),
                                        c_void_p, c_void_p, c_void_p, c_size_t, c_void_p, c_void_p, c_void_p, c_void_p]),
    "pinn_mlp_train_step": (c_int, [ctypes.POINTER(Net), c_void_p, c_void_p, c_void_p, c_ll, c_ll, ctypes.POINTER(Dropout),
                                    c_void_p, c_void_p, c_void_p, c_size_t, c_void_p, c_void_p, c_float, c_int, c_void_p]),
    "pinn_residuals_prepare": (c_int, [c_void_p, c_void_p, c_void_p, ctypes.POINTER(Affine), c_void_p, c_uint, c_ll, c_void_p, c_void_p]),
    "pinn_residuals_cached": (c_int, [c_void_p, ctypes.POINTER(Affine), c_void_p, c_uint, c_ll, c_void_p, c_void_p, c_size_t, c_void_p]),
    "pinn_net_f_t": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, ctypes.POINTER(Affine), c_void_p, c_ll, c_void_p, c_void_p, c_void_p,
                             c_void_p]),
    "pinn_results_assemble": (c_int, [c_void_p, c_void_p, ctypes.POINTER(Affine), ctypes.c_double, ctypes.c_double, c_int, c_void_p, c_int,
                                      c_void_p, c_void_p, c_void_p, c_void_p, c_ll, c_void_p, c_ll, c_void_p, c_void_p]),
}

_lib = None


def lib_path():
    return _build.LIB


def load(build_if_missing=True):
    """Load (building first if needed and possible) the HIP library. Raises if unavailable."""
    global _lib
    if _lib is not None:
        return _lib
    # torch must be imported BEFORE the library: torch bundles its own HIP runtime (libamdhip64.so.7);
    # loading ours first would pull /opt/rocm's copy as a second runtime, and streams / device
    # pointers from torch are meaningless to a different runtime instance (hipErrorNoDevice).
    import torch  # noqa: F401
    path = _build.LIB
    override = os.environ.get("PINN_HIP_LIB")      # experiments: load another build of the same ABI
    if override:
        path, build_if_missing = override, False
    if build_if_missing:
        try:
            path = _build.build()          # no-op when the library matches the sources (content hashes)
        except _build.NoCompiler:
            # a box without hipcc may only run a library that was built from exactly these sources;
            # a compile or link FAILURE (BuildError) always propagates: never run a stale binary
            if not _build.is_current():
                raise PinnError("libpinn_hip.so is missing or does not match the sources in %s, and hipcc is not "
                                "available to rebuild it" % _build.CSRC)
    if not os.path.exists(path):
        raise PinnError("libpinn_hip.so is missing (%s): build it with __graft_entry__.build()" % path)
    lib = ctypes.CDLL(path)
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)          # AttributeError if the library does not export a declared symbol
        fn.restype, fn.argtypes = res, args
    if lib.pinn_abi_version() != 2:
        raise PinnError("libpinn_hip.so ABI version mismatch")
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        raise PinnError("%s failed with code %d" % (what, rc))


def declared_symbols():
    return list(_SIGS.keys())
