"""CPU: the C-ABI library builds for gfx950, loads without a GPU and exports every symbol that
include/pinn_hip.h declares; host-only entry points behave (no compute calls here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    g.build()
    from pinn_amd import _lib
    return _lib.load(build_if_missing=False)


def _header_functions():
    src = open(os.path.join(ROOT, "include", "pinn_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(pinn_[a-z_0-9]+)\s*\(", src)))


def test_header_symbols_exported_and_bound(lib):
    from pinn_amd import _lib
    names = _header_functions()
    assert len(names) >= 10
    for n in names:
        assert hasattr(lib, n), "library does not export %s" % n
    assert sorted(_lib.declared_symbols()) == names, "ctypes signatures out of sync with include/pinn_hip.h"


def test_host_only_entry_points(lib):
    from pinn_amd import _lib, layout
    assert lib.pinn_abi_version() == 2
    assert lib.pinn_residuals_workspace_bytes() == 1024 * 32 * 8
    for H, nh in ((256, 3), (128, 3), (256, 1), (128, 8)):
        net = _lib.Net(8, H, nh)
        offs, total = layout.param_offsets(8, H, nh)
        assert lib.pinn_param_count(ctypes.byref(net)) == total
        assert lib.pinn_train_workspace_bytes(ctypes.byref(net), 1000) > 0
    assert layout.param_offsets(8, 256, 3)[1] == 175362 + 6          # 175 362 parameters + two 3-float alignment pads
    assert lib.pinn_packed_bytes(ctypes.byref(_lib.Net(8, 256, 3, _lib.PREC_BF16))) == 688128
    assert lib.pinn_packed_bytes(ctypes.byref(_lib.Net(8, 256, 3))) == 0
    assert lib.pinn_train_workspace_bytes(ctypes.byref(_lib.Net(8, 256, 3, _lib.PREC_BF16, None)), 1000) == 0      # bf16 needs its scratch buffer
    # wide nets (layer-by-layer kernels): sized by shape alone; every precision but exact fp32 (which needs no scratch anywhere)
    assert lib.pinn_param_count(ctypes.byref(_lib.Net(8, 1024, 4))) == layout.param_offsets(8, 1024, 4)[1]
    wide_bytes = lib.pinn_packed_bytes(ctypes.byref(_lib.Net(8, 1024, 4, _lib.PREC_F32X6)))
    assert wide_bytes > 5 * 2 * 4 * 1024 * 1024                 # five 16-bit copies of the matrices + the activation scratch
    assert lib.pinn_packed_bytes(ctypes.byref(_lib.Net(8, 1024, 4, _lib.PREC_BF16))) == wide_bytes
    assert lib.pinn_packed_bytes(ctypes.byref(_lib.Net(8, 1024, 4, _lib.PREC_FP32))) == 0
    assert lib.pinn_param_count(ctypes.byref(_lib.Net(8, 384, 2))) < 0
    for bad in (_lib.Net(8, 96, 3), _lib.Net(7, 256, 3), _lib.Net(8, 256, 0), _lib.Net(8, 384, 3), _lib.Net(8, 256, 9)):
        assert lib.pinn_param_count(ctypes.byref(bad)) == -2
        assert lib.pinn_train_workspace_bytes(ctypes.byref(bad), 1000) == 0
    # struct layouts agree with the C side (sizes from the header's field lists)
    assert ctypes.sizeof(_lib.Affine) == 8 * 8 * 2 + 8 * 2 + 4 * 2
    assert ctypes.sizeof(_lib.Net) == 24          # 4 ints + 4 pad + device pointer
    assert ctypes.sizeof(_lib.Dropout) == 4 + 36 + 8 + 4 + 4 + 8 + 8 + 8   # incl. padding before row_offset


def test_null_workspace_and_buffer_pointers_are_argument_errors(lib):
    """Every entry point that takes a workspace / stash / output pointer returns PINN_E_ARG (-1) for NULL before it touches
    the GPU (the checks run on the host, so this needs no device), and PINN_E_WORKSPACE (-3) for a workspace that is too
    small: a kernel is never handed a pointer the entry point has not seen to be non-NULL and large enough.  (Round 2's
    scratch ablation builds faulted on address nil with a stash pointer their own edit had nulled, gpurun_out/ab2.txt.)"""
    from pinn_amd import _lib
    aff = _lib.Affine()
    one = ctypes.c_void_p(0x1000)          # a non-NULL address that is never dereferenced: every call below fails on the host
    E_ARG, E_WS = -1, -3
    # residual pass: sums requested without a workspace; workspace too small
    assert lib.pinn_residuals(one, one, one, ctypes.byref(aff), one, _lib.RES_ALL, 10, None, 0, one, None, 0, None) == E_ARG
    assert lib.pinn_residuals(one, one, one, ctypes.byref(aff), one, _lib.RES_ALL, 10, None, 0, one, one, 16, None) == E_WS
    assert lib.pinn_residuals(None, one, one, ctypes.byref(aff), one, _lib.RES_ALL, 10, None, 0, None, None, 0, None) == E_ARG
    assert lib.pinn_residuals_cached(one, ctypes.byref(aff), one, _lib.RES_T, 10, one, None, 0, None) == E_ARG
    assert lib.pinn_residuals_cached(one, ctypes.byref(aff), one, _lib.RES_T, 10, one, one, 16, None) == E_WS
    assert lib.pinn_residuals_cached(None, ctypes.byref(aff), one, _lib.RES_T, 10, one, one, 1 << 20, None) == E_ARG
    assert lib.pinn_residuals_prepare(one, None, None, ctypes.byref(aff), one, _lib.RES_T, 10, None, None) == E_ARG
    assert lib.pinn_lambda_stage_run(_lib.STAGE_THERMAL, _lib.RES_T, one, None, None, ctypes.byref(aff), 10, 1.0, 0.8, 1000, 0, 5, one, one, one,
                                     None, 1000, None, None, 0, None) == E_ARG
    assert lib.pinn_lambda_stage_run(_lib.STAGE_THERMAL, _lib.RES_T, one, None, None, ctypes.byref(aff), 10, 1.0, 0.8, 1000, 0, 5, one, one, one,
                                     None, 1000, None, one, 8, None) == E_WS
    assert lib.pinn_lambda_step(_lib.STAGE_THERMAL, None, 10, 1.0, 0.1, 1, one, one, one, None) == E_ARG
    assert lib.pinn_net_f_t(one, one, None, None, ctypes.byref(aff), one, 10, None, one, one, None) == E_ARG
    # network entry points: NULL parameter / row / output / workspace / packed-weight pointers
    net = _lib.Net(8, 256, 3)
    assert lib.pinn_mlp_forward(ctypes.byref(net), one, one, 10, None, None, one, None) == E_ARG
    assert lib.pinn_mlp_forward(ctypes.byref(net), one, None, 10, None, one, one, None) == E_ARG
    d = _lib.Dropout(); d.mode = _lib.DROP_PHILOX
    assert lib.pinn_mc_dropout(ctypes.byref(net), one, one, 10, ctypes.byref(d), 4, one, None, one, None) == E_ARG
    assert lib.pinn_mlp_train_grads(ctypes.byref(net), one, one, one, 10, 10, None, one, one, None, 1 << 30, None) == E_ARG
    assert lib.pinn_mlp_train_grads(ctypes.byref(net), one, one, one, 10, 10, None, ctypes.c_void_p(0x1000), one, ctypes.c_void_p(0x2000), 16, None) == E_WS
    assert lib.pinn_mlp_train_grads(ctypes.byref(net), one, one, one, 10, 10, None, ctypes.c_void_p(0x1004), one, ctypes.c_void_p(0x2000), 1 << 30, None) == E_ARG   # alignment
    for prec in (_lib.PREC_BF16, _lib.PREC_F32X6, _lib.PREC_F32X6_G6):
        split = _lib.Net(8, 256, 3, prec, None)           # a split-operand precision without its packed-weight scratch
        assert lib.pinn_mlp_forward(ctypes.byref(split), one, one, 10, None, one, one, None) == E_ARG
        assert lib.pinn_mlp_train_grads(ctypes.byref(split), one, one, one, 10, 10, None, one, one, one, 1 << 30, None) == E_ARG
    d.mode = _lib.DROP_BITS            # injected masks without the mask buffer
    assert lib.pinn_mlp_forward(ctypes.byref(net), one, one, 10, ctypes.byref(d), one, one, None) == E_ARG
    assert lib.pinn_adam_step(one, one, None, one, 10, 0.01, 1, None) == E_ARG
    # gradients + Adam as one launch sequence: NULL moments / coefficient table / step counter, a step count below 1, and the
    # device-counter form on a precision whose kernels do not leave the counter's snapshot
    al = ctypes.c_void_p(0x1000)
    assert lib.pinn_mlp_train_step(ctypes.byref(net), al, one, one, 10, 10, None, al, one, ctypes.c_void_p(0x2000), 1 << 30, None, al, 0.01, 1, None) == E_ARG
    assert lib.pinn_mlp_train_step(ctypes.byref(net), al, one, one, 10, 10, None, al, one, ctypes.c_void_p(0x2000), 1 << 30, al, al, 0.01, 0, None) == E_ARG
    assert lib.pinn_mlp_train_step(ctypes.byref(net), al, one, one, 10, 10, None, al, one, None, 1 << 30, al, al, 0.01, 1, None) == E_ARG
    dc = _lib.Dropout(); dc.mode = _lib.DROP_PHILOX
    assert lib.pinn_mlp_train_step_dev(ctypes.byref(net), al, one, one, 10, 10, ctypes.byref(dc), al, one, ctypes.c_void_p(0x2000), 1 << 30, al, al, al, None) == E_ARG  # no counter
    dc.d_step_counter = 0x3000
    assert lib.pinn_mlp_train_step_dev(ctypes.byref(net), al, one, one, 10, 10, ctypes.byref(dc), al, one, ctypes.c_void_p(0x2000), 1 << 30, al, al, None, None) == E_ARG
    assert lib.pinn_mlp_train_step_dev(ctypes.byref(net), al, one, one, 10, 10, ctypes.byref(dc), al, one, ctypes.c_void_p(0x2000), 1 << 30, al, al, al, None) == -2     # PINN_E_ARCH: exact fp32
    assert lib.pinn_results_assemble(one, one, ctypes.byref(aff), 0.0, 1.0, 200, None, 0, one, one, one, one, 10, None, 10, None, None) == E_ARG


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "physics-informed-neural-network-for-explainable-fault-diagnosis-in-fuel-cells_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "pinn_oracle" not in txt and "import oracle" not in txt, f


def test_model_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import pinn_amd
    from pinn_amd import synth
    ds = synth.make_dataset(64, (), seed=0)
    with pytest.raises(Exception) as e:
        pinn_amd.PhysicsInformedNN(ds[0], ds[1], [8, 256, 256, 256, 1], ds[4], ds[5], p=0.2, logvar=True)
    assert "GPU" in str(e.value)
