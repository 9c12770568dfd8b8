"""GPU parity of pinn_results_assemble (the [N,22] comprehensive_results fill, 01:1877-2010) through the C ABI, against the
oracle's float64 host arithmetic: sklearn-style float32 de-normalisation, pandas even-window centred moving average inside
each segment, column order.  (The end-to-end array against the reference's own output: test_gpu_model.py, g_results.npz.)"""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import pinn_oracle as O


def _assemble(n, window, seg_end, seed=0, with_labels=True):
    import hip_helpers as hh
    from pinn_amd import _lib, synth
    lib = _lib.load()
    rng = np.random.default_rng(seed)
    ds = synth.make_dataset(max(n, 8), (), seed=seed, as_torch=False)
    sx, sy = ds[4], ds[5]
    x = rng.uniform(-1, 1, size=(n, 8)).astype(np.float32)
    y = rng.uniform(-1, 1, size=n).astype(np.float32)
    pm = rng.normal(size=n).astype(np.float32)
    au = rng.uniform(0.01, 1, size=n).astype(np.float32)
    eu = rng.uniform(0.001, 0.1, size=n).astype(np.float32)
    cols = rng.normal(size=(_lib.NCOLS, n)).astype(np.float32)
    labels = rng.integers(0, 5, size=n).astype(np.float32) if with_labels else None
    aff = hh.affine_struct(sx, sy)
    mc_scale = 2.0 / (float(sy.data_max_[0]) - float(sy.data_min_[0]) + 1e-12)
    mc_min = -1.0 - float(sy.data_min_[0]) * mc_scale
    d = lambda a: torch.from_numpy(a).to(hh.dev()) if a is not None else None
    dx, dy, dpm, dau, deu, dcols, dlab = d(x), d(y), d(pm), d(au), d(eu), d(cols), d(labels)
    dseg = torch.tensor(seg_end, dtype=torch.int64, device=hh.dev()) if seg_end else None
    out = torch.full((n, 22), float("nan"), dtype=torch.float64, device=hh.dev())
    rc = lib.pinn_results_assemble(hh.ptr(dx), hh.ptr(dy), ctypes.byref(aff), mc_min, mc_scale, window, hh.ptr(dseg),
                                   len(seg_end) if seg_end else 0, hh.ptr(dpm), hh.ptr(dau), hh.ptr(deu), hh.ptr(dcols), n,
                                   hh.ptr(dlab), n, hh.ptr(out), hh.stream())
    _lib.check(rc, "pinn_results_assemble")
    got = out.cpu().numpy()
    # float64 host restatement
    want = np.zeros((n, 22))
    want[:, 0:8] = O.denorm(x, *O.scaler_affine(sx))
    want[:, 8] = O.denorm(y.reshape(-1, 1), *O.scaler_affine(sy)).reshape(-1)
    div = mc_scale + 1e-12
    want[:, 9] = (pm.astype(np.float64) - mc_min) / div
    smooth = (lambda v: O.smooth_by_segments(v, list(seg_end), window)) if seg_end else (lambda v: O.moving_average_centered(v, window))
    want[:, 10] = smooth(au.astype(np.float64) / div)
    want[:, 11] = smooth(eu.astype(np.float64) / div)
    want[:, 12] = want[:, 8] - want[:, 9]
    for c, name in ((13, "FV"), (14, "FT"), (15, "FH"), (16, "FO"), (18, "VEST5"), (19, "TPRED"), (20, "ACTH"), (21, "ACTO")):
        want[:, c] = cols[_lib.C[name]]
    want[:, 17] = labels if labels is not None else 0.0
    return got, want


@pytest.mark.parametrize("n,window,seg_end", [
    (700, 200, [300, 450, 700]),        # the golden layout: window > the 150-row segment
    (700, 200, None),                   # no segments
    (1, 200, None), (5, 4, [2, 5]), (257, 7, [100, 257]), (1000, 1, [1000]), (3000, 1024, [1, 2, 1500, 3000]),
    (20001, 201, [7, 10000, 10001, 20001]),
])
def test_results_assemble_matches_host_arithmetic(n, window, seg_end):
    got, want = _assemble(n, window, seg_end, seed=n % 7)
    exact = [c for c in range(22) if c not in (10, 11)]
    assert np.array_equal(got[:, exact], want[:, exact])                 # same float64 operations, bit for bit
    np.testing.assert_allclose(got[:, 10:12], want[:, 10:12], rtol=1e-12, atol=0)        # window sums: order of summation only


def test_results_assemble_large_and_errors():
    import hip_helpers as hh
    from pinn_amd import _lib
    n = 1_000_000
    got, want = _assemble(n, 200, [400000, 400100, n], seed=3, with_labels=False)
    assert np.array_equal(got[:, 17], np.zeros(n)) and np.array_equal(got[:, 9], want[:, 9])
    np.testing.assert_allclose(got[:, 10:12], want[:, 10:12], rtol=1e-11, atol=0)
    # smoothing a constant is the identity whatever the segments: a size-independent check of the window bookkeeping
    lib = _lib.load()
    aff = _lib.Affine()
    for c in range(8):
        aff.x_scale[c] = 1.0
    aff.y_scale = 1.0
    z = torch.zeros(n, 8, device=hh.dev())
    one = torch.full((n,), 0.25, device=hh.dev())
    cols = torch.zeros(_lib.NCOLS, n, device=hh.dev())
    seg = torch.tensor([17, 123456, 123457, 999999, n], dtype=torch.int64, device=hh.dev())
    out = torch.empty(n, 22, dtype=torch.float64, device=hh.dev())
    args = lambda window, nseg: (hh.ptr(z), hh.ptr(one), ctypes.byref(aff), 0.0, 1.0 - 1e-12, window, hh.ptr(seg), nseg, hh.ptr(one),
                                 hh.ptr(one), hh.ptr(one), hh.ptr(cols), n, None, n, hh.ptr(out), hh.stream())
    assert lib.pinn_results_assemble(*args(200, 5)) == 0
    o = out.cpu().numpy()
    assert np.all(o[:, 10] == 0.25) and np.all(o[:, 11] == 0.25) and np.all(o[:, 12] == 0.0)
    assert lib.pinn_results_assemble(*args(0, 5)) == -1
    assert lib.pinn_results_assemble(*args(2000, 5)) == -1
    assert lib.pinn_results_assemble(*args(200, -1)) == -1
