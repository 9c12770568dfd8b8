"""CPU: host-side logic of the package (layout, synthetic data, scaler, results assembly helpers)."""
import numpy as np
import pytest

import pinn_oracle as O


def test_layout_matches_reference_state_dict_order():
    from pinn_amd import layout
    offs, total = layout.param_offsets(8, 256, 3)
    assert [n for n, _, _ in offs] == O.param_names(3)
    assert all(off % 4 == 0 for _, _, off in offs) and total % 4 == 0
    sizes = [int(np.prod(s)) for _, s, _ in offs]
    assert sum(sizes) == 175362                      # P = 175 362
    for (_, s, off), nxt in zip(offs[:-1], offs[1:]):
        assert nxt[2] >= off + int(np.prod(s))
    with pytest.raises(ValueError):
        layout.check_arch([8, 100, 100, 1])
    with pytest.raises(ValueError):
        layout.check_arch([8, 256, 128, 1])
    with pytest.raises(ValueError):
        layout.check_arch([9, 256, 256, 1])
    assert layout.check_arch([8, 256, 256, 256, 1]) == (8, 256, 3)


def test_minmax_affine_matches_sklearn():
    from sklearn.preprocessing import MinMaxScaler
    from pinn_amd import synth
    X, U = synth.synth_rows(500, seed=3)
    a, b = synth.MinMaxAffine((-1, 1)).fit(X), MinMaxScaler(feature_range=(-1, 1)).fit(X)
    for attr in ("min_", "scale_", "data_min_", "data_max_"):
        assert np.array_equal(getattr(a, attr), getattr(b, attr))
    xn = a.transform(X)
    assert np.array_equal(xn, b.transform(X))
    x32 = xn.astype(np.float32)
    assert np.array_equal(a.inverse_transform(x32), b.inverse_transform(x32))
    assert a.inverse_transform(x32).dtype == np.float32
    assert np.array_equal(O.denorm(x32, a.min_, a.scale_), b.inverse_transform(x32))


def test_synthetic_dataset_contract():
    from pinn_amd import synth
    ds = synth.make_dataset(300, (150, 250), seed=3)
    x_train, y_train, x_test, y_test, sx, sy, info = ds
    assert x_train.shape == (300, 8) and y_train.shape == (300, 1) and x_test.shape == (700, 8)
    assert info["boundary_lines"] == [300, 450, 700] and len(info["fault_data_list"]) == 2
    assert float(x_train.min()) >= -1.0 - 1e-6 and float(x_train.max()) <= 1.0 + 1e-6
    # valid physical domain: i < lambda_3's lower clamp (2.0) so V_conc stays finite
    X = sx.inverse_transform(x_test.numpy().astype(np.float64))
    assert X[:, 0].max() / 270 < 2.0 and X[:, 0].min() > 50


def test_results_helpers_match_oracle_and_pandas():
    import pandas as pd
    from pinn_amd import results
    rng = np.random.default_rng(1)
    vals = rng.normal(size=700)
    for w in (200, 7, 1):
        want = pd.Series(vals).rolling(window=w, center=True, min_periods=1).mean().values
        np.testing.assert_allclose(results._moving_average_centered(vals, w), want, rtol=1e-10, atol=1e-12)
    a = results.smooth_by_segments(vals, [300, 450, 700], 200)
    b = O.smooth_by_segments(vals, [300, 450, 700], 200)
    np.testing.assert_allclose(a, b, rtol=1e-10, atol=1e-12)
    assert abs(a[300:450].std()) < abs(vals[300:450].std())        # a 150-row segment under a 200 window is heavily smoothed
    # boundary not ending at n: falls back to whole-array smoothing (01:1859-1862)
    np.testing.assert_allclose(results.smooth_by_segments(vals, [300], 200), results._moving_average_centered(vals, 200))
    info = {"boundary_lines": [300, 450, 700], "fault_data_list": [(None, None, "a"), (None, None, "b")]}
    lab = results.create_fault_labels(700, info)
    assert np.array_equal(lab, O.fault_labels(700, info)) and set(lab) == {0.0, 1.0, 2.0}


def test_dp_shard_bounds_cover_rows():
    from pinn_amd import dp
    for n in (10, 1000003, 7):
        for g in (1, 2, 3, 8):
            b = [dp.shard_bounds(n, r, g) for r in range(g)]
            assert b[0][0] == 0 and b[-1][1] == n and all(b[i][1] == b[i + 1][0] for i in range(g - 1))


def test_check_arch_widths():
    """Fused widths, wide widths (layer-by-layer kernels) and rejected ones."""
    from pinn_amd import layout
    assert layout.check_arch([8, 256, 256, 256, 1]) == (8, 256, 3)
    assert layout.check_arch([8, 1024, 1024, 1024, 1024, 1]) == (8, 1024, 4)
    for bad in ([8, 384, 1], [8, 256, 128, 1], [7, 256, 1], [8, 256, 2], [8] + [256] * 9 + [1]):
        with pytest.raises(ValueError):
            layout.check_arch(bad)
