"""CPU: the ingest step before the hot path (pinn_amd.ingest) against outputs of the reference's own
loaders / `combine_and_normalize_datasets` on synthetic recordings (tests/golden/g_ingest.npz, oracle/make_golden_io.py)."""
import os

import numpy as np
import pytest
import scipy.io

from conftest import load_golden

KEYS = ("I", "m_W", "T_W_in", "P_H_in", "P_O_in", "T_W_out", "m_H2", "m_O2", "U")
CASES = {"plain": None,
         "gauss_all": {"noise_type": "gaussian", "noise_level": 0.02, "noise_target": "all"},
         "unif_random": {"noise_type": "uniform", "noise_level": 0.05, "noise_target": "random"}}


def _recordings(g, tmp_path):
    from pinn_amd import ingest
    pn = os.path.join(tmp_path, "normal.mat")
    scipy.io.savemat(pn, {k: g["normal." + k] for k in KEYS})
    Xn, Yn = ingest.load_data_normal_raw(pn, verbose=False)
    faults = []
    for j in range(2):
        pf = os.path.join(tmp_path, "fault%d.mat" % j)
        scipy.io.savemat(pf, {"segment_double": g["fault%d.segment_double" % j]})
        Xf, Yf = ingest.load_data_fault_raw(pf, verbose=False)
        faults.append((Xf, Yf, "fault_%d" % (j + 1)))
    return Xn, Yn, faults


def test_loaders_filter_rows_like_the_reference(tmp_path):
    g = load_golden("g_ingest.npz")
    Xn, Yn, faults = _recordings(g, str(tmp_path))
    assert np.array_equal(Xn, g["normal.X"]) and np.array_equal(Yn, g["normal.Y"])
    # 50 < I < 800 is strict on both sides; four rows of the recording violate it (0, 50 and 800 exactly, 1200)
    assert len(Xn) == len(g["normal.I"]) - 4 and Xn[:, 0].min() > 50 and Xn[:, 0].max() < 800
    for j, (Xf, Yf, _) in enumerate(faults):
        assert np.array_equal(Xf, g["fault%d.X" % j]) and np.array_equal(Yf, g["fault%d.Y" % j])
        assert len(Xf) == len(g["fault%d.segment_double" % j]) - 3 and np.all(Xf[:, 1] != 0)


@pytest.mark.parametrize("case", list(CASES))
def test_combine_and_normalize(case, tmp_path):
    from pinn_amd import ingest
    g = load_golden("g_ingest.npz")
    Xn, Yn, faults = _recordings(g, str(tmp_path))
    state = np.random.get_state()[1].copy()
    ds = ingest.combine_and_normalize_datasets((Xn, Yn), faults, training_rate=0.8, noise_config=CASES[case], seed=42, verbose=False)
    assert np.array_equal(np.random.get_state()[1], state)          # numpy's global generator is left alone
    pre = "combine.%s." % case
    assert len(ds) == 7
    for i, name in enumerate(("x_train", "y_train", "x_test", "y_test")):
        assert ds[i].dtype.is_floating_point and ds[i].dtype.itemsize == 4
        assert np.array_equal(ds[i].numpy(), g[pre + name]), name
    for sc, tag in ((ds[4], "sx."), (ds[5], "sy.")):
        for attr in ("min_", "scale_", "data_min_", "data_max_"):
            assert np.array_equal(np.asarray(getattr(sc, attr), np.float64).reshape(-1), g[pre + tag + attr].reshape(-1)), tag + attr
        assert tuple(sc.feature_range) == (-1, 1)
    info = ds[6]
    assert list(info["boundary_lines"]) == g[pre + "boundary_lines"].tolist()
    assert np.array_equal(info["train_indices"], g[pre + "train_indices"])
    assert [info["normal_samples"], info["fault_samples"]] == g[pre + "counts"].tolist()
    assert np.array_equal(info["Y_combined"], g[pre + "Y_combined"])            # clean, also with a noise_config (01:274)
    assert np.array_equal(info["Y_combined_scaled"], g[pre + "Y_combined_scaled"])
    assert len(info["data_labels"]) == len(ds[2]) and info["data_labels"][-1] == "fault_2"
    assert len(ds[0]) == int(len(Xn) * 0.8)
    if CASES[case] is None:
        assert info["noise_info"] is None
    else:
        ni = info["noise_info"]
        assert ni["noise_std"] == float(g[pre + "noise_std"])
        assert np.array_equal(ni["noise_mask"], g[pre + "noise_mask"]) and int(ni["affected_samples"]) == int(g[pre + "affected"])
        Yc = np.vstack([Yn] + [f[1] for f in faults])
        yn, _ = ingest.add_noise_to_combined_data(Yc, **CASES[case], seed=42, verbose=False)
        assert np.array_equal(yn, g["noise.%s.Y_noisy" % case])


def test_ingest_errors():
    from pinn_amd import ingest
    X, Y = np.zeros((10, 8)), np.zeros((10, 1))
    with pytest.raises(ValueError):
        ingest.combine_and_normalize_datasets((X, Y), (X, Y, "a"), verbose=False)              # not a list
    with pytest.raises(ValueError):
        ingest.combine_and_normalize_datasets((X, Y), [(X, Y)], verbose=False)                 # item without label
    with pytest.raises(ValueError):
        ingest.combine_and_normalize_datasets((X, Y), [(np.zeros((4, 7)), np.zeros((4, 1)), "a")], verbose=False)
    with pytest.raises(ValueError):
        ingest.add_noise_to_combined_data(Y, noise_type="pink", verbose=False)
    # 'fault_only' selects nothing (01:88-91)
    yn, ni = ingest.add_noise_to_combined_data(np.arange(10.0).reshape(-1, 1), noise_target="fault_only", verbose=False)
    assert np.array_equal(yn, np.arange(10.0).reshape(-1, 1)) and ni["affected_samples"] == 0
