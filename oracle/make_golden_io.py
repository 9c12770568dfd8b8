"""Generate tests/golden/g_ingest.npz and g_stats.npz by running the REAL reference (build container only).

    MPLBACKEND=Agg python oracle/make_golden_io.py

Covers the steps either side of the hot path: the `.mat` loaders with their row filters,
`combine_and_normalize_datasets` (train-subset scaler fit, with and without a noise_config), and the statistics dict
of `plot_model_results_detailed_split`.  Inputs are synthetic (the real recordings are not available offline); the
fixtures hold those inputs and the reference's numeric outputs only.  Same import recipe and the same single shim as
`make_golden.py`.
"""
import contextlib
import io
import os
import sys
import tempfile

import numpy as np
import scipy.io
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
from make_golden import load_reference, state, scaler_arrays, OUT  # noqa: E402
from pinn_amd import synth  # noqa: E402


def raw_normal(n, seed):
    """A polarisation recording as the reference's loader expects it: one variable per signal; currents outside
    (50, 800) A and on both edges included so the row filter is exercised."""
    X, U = synth.synth_rows(n, seed)
    X = X.copy()
    X[0, 0], X[1, 0], X[2, 0], X[3, 0], X[4, 0] = 0.0, 50.0, 800.0, 50.0001, 799.9999
    X[n // 2, 0] = 1200.0
    keys = ("I", "m_W", "T_W_in", "P_H_in", "P_O_in", "T_W_out", "m_H2", "m_O2")
    d = {k: X[:, [i]] for i, k in enumerate(keys)}
    d["U"] = U
    return d


def raw_fault(n, seed, kind):
    """A fault recording: one [n, 70] matrix `segment_double`; the nine used columns filled from synthetic fault rows,
    the others with noise; some rows with a zero in the second input column (dropped by the loader)."""
    X, U = synth.synth_fault_rows(n, seed, kind)
    rng = np.random.default_rng(seed + 100)
    M = rng.normal(size=(n, 70))
    cols = np.array([20, 25, 65, 68, 69, 66, 14, 16]) - 3
    M[:, cols] = X
    M[:, 16] = U[:, 0]
    M[[1, n // 3, n - 1], cols[1]] = 0.0
    return M


def main():
    ref = load_reference()
    out = {}
    with tempfile.TemporaryDirectory() as tmp, contextlib.redirect_stdout(io.StringIO()):
        dn = raw_normal(400, 11)
        pn = os.path.join(tmp, "normal.mat")
        scipy.io.savemat(pn, dn)
        for k, v in dn.items():
            out["normal." + k] = v
        Xn, Yn = ref.load_data_normal_raw(pn)
        out["normal.X"], out["normal.Y"] = Xn, Yn
        faults = []
        for j, (nf, kind) in enumerate(((120, 0), (90, 1))):
            M = raw_fault(nf, 20 + j, kind)
            pf = os.path.join(tmp, "fault%d.mat" % j)
            scipy.io.savemat(pf, {"segment_double": M})
            out["fault%d.segment_double" % j] = M
            Xf, Yf = ref.load_data_fault_raw(pf)
            out["fault%d.X" % j], out["fault%d.Y" % j] = Xf, Yf
            faults.append((Xf, Yf, "fault_%d" % (j + 1)))
        cases = {"plain": None,
                 "gauss_all": {"noise_type": "gaussian", "noise_level": 0.02, "noise_target": "all"},
                 "unif_random": {"noise_type": "uniform", "noise_level": 0.05, "noise_target": "random"}}
        for name, cfg in cases.items():
            ds = ref.combine_and_normalize_datasets((Xn, Yn), faults, training_rate=0.8, noise_config=cfg, seed=42)
            pre = "combine.%s." % name
            out[pre + "x_train"], out[pre + "y_train"] = ds[0].numpy(), ds[1].numpy()
            out[pre + "x_test"], out[pre + "y_test"] = ds[2].numpy(), ds[3].numpy()
            out.update(scaler_arrays(pre + "sx.", ds[4])); out.update(scaler_arrays(pre + "sy.", ds[5]))
            info = ds[6]
            out[pre + "boundary_lines"] = np.array(info["boundary_lines"])
            out[pre + "train_indices"] = info["train_indices"]
            out[pre + "Y_combined"] = info["Y_combined"]
            out[pre + "Y_combined_scaled"] = info["Y_combined_scaled"]
            out[pre + "counts"] = np.array([info["normal_samples"], info["fault_samples"]])
            if cfg is not None:
                ni = info["noise_info"]
                out[pre + "noise_std"] = np.array(ni["noise_std"])
                out[pre + "noise_mask"] = ni["noise_mask"]
                out[pre + "affected"] = np.array(ni["affected_samples"])
        # the noisy targets themselves (combine discards them): one direct call per noise type
        Yc = np.vstack([Yn] + [f[1] for f in faults])
        for name, cfg in cases.items():
            if cfg is not None:
                yn, _ = ref.add_noise_to_combined_data(Yc, **cfg, seed=42)
                out["noise.%s.Y_noisy" % name] = yn
    np.savez_compressed(os.path.join(OUT, "g_ingest.npz"), **out)
    print("g_ingest.npz ok")

    # ------------------------------------------------------------------ statistics dict (01:1764-1828)
    ds = synth.make_dataset(300, (150, 250), seed=5)
    torch.manual_seed(51)
    with contextlib.redirect_stdout(io.StringIO()):
        m = ref.PhysicsInformedNN(ds[0], ds[1], [8, 128, 128, 128, 1], ds[4], ds[5], p=0.2, logvar=True)
        # move the thermal parameters off their all-10 start so that every term of net_f_T matters
        with torch.no_grad():
            m.lambda_T1.fill_(0.9); m.lambda_T2.fill_(35000.0); m.lambda_T3.fill_(1.5); m.lambda_T4.fill_(4.0)
        st = ref.plot_model_results_detailed_split(m, ds, windows=100)
    out = {"x_test": ds[2].numpy(), "y_test": ds[3].numpy(), "x_train": ds[0].numpy(), "y_train": ds[1].numpy(),
           "boundary_lines": np.array(ds[6]["boundary_lines"]),
           "lambda_T": np.array([0.9, 35000.0, 1.5, 4.0, 10.0])}
    out.update(scaler_arrays("sx.", ds[4])); out.update(scaler_arrays("sy.", ds[5]))
    out.update({"w." + k: v for k, v in state(m.dnn).items()})
    for k, v in st.items():
        out["stat." + k] = np.array(v, dtype=np.float64)
    np.savez_compressed(os.path.join(OUT, "g_stats.npz"), **out)
    print("g_stats.npz ok", {k: float(v) for k, v in st.items()})


if __name__ == "__main__":
    main()
