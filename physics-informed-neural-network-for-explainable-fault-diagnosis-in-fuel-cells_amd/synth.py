"""Synthetic fuel-cell rows and the MinMax(-1, 1) affine map (host side, numpy only).

No data ships with the reference (its .mat inputs are private paths,
01_train_pinn_multiphysics_model.py:2057-2084), so every test / bench feeds rows
drawn from the generator SURVEY.md §8(d) D1 specifies, pushed through the same
affine normalisation `combine_and_normalize_datasets` applies
(01_train_pinn_multiphysics_model.py:271-289): MinMaxScaler(feature_range=(-1, 1))
fitted on the normal-training rows only, then `transform` on everything.

Column order (01_train_pinn_multiphysics_model.py:136-137):
  0 I [A]   1 coolant flow   2 T_in [degC]   3 P_H2_in   4 P_air_in
  5 T_out [degC]   6 H2 flow [slpm]   7 air flow [slpm]      target: stack voltage U [V]
"""
import numpy as np

FARADAY = 96485.0
N_CELLS = 5.0
V_MOLAR = 22.4


class MinMaxAffine:
    """The subset of sklearn.preprocessing.MinMaxScaler the hot path touches.

    Attributes match sklearn's (`feature_range`, `data_min_`, `data_max_`, `scale_`, `min_`),
    which is what 01:1017-1022 and 01:1920-1925 read, and `inverse_transform` reproduces
    sklearn's dtype behaviour: a float32 array stays float32, `X -= min_; X /= scale_`
    are each computed against float64 operands and rounded back to float32.
    A real sklearn scaler can be passed anywhere this class is accepted.
    """

    def __init__(self, feature_range=(-1, 1)):
        self.feature_range = tuple(feature_range)

    def fit(self, X):
        X = np.asarray(X, dtype=np.float64)
        lo, hi = self.feature_range
        self.data_min_ = X.min(axis=0)
        self.data_max_ = X.max(axis=0)
        rng = self.data_max_ - self.data_min_
        rng = np.where(rng == 0.0, 1.0, rng)          # sklearn: _handle_zeros_in_scale
        self.data_range_ = self.data_max_ - self.data_min_
        self.scale_ = (hi - lo) / rng
        self.min_ = lo - self.data_min_ * self.scale_
        self.n_features_in_ = X.shape[1]
        return self

    def transform(self, X):
        X = np.array(X, dtype=np.float64 if np.asarray(X).dtype != np.float32 else np.float32, copy=True)
        X *= self.scale_
        X += self.min_
        return X

    def fit_transform(self, X):
        return self.fit(X).transform(X)

    def inverse_transform(self, X):
        X = np.asarray(X)
        dt = np.float32 if X.dtype == np.float32 else np.float64
        X = np.array(X, dtype=dt, copy=True)
        X -= self.min_
        X /= self.scale_
        return X


def synth_rows(n, seed=0):
    """n physical rows [n, 8] float64 + stack voltage [n, 1] float64.

    I ~ U(54, 405) A keeps i = I/270 below the limiting current density lambda_3 >= 2.0,
    so V_conc = log(1 - i/il) (01:760) stays finite.
    """
    rng = np.random.default_rng(seed)
    I = rng.uniform(54.0, 405.0, n)
    m_cool = rng.uniform(5.0, 40.0, n)
    T_in = rng.uniform(55.0, 70.0, n)
    P_H2 = rng.uniform(20.0, 150.0, n)
    P_air = rng.uniform(10.0, 130.0, n)
    T_out = T_in + rng.uniform(2.0, 10.0, n)
    h2 = I * N_CELLS / (2.0 * FARADAY) * V_MOLAR * 60.0 * rng.uniform(1.2, 2.0, n)
    air = I * N_CELLS / (4.0 * FARADAY) * V_MOLAR * 60.0 / 0.21 * rng.uniform(1.8, 3.0, n)
    U = 5.0 * (0.95 - 0.0004 * I - 0.03 * np.log(I / 50.0)) + rng.normal(0.0, 0.01, n)
    X = np.column_stack([I, m_cool, T_in, P_H2, P_air, T_out, h2, air])
    return X, U.reshape(-1, 1)


def synth_fault_rows(n, seed, kind):
    """A fault segment = normal rows with one drifting column (layout tests only)."""
    X, U = synth_rows(n, seed)
    ramp = np.linspace(0.0, 1.0, n)
    k = kind % 4
    if k == 0:      # flooding-like: voltage sags
        U[:, 0] -= 0.4 * ramp
    elif k == 1:    # oxygen starvation: air flow collapses
        X[:, 7] *= (1.0 - 0.6 * ramp)
        U[:, 0] -= 0.3 * ramp
    elif k == 2:    # membrane drying: outlet temperature climbs
        X[:, 5] += 8.0 * ramp
        U[:, 0] -= 0.2 * ramp
    else:           # hydrogen starvation
        X[:, 6] *= (1.0 - 0.5 * ramp)
        U[:, 0] -= 0.35 * ramp
    return X, U


def make_dataset(n_normal, fault_sizes=(), seed=0, as_torch=True):
    """The 7-tuple `combine_and_normalize_datasets` returns (01:386), on synthetic rows.

    (x_train, y_train, x_test, y_test, scaler_X, scaler_Y, data_info); training rows are
    the normal rows (training_rate=1, 01:2132), test rows are normal + every fault segment
    in order, `data_info['boundary_lines']` holds each segment's exclusive end (01:334-338).
    """
    Xn, Yn = synth_rows(n_normal, seed)
    faults = []
    for k, nf in enumerate(fault_sizes):
        Xf, Yf = synth_fault_rows(int(nf), seed + 1 + k, k)
        faults.append((Xf, Yf, "fault_%d" % (k + 1)))
    sx = MinMaxAffine((-1, 1)).fit(Xn)
    sy = MinMaxAffine((-1, 1)).fit(Yn)
    X_all = np.vstack([Xn] + [f[0] for f in faults])
    Y_all = np.vstack([Yn] + [f[1] for f in faults])
    x_all = sx.transform(X_all).astype(np.float32)
    y_all = sy.transform(Y_all).astype(np.float32)
    boundary, pos = [n_normal], n_normal
    for f in faults:
        pos += len(f[0])
        boundary.append(pos)
    info = {
        "normal_samples": n_normal,
        "fault_samples": len(X_all) - n_normal,
        "X_combined": X_all,
        "Y_combined": Y_all,
        "fault_data_list": faults,
        "boundary_lines": boundary,
        "train_indices": np.arange(n_normal),
    }
    x_train, y_train = x_all[:n_normal], y_all[:n_normal]
    if as_torch:
        import torch
        return (torch.from_numpy(x_train.copy()), torch.from_numpy(y_train.copy()),
                torch.from_numpy(x_all), torch.from_numpy(y_all), sx, sy, info)
    return (x_train, y_train, x_all, y_all, sx, sy, info)
