"""Ingest + normalisation: the step directly before the hot path.

Same names, arguments, return tuples and error behaviour as the reference's
`add_noise_to_combined_data` (01:59-112), `load_data_normal_raw` (01:115-154),
`load_data_fault_raw` (01:157-187) and `combine_and_normalize_datasets` (01:190-386), so the
reference's `__main__` (01:2089-2135) runs unchanged on top of them.  Host-side numpy only (the
real data sets are 1e3-1e4 rows); the rows go to the GPU once, in `PhysicsInformedNN.__init__`.
What is deliberately different:
  * no matplotlib figure (01:302-368 draws one and calls `plt.show()`); `plot=` is accepted and ignored;
  * the scalers are `synth.MinMaxAffine` (sklearn's `MinMaxScaler` arithmetic and attributes,
    `tests/test_host_logic.py` checks them against sklearn) -- sklearn is not needed at run time;
  * random numbers come from a private `numpy.random.RandomState(seed)`: the same MT19937 stream and
    draw order as the reference's `np.random.seed(seed)`, without reseeding numpy's global state.
"""
import numpy as np

from .synth import MinMaxAffine

# columns of the fault recordings' `segment_double` matrix (1-based sheet columns minus 3, 01:172-176):
# current, coolant flow, coolant inlet temperature, H2 inlet pressure, air inlet pressure, coolant outlet
# temperature, H2 flow, air flow; stack voltage
FAULT_COLUMNS = np.array([20, 25, 65, 68, 69, 66, 14, 16]) - 3
FAULT_TARGET_COLUMN = 19 - 3
NORMAL_KEYS = ("I", "m_W", "T_W_in", "P_H_in", "P_O_in", "T_W_out", "m_H2", "m_O2")


def _log(verbose, *a):
    if verbose:
        print(*a)


def add_noise_to_combined_data(Y_data, noise_type='gaussian', noise_level=0.02, noise_target='fault_only', seed=42,
                               verbose=True):
    """01:59-112.  Returns (Y_noisy, noise_info).  `noise_level` is relative to std(Y_data).
    noise_target: 'all' | 'random' (half of the rows, without replacement) | 'fault_only' / 'normal_only'
    (as in the reference these two select nothing: the caller would have to supply the segment lengths).
    Raises ValueError for an unknown noise_type."""
    rng = np.random.RandomState(seed)
    Y_data = np.asarray(Y_data)
    noise_std = noise_level * np.std(Y_data)
    if noise_type == 'gaussian':
        noise = rng.normal(0, noise_std, Y_data.shape)
    elif noise_type == 'uniform':
        width = noise_std * np.sqrt(12)
        noise = rng.uniform(-width / 2, width / 2, Y_data.shape)
    else:
        raise ValueError("Unsupported noise type")
    noise_mask = np.zeros(len(Y_data), dtype=bool)
    if noise_target == 'all':
        noise_mask[:] = True
    elif noise_target == 'random':
        noise_mask[rng.choice(len(Y_data), size=len(Y_data) // 2, replace=False)] = True
    Y_noisy = Y_data.copy()
    Y_noisy[noise_mask] += noise[noise_mask]
    noise_info = {
        'noise_type': noise_type,
        'noise_level': noise_level,
        'noise_std': noise_std,
        'noise_mask': noise_mask,
        'affected_samples': np.sum(noise_mask),
    }
    _log(verbose, "noise: %s, level %.3f, std %.6f, rows affected %d/%d" %
         (noise_type, noise_level, noise_std, int(np.sum(noise_mask)), len(Y_data)))
    return Y_noisy, noise_info


def _column(a):
    a = np.asarray(a)
    return a.reshape(-1, 1) if a.ndim == 1 else a


def _loadmat(path):
    import scipy.io
    return scipy.io.loadmat(path)


def load_data_normal_raw(data_path, verbose=True):
    """01:115-154.  A polarisation recording: one MAT variable per signal (`I, m_W, T_W_in, P_H_in, P_O_in,
    T_W_out, m_H2, m_O2, U`).  Returns raw (X [n,8], Y [n,1]) of the rows with 50 < I < 800 A."""
    data = _loadmat(data_path)
    cols = [_column(data[k]) for k in NORMAL_KEYS]
    X_data = np.column_stack(cols)
    Y_data = _column(data['U'])
    current = cols[0]
    valid = np.where((current > 50) & (current < 800))[0]
    X_data, Y_data = X_data[valid], Y_data[valid]
    _log(verbose, "normal data: %d rows" % X_data.shape[0])
    return X_data, Y_data


def load_data_fault_raw(data_path, verbose=True):
    """01:157-187.  A fault recording: one matrix `segment_double`; the 8 inputs are FAULT_COLUMNS, the target
    FAULT_TARGET_COLUMN.  Rows whose SECOND input column is exactly 0 are dropped -- the reference's comment says
    "current" but the column it tests is `X_data[:, 1:2]` (01:180), and that is what is kept here."""
    data = _loadmat(data_path)['segment_double']
    X_data = data[:, FAULT_COLUMNS]
    Y_data = data[:, [FAULT_TARGET_COLUMN]]
    valid = np.where(X_data[:, 1:2] != 0)[0]
    X_data, Y_data = X_data[valid], Y_data[valid]
    _log(verbose, "fault data: %d rows" % X_data.shape[0])
    return X_data, Y_data


def combine_and_normalize_datasets(normal_data, fault_data_list, training_rate=0.8, noise_config=None, seed=42,
                                   plot=False, verbose=True):
    """01:190-386.  Returns (x_train, y_train, x_test, y_test, scaler_X, scaler_Y, data_info).

    Training rows = the first int(n_normal * training_rate) NORMAL rows; both scalers (range (-1, 1)) are fitted on
    those rows only (01:262-271); the test set is every row, normal first, then each fault segment in list order;
    `data_info['boundary_lines']` = exclusive end of each segment.  As in the reference, `noise_config` only fills
    `data_info['noise_info']`: the noisy targets are computed (01:243-246) but the arrays that are scaled and
    returned are rebuilt from the clean inputs (01:274-275).
    Raises ValueError if `fault_data_list` is not a list, an item is not (X, Y, label), or a feature count differs."""
    import torch

    X_normal, Y_normal = normal_data
    if not isinstance(fault_data_list, list):
        raise ValueError("fault_data_list must be a list")
    for i, item in enumerate(fault_data_list):
        if len(item) != 3:
            raise ValueError("fault data %d malformed, expected (X_fault, Y_fault, label)" % (i + 1))
        X_fault, _, label = item
        if X_fault.shape[1] != X_normal.shape[1]:
            raise ValueError("%s has %d features, the normal data %d" % (label, X_fault.shape[1], X_normal.shape[1]))

    all_X, all_Y = [X_normal], [Y_normal]
    data_labels = ['正常数据'] * len(X_normal)      # the reference's label for normal rows (01:225), kept for data_info parity
    for X_fault, Y_fault, label in fault_data_list:
        all_X.append(X_fault)
        all_Y.append(Y_fault)
        data_labels.extend([label] * len(X_fault))
    X_combined = np.vstack(all_X)
    Y_combined = np.vstack(all_Y)
    _log(verbose, "combined: %d rows, %d features" % X_combined.shape)

    noise_info = None
    if noise_config is not None:
        _, noise_info = add_noise_to_combined_data(Y_combined, **noise_config, seed=seed, verbose=verbose)

    n_normal = len(X_normal)
    n_train = int(n_normal * training_rate)
    train_indices = np.arange(n_train)
    scaler_X = MinMaxAffine(feature_range=(-1, 1)).fit(X_normal[train_indices])
    scaler_Y = MinMaxAffine(feature_range=(-1, 1)).fit(Y_normal[train_indices])
    X_scaled = scaler_X.transform(X_combined)
    Y_scaled = scaler_Y.transform(Y_combined)

    boundary_lines, pos = [n_normal], n_normal
    for X_fault, _, _ in fault_data_list:
        pos += len(X_fault)
        boundary_lines.append(pos)
    _log(verbose, "train %d rows (normal only), test %d rows, segment ends %s" % (n_train, len(X_scaled), boundary_lines))

    x_train = torch.from_numpy(X_scaled[train_indices]).float()
    y_train = torch.from_numpy(Y_scaled[train_indices]).float()
    x_test = torch.from_numpy(X_scaled).float()
    y_test = torch.from_numpy(Y_scaled).float()
    data_info = {
        'data_labels': data_labels,
        'train_indices': train_indices,
        'normal_samples': n_normal,
        'fault_samples': len(X_combined) - n_normal,
        'X_combined': X_combined,
        'Y_combined': Y_combined,
        'Y_combined_scaled': Y_scaled,
        'noise_info': noise_info,
        'fault_data_list': fault_data_list,
        'boundary_lines': boundary_lines,
    }
    return (x_train, y_train, x_test, y_test, scaler_X, scaler_Y, data_info)
