"""Results assembly: the 22-column `comprehensive_results` array of 01:1877-2010.

The numeric columns come from the HIP kernels (MC-dropout launch + one fused residual pass);
de-normalisation, the per-segment centred moving average (pandas even-window semantics,
01:1832-1834), labels and the float64 [N,22] fill are host-side numpy, as in the reference.
`scipy.io.savemat('F01_output.mat', {'comprehensive_results': arr})` (01:2185-2186) then
gives the file scripts 02-05 read.
"""
import numpy as np
import torch

from . import _lib
from .mc import get_MC_samples


def _moving_average_centered(arr, window):
    """`pd.Series(arr).rolling(window, center=True, min_periods=1).mean()` (01:1830-1845).

    pandas' centred window of even size w covers rows [i - w//2, i + w//2 - 1] (clipped to the
    array); odd w covers [i - w//2, i + w//2].  O(n) via a float64 prefix sum, no pandas needed."""
    arr = np.asarray(arr, dtype=np.float64)
    n = len(arr)
    if n == 0:
        return arr
    half = window // 2
    right = half - 1 if window % 2 == 0 else half
    idx = np.arange(n)
    s = np.maximum(0, idx - half)
    e = np.minimum(n, idx + right + 1)
    e = np.maximum(e, s)
    cs = np.concatenate([[0.0], np.cumsum(arr)])
    cnt = (e - s).astype(np.float64)
    out = np.full(n, np.nan)
    ok = cnt > 0
    out[ok] = (cs[e[ok]] - cs[s[ok]]) / cnt[ok]
    return out


def smooth_by_segments(values, boundary_lines, window):
    """01:1848-1872."""
    values = np.asarray(values, dtype=float).copy()
    n = len(values)
    out = np.empty_like(values, dtype=float)
    boundary_lines = list(boundary_lines) if boundary_lines is not None else []
    if not boundary_lines or boundary_lines[-1] != n:
        if not boundary_lines or boundary_lines[-1] < n:
            return _moving_average_centered(values, window)
        boundary_lines = [b for b in boundary_lines if 0 < b <= n]
    starts = [0] + boundary_lines[:-1]
    for s, e in zip(starts, boundary_lines):
        out[s:e] = _moving_average_centered(values[s:e], window)
    return out


def create_fault_labels(n_samples, data_info, verbose=False):
    """01:2013-2047: 0 = normal rows, k = k-th fault segment."""
    fault_labels = np.zeros(n_samples)
    if data_info and 'boundary_lines' in data_info:
        if 'fault_data_list' in data_info:
            for i, (_, _, label) in enumerate(data_info['fault_data_list']):
                start_idx = data_info['boundary_lines'][i]
                end_idx = data_info['boundary_lines'][i + 1]
                fault_labels[start_idx:end_idx] = i + 1
                if verbose:
                    print(f"fault label {i + 1}: {label}, rows [{start_idx}:{end_idx - 1}]")
    return fault_labels


def create_comprehensive_results_array_v2(model, dataset, mc_times=2000, dropout=0.2):
    """01:1877-2010.  Accepts the 7- or 9-tuple dataset (01:1900-1903); returns float64 [N, 22]:
    0-7 inputs, 8 y_true, 9 y_pred, 10 ale (smoothed), 11 epi (smoothed), 12 y_true - y_pred,
    13 f_V, 14 f_T, 15 f_H2, 16 f_O2, 17 label, 18 V_phys*5, 19 T_phys, 20 ratio_H, 21 ratio_O."""
    if len(dataset) == 9:
        x_train, y_train, x_val, y_val, x_test, y_test, scaler_X, scaler_Y, data_info = dataset
    else:
        x_train, y_train, x_test, y_test, scaler_X, scaler_Y, data_info = dataset

    x_test_np = x_test.detach().cpu().numpy()
    y_test_np = y_test.detach().cpu().numpy()
    x_test_rescaled = scaler_X.inverse_transform(x_test_np)
    y_test_rescaled = scaler_Y.inverse_transform(y_test_np).flatten()

    pred_mean_norm, ale_std_norm, epi_std_norm = get_MC_samples(model, x_test, scaler_X, mc_times=mc_times, dropout=dropout)

    lo_y, hi_y = float(scaler_Y.feature_range[0]), float(scaler_Y.feature_range[1])
    data_min_y = np.asarray(scaler_Y.data_min_).astype(np.float64)
    data_max_y = np.asarray(scaler_Y.data_max_).astype(np.float64)
    scale_y = (hi_y - lo_y) / (data_max_y - data_min_y + 1e-12)
    min_y = lo_y - data_min_y * scale_y
    pred_mean_rescaled = np.asarray((pred_mean_norm - min_y) / (scale_y + 1e-12)).reshape(-1)
    ale_std_rescaled = np.asarray(ale_std_norm / (scale_y + 1e-12)).reshape(-1)
    epi_std_rescaled = np.asarray(epi_std_norm / (scale_y + 1e-12)).reshape(-1)
    prediction_residual = y_test_rescaled - pred_mean_rescaled

    # physics residuals + physics-model outputs: one eval forward + one fused residual pass (01:1944-1969)
    model.dnn.eval()
    xd = model._dev_rows(x_test)
    u, _ = model.net_u(xd)
    c = model._residuals(xd, scaler_X, _lib.RES_ALL, u=u.reshape(-1)).cpu().numpy()
    col = lambda n: c[_lib.C[n]]

    smooth_window = 200
    n_samples = len(x_test)
    boundaries = None
    if data_info and 'boundary_lines' in data_info and len(data_info['boundary_lines']) > 0:
        boundaries = list(data_info['boundary_lines'])
        if boundaries[-1] != n_samples:
            boundaries = boundaries + [n_samples]
    if boundaries:
        ale_std_smooth = smooth_by_segments(ale_std_rescaled, boundaries, smooth_window)
        epi_std_smooth = smooth_by_segments(epi_std_rescaled, boundaries, smooth_window)
    else:
        ale_std_smooth = _moving_average_centered(ale_std_rescaled, smooth_window)
        epi_std_smooth = _moving_average_centered(epi_std_rescaled, smooth_window)

    fault_labels = create_fault_labels(n_samples, data_info)

    results_array = np.zeros((n_samples, 22), dtype=float)
    results_array[:, 0:8] = x_test_rescaled
    results_array[:, 8] = y_test_rescaled
    results_array[:, 9] = pred_mean_rescaled
    results_array[:, 10] = ale_std_smooth
    results_array[:, 11] = epi_std_smooth
    results_array[:, 12] = prediction_residual
    results_array[:, 13] = col("FV")
    results_array[:, 14] = col("FT")
    results_array[:, 15] = col("FH")
    results_array[:, 16] = col("FO")
    results_array[:, 17] = fault_labels
    results_array[:, 18] = col("VEST5")
    results_array[:, 19] = col("TPRED")
    results_array[:, 20] = col("ACTH")
    results_array[:, 21] = col("ACTO")
    return results_array
