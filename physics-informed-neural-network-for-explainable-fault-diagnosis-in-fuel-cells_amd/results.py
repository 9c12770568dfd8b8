"""Results assembly: the 22-column `comprehensive_results` array of 01:1877-2010.

Everything per-row runs on the device: the MC-dropout launch, one fused residual pass and
`pinn_results_assemble` (de-normalisation, the per-segment centred moving average with pandas'
even-window semantics of 01:1832-1834, the float64 [N,22] fill); only segment ends and labels (host
metadata) go up and the finished array comes down once.  `scipy.io.savemat('F01_output.mat',
{'comprehensive_results': arr})` (01:2185-2186) then gives the file scripts 02-05 read.
`_moving_average_centered` / `smooth_by_segments` are the reference's free functions (numpy, host).
"""
import ctypes

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

    n_samples = len(x_test)
    smooth_window = 200                                                   # 01:1972
    # MC-dropout with the reference's dropout-rate override / restore (01:1449-1473); outputs stay on the device
    pred_mean, a_u, e_u = get_MC_samples(model, x_test, scaler_X, mc_times=mc_times, dropout=dropout, device_outputs=True)

    # min_y / scale_y as 01:1925-1930 rebuilds them from the target scaler, in float64
    lo_y, hi_y = float(scaler_Y.feature_range[0]), float(scaler_Y.feature_range[1])
    data_min_y = float(np.asarray(scaler_Y.data_min_, dtype=np.float64).reshape(-1)[0])
    data_max_y = float(np.asarray(scaler_Y.data_max_, dtype=np.float64).reshape(-1)[0])
    scale_y = (hi_y - lo_y) / (data_max_y - data_min_y + 1e-12)
    min_y = lo_y - data_min_y * scale_y

    # physics residuals + physics-model outputs: one eval forward + one fused residual pass (01:1944-1969)
    model.dnn.eval()
    xd = model._dev_rows(x_test)
    yd = y_test.detach().to(xd.device, torch.float32).reshape(-1).contiguous()
    u, _ = model.net_u(xd)
    cols = model._residuals(xd, scaler_X, _lib.RES_ALL, u=u.reshape(-1))

    # smoothing segments (01:1974-1986) and labels (01:2013-2047): host metadata, a few integers
    seg_end = None
    if data_info and 'boundary_lines' in data_info and len(data_info['boundary_lines']) > 0:
        boundaries = [int(b) for b in data_info['boundary_lines']]
        if boundaries[-1] != n_samples:
            boundaries = boundaries + [n_samples]
        if not (all(0 < b <= n_samples for b in boundaries) and all(a < b for a, b in zip(boundaries[:-1], boundaries[1:]))):
            # (the reference's slicing would leave rows of an uninitialised np.empty_like array in columns 10-11)
            raise ValueError("data_info['boundary_lines'] must be ascending segment ends within the %d test rows" % n_samples)
        seg_end = torch.tensor(boundaries, dtype=torch.int64, device=xd.device)
    labels = torch.from_numpy(create_fault_labels(n_samples, data_info).astype(np.float32)).to(xd.device)

    # the target scaler of THIS call may differ from the one the model was built with: its own affine map for column 8
    aff = _lib.Affine.from_buffer_copy(model._affine(scaler_X))
    aff.y_min = float(np.asarray(scaler_Y.min_, dtype=np.float64).reshape(-1)[0])
    aff.y_scale = float(np.asarray(scaler_Y.scale_, dtype=np.float64).reshape(-1)[0])
    out = torch.empty(n_samples, 22, dtype=torch.float64, device=xd.device)
    rc = model._lib.pinn_results_assemble(
        xd.data_ptr(), yd.data_ptr(), ctypes.byref(aff), min_y, scale_y, smooth_window,
        None if seg_end is None else seg_end.data_ptr(), 0 if seg_end is None else int(seg_end.numel()),
        pred_mean.data_ptr(), a_u.data_ptr(), e_u.data_ptr(), cols.data_ptr(), n_samples, labels.data_ptr(), n_samples,
        out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    _lib.check(rc, "pinn_results_assemble")
    return out.cpu().numpy()
