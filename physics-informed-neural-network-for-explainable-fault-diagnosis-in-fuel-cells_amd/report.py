"""Summary statistics of a trained model and a checkpoint.

`model_statistics` is the numeric half of the reference's `plot_model_results_detailed_split`
(01:1626-1828: statistics at 01:1764-1828); the figure half is out of scope.  The name
`plot_model_results_detailed_split` is kept as an alias that takes the reference's arguments (font sizes
ignored), prints the same report and returns the same dict, so the reference's `__main__` (01:2198-2201) runs.
All per-row quantities come from the HIP kernels (`net_f_V`, `predict`, `net_f_T`, `net_f_H`, `net_f_O` of the
model); the reductions over <= 1e4 rows are numpy on the host, in float64 like the reference's.
"""
import numpy as np


def _np(t):
    return t.detach().cpu().numpy().flatten()


def _moving_avg_same(x, w):
    """01:1776-1779: `np.convolve(x, ones(w)/w, mode='same')` (zero-padded edges), identity if len(x) < w."""
    if len(x) < w:
        return x
    return np.convolve(x, np.ones(w) / w, mode='same')


def model_statistics(model, dataset, data_info=None, windows=100, verbose=True):
    """-> {'voltage_mae', 'voltage_rmse', 'voltage_r2', 'physics_v_mae', 'temp_mae_smooth', 'hydrogen_mae',
    'oxygen_mae'} (01:1819-1827) on the TEST rows of the 7-tuple `dataset`; evaluates in eval mode."""
    x_train, y_train, x_test, y_test, scaler_X, scaler_Y, _info = dataset
    y_rescal = scaler_Y.inverse_transform(y_test.detach().cpu().numpy()).flatten()

    model.dnn.eval()
    f_pred = model.net_f_V(x_test, scaler_X)[0]
    u_pred, _ = model.predict(x_test, scaler_X)
    u_pred = scaler_Y.inverse_transform(u_pred).flatten()
    voltage_error = y_rescal - u_pred
    f_V = _np(f_pred)
    f_T = _np(model.net_f_T(x_test, scaler_X)[0])
    f_H, act_H, tgt_H = [_np(t) for t in model.net_f_H(x_test, scaler_X)[:3]]
    f_O, act_O, tgt_O = [_np(t) for t in model.net_f_O(x_test, scaler_X)[:3]]

    mae = lambda e: np.mean(np.abs(e))
    rmse = lambda e: np.sqrt(np.mean(e ** 2))
    f_T_smooth = _moving_avg_same(f_T, windows)
    stats = {
        'voltage_mae': mae(voltage_error),
        'voltage_rmse': rmse(voltage_error),
        'voltage_r2': 1 - np.sum(voltage_error ** 2) / np.sum((y_rescal - np.mean(y_rescal)) ** 2),
        'physics_v_mae': mae(f_V),
        'temp_mae_smooth': mae(f_T_smooth),
        'hydrogen_mae': mae(f_H),
        'oxygen_mae': mae(f_O),
    }
    if verbose:
        print("=" * 60)
        print("Model prediction statistics")
        print("=" * 60)
        print("Voltage:")
        print("  MAE: %.6f V" % stats['voltage_mae'])
        print("  RMSE: %.6f V" % stats['voltage_rmse'])
        print("  R^2: %.6f" % stats['voltage_r2'])
        print("  Max abs error: %.6f V" % np.max(np.abs(voltage_error)))
        print("Voltage physics consistency:")
        print("  Residual MAE: %.6f" % stats['physics_v_mae'])
        print("  Residual RMSE: %.6f" % rmse(f_V))
        print("Temperature physics consistency:")
        print("  Original MAE: %.6f degC -> Smoothed MAE: %.6f degC" % (mae(f_T), stats['temp_mae_smooth']))
        print("  Original RMSE: %.6f degC -> Smoothed RMSE: %.6f degC" % (rmse(f_T), rmse(f_T_smooth)))
        for name, f, act, tgt in (("Hydrogen", f_H, act_H, tgt_H), ("Oxygen", f_O, act_O, tgt_O)):
            print("%s physics consistency:" % name)
            print("  Residual MAE: %.6f" % mae(f))
            print("  Residual RMSE: %.6f" % rmse(f))
            print("  Actual ratio range: [%.3f, %.3f]" % (act.min(), act.max()))
            print("  Target ratio range: [%.3f, %.3f]" % (tgt.min(), tgt.max()))
        print("=" * 60)
    return stats


def plot_model_results_detailed_split(model, dataset, data_info=None, fig_title="Detailed Model Analysis", windows=100,
                                      title_size=16, label_size=26, tick_size=10, legend_size=10):
    """Reference signature (01:1626-1629); no figure is drawn here."""
    return model_statistics(model, dataset, data_info=data_info, windows=windows)


# ---------------------------------------------------------------------------------------
# checkpoint: tensors only, so that `torch.load(path, weights_only=True)` reads it back
# ---------------------------------------------------------------------------------------
CHECKPOINT_VERSION = 1


def save_checkpoint(model, path):
    """Network parameters under the reference's `state_dict` keys (`layers.layer_0.weight`, ..., 01:389-419), the 17
    physics parameters by name, the Adam moments of `train_dnn` (informational: like the reference, every trainer call
    starts a fresh optimizer) and the step / dropout-stream counters, so that a restored model draws the masks the saved
    one would have drawn next."""
    import torch
    from .model import LAMBDA_NAMES
    dnn = model.dnn
    flat = dnn.flat_params()
    ck = {"version": torch.tensor(CHECKPOINT_VERSION), "layers": torch.tensor([dnn.n_in] + [dnn.hidden] * dnn.n_hidden + [1])}
    for name, shape, off in dnn._offsets:
        n = int(np.prod(shape))
        ck["dnn." + name] = flat[off:off + n].view(shape).detach().cpu().clone()
    lam = model._lambdas().detach().cpu()
    for i, name in enumerate(LAMBDA_NAMES):
        ck[name] = lam[i:i + 1].clone()
    ck["adam.m"] = model._adam_m.detach().cpu().clone()
    ck["adam.v"] = model._adam_v.detach().cpu().clone()
    ck["counters"] = torch.tensor([model._step_counter, dnn._fwd_counter, getattr(model, "_mc_calls", 0), dnn.seed], dtype=torch.int64)
    torch.save(ck, path)


def load_checkpoint(model, path):
    """Restores what `save_checkpoint` wrote into an existing model of the same architecture.
    Raises ValueError on an architecture or version mismatch."""
    import torch
    from .model import LAMBDA_NAMES
    ck = torch.load(path, map_location="cpu", weights_only=True)
    if int(ck["version"]) != CHECKPOINT_VERSION:
        raise ValueError("checkpoint version %d, expected %d" % (int(ck["version"]), CHECKPOINT_VERSION))
    dnn = model.dnn
    layers = [dnn.n_in] + [dnn.hidden] * dnn.n_hidden + [1]
    if ck["layers"].tolist() != layers:
        raise ValueError("checkpoint is for layers %s, the model has %s" % (ck["layers"].tolist(), layers))
    flat = dnn.flat_params()
    for name, shape, off in dnn._offsets:
        n = int(np.prod(shape))
        flat[off:off + n].copy_(ck["dnn." + name].reshape(-1))
    lam = model._lambdas()
    for i, name in enumerate(LAMBDA_NAMES):
        lam[i:i + 1].copy_(ck[name])
    model._adam_m.copy_(ck["adam.m"])
    model._adam_v.copy_(ck["adam.v"])
    c = ck["counters"].tolist()
    model._step_counter, dnn._fwd_counter, model._mc_calls, dnn.seed = int(c[0]), int(c[1]), int(c[2]), int(c[3])
    return model
