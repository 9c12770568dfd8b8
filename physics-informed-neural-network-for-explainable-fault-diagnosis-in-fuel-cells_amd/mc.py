"""`get_MC_samples` (01:1413-1491) on the persistent MC-dropout kernel."""
import numpy as np
import torch


def get_MC_samples(network, X, x_scal, mc_times=64, dropout=0.6, device_outputs=False):
    """(pred_mean [N], a_u [N], e_u [N]) float32 numpy, normalised units -- same contract as 01:1413-1491.

    The reference runs `mc_times` identical eval passes, then `mc_times` stochastic `predict`
    calls (each 2 DNN forwards) and reduces 3*T*N floats on the host.  Here ONE launch does
    1 eval pass + mc_times stochastic passes per row tile and reduces on chip:
        pred_mean = eval forward                         (01:1480: mean of T identical copies)
        a_u = sqrt(exp(mean_t logvar_t))                 (01:1483)
        e_u = sqrt(mean_t u_t^2 - (mean_t u_t)^2)        (01:1486: np.var, ddof=0)
    Dropout probability is overridden on ALL Dropout modules for the stochastic passes and
    restored afterwards, and the net is left in eval mode, exactly as 01:1449-1473 do.
    device_outputs=True (not in the reference) returns the three [N] device tensors instead of numpy arrays.
    """
    original = {}
    for name, module in network.dnn.named_modules():
        if isinstance(module, torch.nn.Dropout):
            original[name] = module.p
    network.dnn.eval()
    for name, module in network.dnn.named_modules():
        if isinstance(module, torch.nn.Dropout):
            module.p = dropout
    try:
        network.dnn.train()
        row_offset = network.row_offset if X.shape[0] == network.n_local else 0
        pm, au, eu = network.mc_dropout(X, mc_times, row_offset=row_offset)
        if not device_outputs:
            network.dnn.check_range()          # (the host waits for the results here anyway)
        out = (pm, au, eu) if device_outputs else (pm.cpu().numpy(), au.cpu().numpy(), eu.cpu().numpy())
    finally:
        for name, module in network.dnn.named_modules():
            if isinstance(module, torch.nn.Dropout):
                module.p = original[name]
        network.dnn.eval()
    if device_outputs:
        return out
    return tuple(np.asarray(o).squeeze() for o in out)
