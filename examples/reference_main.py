#!/usr/bin/env python
"""The reference's `__main__` (01:2055-2201) on the MI355X path, end to end:

    recordings (.mat) -> load_data_*_raw -> combine_and_normalize_datasets -> PhysicsInformedNN
    -> train_dnn / train_lambda x2 / train_dnn / train_thermal / train_hydrogen / train_oxygen
    -> create_comprehensive_results_array_v2 -> F01_output.mat -> statistics report

Only the import line differs from the reference's driver code.  The real recordings are not public, so without
`--normal/--fault` paths synthetic recordings with the reference's MAT variable names are written to a temporary
directory first.  `--scale 1` runs the reference's full schedule (58 009 optimizer steps, 2000 MC passes); the default
0.01 is a smoke-sized run.

    python examples/reference_main.py --out /tmp/F01_output.mat
"""
import argparse
import os
import sys
import tempfile
import time

import numpy as np
import scipy.io

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pinn_amd import (PhysicsInformedNN, combine_and_normalize_datasets, create_comprehensive_results_array_v2,   # noqa: E402
                      load_data_fault_raw, load_data_normal_raw, plot_model_results_detailed_split, synth)


def write_synthetic_recordings(folder, n_normal=3000, fault_sizes=(400, 400, 400), seed=0):
    """Recordings in the two formats the reference reads (01:115-187): one variable per signal / one `segment_double` matrix."""
    X, U = synth.synth_rows(n_normal, seed)
    keys = ("I", "m_W", "T_W_in", "P_H_in", "P_O_in", "T_W_out", "m_H2", "m_O2")
    normal = os.path.join(folder, "Polar-1.mat")
    scipy.io.savemat(normal, dict({k: X[:, [i]] for i, k in enumerate(keys)}, U=U))
    faults = []
    cols = np.array([20, 25, 65, 68, 69, 66, 14, 16]) - 3
    for k, n in enumerate(fault_sizes):
        Xf, Uf = synth.synth_fault_rows(n, seed + 1 + k, k)
        M = np.zeros((n, 70))
        M[:, cols] = Xf
        M[:, 16] = Uf[:, 0]
        path = os.path.join(folder, "fault_%d.mat" % (k + 1))
        scipy.io.savemat(path, {"segment_double": M})
        faults.append((path, "fault_%d" % (k + 1)))
    return normal, faults


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--normal", help="normal-operation recording (.mat, 01:115-154); default: synthetic")
    ap.add_argument("--fault", action="append", default=[], help="fault recording (.mat, 01:157-187); repeatable")
    ap.add_argument("--scale", type=float, default=0.01, help="fraction of the reference's iteration counts / MC passes")
    ap.add_argument("--out", default="F01_output.mat")
    ap.add_argument("--quiet", action="store_true")
    args = ap.parse_args(argv)
    it = lambda n: max(2, int(round(n * args.scale)))

    with tempfile.TemporaryDirectory() as tmp:
        if args.normal:
            normal, faults = args.normal, [(p, os.path.splitext(os.path.basename(p))[0]) for p in args.fault]
        else:
            normal, faults = write_synthetic_recordings(tmp)
        X_normal, Y_normal = load_data_normal_raw(normal, verbose=not args.quiet)
        fault_data_list = []
        for path, label in faults:
            X_fault, Y_fault = load_data_fault_raw(path, verbose=not args.quiet)
            fault_data_list.append((X_fault, Y_fault, label))
    Dataset = combine_and_normalize_datasets(normal_data=(X_normal, Y_normal), fault_data_list=fault_data_list, training_rate=1,
                                             noise_config=None, seed=42, verbose=not args.quiet)                  # 01:2127-2133

    t0 = time.perf_counter()
    Layers = [Dataset[0].shape[1], 256, 256, 256, Dataset[1].shape[1]]                                           # 01:2139
    Model_Pinn = PhysicsInformedNN(*Dataset[0:2], Layers, Dataset[4], Dataset[5], p=0.2, logvar=True)
    Model_Pinn.verbose = not args.quiet
    Model_Pinn.train_dnn(nIter=it(4001))
    Model_Pinn.train_lambda(nIter=it(4001), dnn_para=False)
    Model_Pinn.train_lambda(nIter=it(4001), dnn_para=True)
    Model_Pinn.train_dnn(nIter=it(8001))
    Model_Pinn.train_thermal(nIter=it(10001))
    Model_Pinn.train_hydrogen(nIter=it(8001))
    Model_Pinn.train_oxygen(nIter=it(8001))
    comprehensive_results = create_comprehensive_results_array_v2(Model_Pinn, Dataset, mc_times=it(2000), dropout=0.4)   # 01:2155-2158
    scipy.io.savemat(args.out, {'comprehensive_results': comprehensive_results})                                  # 01:2185-2186
    stats = None
    if not args.quiet:
        stats = plot_model_results_detailed_split(Model_Pinn, Dataset, fig_title="Fuel cell voltage prediction", windows=100)   # 01:2198
    import torch
    torch.cuda.synchronize()
    if not args.quiet:
        print("rows %d, results %s -> %s, %.2f s for training + inference" % (len(Dataset[2]), comprehensive_results.shape, args.out,
                                                                               time.perf_counter() - t0))
    return comprehensive_results, stats


if __name__ == "__main__":
    main()
