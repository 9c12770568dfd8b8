"""get_MC_samples at the reference's own sizes (T = 64 / 2000 on 4200 and 11 000 rows): python tools/time_mc_small.py"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [R]
import torch, pinn_amd
from pinn_amd import synth
for N in (4200, 11000):
    ds = synth.make_dataset(N, (), seed=0)
    m = pinn_amd.PhysicsInformedNN(ds[0], ds[1], [8, 256, 256, 256, 1], ds[4], ds[5], p=0.2, logvar=True)
    m.verbose = False
    for T in (64, 2000):
        pinn_amd.get_MC_samples(m, ds[2], ds[4], mc_times=T, dropout=0.4); torch.cuda.synchronize()
        t0 = time.perf_counter(); pinn_amd.get_MC_samples(m, ds[2], ds[4], mc_times=T, dropout=0.4); torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print("get_MC_samples N=%d T=%d: %.1f ms (%.1f us / pass, %.2e row-passes/s)" % (N, T, dt * 1e3, dt / T * 1e6, N * T / dt), flush=True)
