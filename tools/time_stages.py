"""Iterations/s of the physics-parameter stage trainers at the reference's real data size (N ~ 1e4 rows)."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [R]
import torch
import pinn_amd
from pinn_amd import synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
ds = synth.make_dataset(N, (), seed=0)
m = pinn_amd.PhysicsInformedNN(ds[0], ds[1], [8, 256, 256, 256, 1], ds[4], ds[5], p=0.2, logvar=True)
m.verbose = False
for name, fn, it in (("train_thermal", m.train_thermal, 3000), ("train_hydrogen", m.train_hydrogen, 3000), ("train_oxygen", m.train_oxygen, 3000),
                     ("train_lambda(False)", lambda n: m.train_lambda(n, False), 3000), ("train_lambda(True)", lambda n: m.train_lambda(n, True), 3000)):
    fn(50); torch.cuda.synchronize()
    t0 = time.perf_counter(); fn(it); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("%-20s N=%d: %.1f us / iteration (%d iterations in %.3f s)" % (name, N, dt / it * 1e6, it, dt), flush=True)
