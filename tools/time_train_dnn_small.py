"""train_dnn steps/s at the reference's real data size (N ~ 1e4 rows): launch-bound regime."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [R]
import torch
import pinn_amd
from pinn_amd import synth
for N in (int(a) for a in (sys.argv[1:] or ["10000"])):
    ds = synth.make_dataset(N, (), seed=0)
    for prec in ("f32x6", "fp32"):
        m = pinn_amd.PhysicsInformedNN(ds[0], ds[1], [8, 256, 256, 256, 1], ds[4], ds[5], p=0.2, logvar=True, precision=prec)
        m.verbose = False
        m.train_dnn(20); torch.cuda.synchronize()
        t0 = time.perf_counter(); m.train_dnn(500); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print("train_dnn %-6s N=%d: %.1f us / step" % (prec, N, dt / 500 * 1e6), flush=True)
