"""train_dnn at the reference's data size: captured-graph replay against launch by launch (us / step): python tools/time_train_dnn_graph.py [N ...]"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [R]
import torch
import pinn_amd
from pinn_amd import synth
for N in (int(a) for a in (sys.argv[1:] or ["10000"])):
    ds = synth.make_dataset(N, (), seed=0)
    for graph in (False, True):
        m = pinn_amd.PhysicsInformedNN(ds[0], ds[1], [8, 256, 256, 256, 1], ds[4], ds[5], p=0.2, logvar=True)
        m.verbose = False
        m.use_graph = graph
        m.train_dnn(20); torch.cuda.synchronize()
        t0 = time.perf_counter(); m.train_dnn(1000); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print("train_dnn f32x6 N=%d %s: %.1f us / step" % (N, "graph replay" if graph else "launch by launch", dt / 1000 * 1e6), flush=True)
