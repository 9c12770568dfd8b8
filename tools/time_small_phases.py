"""Phases of one training-gradient call at the reference's row counts, us: python tools/time_small_phases.py [N ...]
(pack + forward, pack + backward, weight gradients, slab reduction, everything; default precision f32x6)"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import _common as hh
from _common import _lib, lib
H, nh = 256, 3
fp = hh.random_params(H, nh)
for N in [int(a) for a in sys.argv[1:]] or [4200, 10000]:
    x = torch.rand(N, 8, device=hh.dev()); y = torch.rand(N, device=hh.dev())
    drop = hh.dropout_struct(1, [0.2] * 4, seed=1, stream_id=2)
    net = hh.make_net(H, nh, 2)
    wb = lib.pinn_train_workspace_bytes(ctypes.byref(net), N)
    work = torch.empty(wb, dtype=torch.uint8, device=hh.dev())
    grads = torch.empty(fp.numel(), device=hh.dev()); loss = torch.zeros(4, dtype=torch.float64, device=hh.dev())
    def run(ph):
        _lib.check(lib.pinn_mlp_train_grads_phases(ctypes.byref(net), hh.ptr(fp), hh.ptr(x), hh.ptr(y), N, N, ctypes.byref(drop), hh.ptr(grads),
                                                   hh.ptr(loss), hh.ptr(work), wb, hh.stream(), ph), "train")
    out = []
    for ph in (8, 16, 2, 4, 7):
        run(7)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(20): run(ph)
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): g.replay()
        e1.record(); torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / 200 * 1e3)
    print("N=%d: pack+fwd %.1f  pack+bwd %.1f  wgrad %.1f  reduce %.1f  all %.1f us" % (N, *out), flush=True)
