"""Forward / MC-dropout / training-chain latency at small row counts (the reference's data sizes)."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _common as hh
from _common import _lib, lib
H, nh = 256, 3
fp = hh.random_params(H, nh)
for N in (int(a) for a in (sys.argv[1:] or ["2000", "10000", "16000"])):
    x = torch.rand(N, 8, device=hh.dev()); y = torch.rand(N, device=hh.dev())
    net = hh.make_net(H, nh, 2)
    d = hh.dropout_struct(1, [0.4] * 4, seed=99, stream_id=1000)
    out = torch.empty(3, N, device=hh.dev())
    T = 256
    def mc():
        _lib.check(lib.pinn_mc_dropout(ctypes.byref(net), hh.ptr(fp), hh.ptr(x), N, ctypes.byref(d), T, hh.ptr(out[0]), hh.ptr(out[1]), hh.ptr(out[2]), hh.stream()), "mc")
    wb = lib.pinn_train_workspace_bytes(ctypes.byref(net), N)
    work = torch.empty(wb, dtype=torch.uint8, device=hh.dev()); grads = torch.empty(fp.numel(), device=hh.dev()); loss = torch.zeros(4, dtype=torch.float64, device=hh.dev())
    def chain():
        _lib.check(lib.pinn_mlp_train_grads_phases(ctypes.byref(net), hh.ptr(fp), hh.ptr(x), hh.ptr(y), N, N, ctypes.byref(d), hh.ptr(grads), hh.ptr(loss), hh.ptr(work), wb, hh.stream(), 1), "chain")
    res = []
    for fn in (mc, chain):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): fn()
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 5)
    print("N=%6d: MC T=%d %.3f ms (%.3e passes/s)   train chain %.1f us" % (N, T, res[0], N * T / res[0] * 1e3, res[1] * 1e3), flush=True)
