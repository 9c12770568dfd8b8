import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _common as hh
from _common import _lib, lib
H, nh, N = int(os.environ.get("PINN_H", "256")), 3, 1_000_000
fp = hh.random_params(H, nh)
x = torch.rand(N, 8, device=hh.dev())
PRECS = [int(a) for a in sys.argv[1:]] or [0, 1, 2]
for prec in PRECS:
    for mode in (0, 1):
        drop = hh.dropout_struct(mode, [0.2] * 4, seed=1, stream_id=2)
        for _ in range(3): hh.forward(H, nh, fp, x, drop, precision=prec)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): hh.forward(H, nh, fp, x, drop, precision=prec)
        e1.record(); torch.cuda.synchronize()
        print(f"prec {prec} mode {mode}: forward {e0.elapsed_time(e1)/10:.3f} ms", flush=True)
    out = torch.empty(3, N, device=hh.dev())
    net = hh.make_net(H, nh, prec)
    d = hh.dropout_struct(1, [0.4] * 4, seed=99, stream_id=1000)
    T = 64
    for it in range(2):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _lib.check(lib.pinn_mc_dropout(ctypes.byref(net), hh.ptr(fp), hh.ptr(x), N, ctypes.byref(d), T, hh.ptr(out[0]), hh.ptr(out[1]), hh.ptr(out[2]), hh.stream()), "mc")
        e1.record(); torch.cuda.synchronize()
    print(f"prec {prec}: MC T={T} {e0.elapsed_time(e1):.2f} ms -> {N*T/e0.elapsed_time(e1)*1e3:.3e} passes/s", flush=True)
