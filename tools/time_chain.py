"""Time the chain phase (forward + loss + backward chain) of pinn_mlp_train_grads at 1e6 rows: python tools/time_chain.py [PREC ...]."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import _common as hh
from _common import _lib, lib
H, nh, N = 256, 3, int(os.environ.get("PINN_N", "1000000"))
fp = hh.random_params(H, nh)
x = torch.rand(N, 8, device=hh.dev()); y = torch.rand(N, device=hh.dev())
drop = hh.dropout_struct(1, [0.2] * 4, seed=1, stream_id=2)
for prec in [int(a) for a in sys.argv[1:]] or [2]:
    net = hh.make_net(H, nh, prec)
    wb = lib.pinn_train_workspace_bytes(ctypes.byref(net), N)
    work = torch.zeros(wb, dtype=torch.uint8, device=hh.dev())
    grads = torch.empty(fp.numel(), device=hh.dev()); loss = torch.zeros(4, dtype=torch.float64, device=hh.dev())
    def run(ph):
        _lib.check(lib.pinn_mlp_train_grads_phases(ctypes.byref(net), hh.ptr(fp), hh.ptr(x), hh.ptr(y), N, N, ctypes.byref(drop), hh.ptr(grads),
                                                   hh.ptr(loss), hh.ptr(work), wb, hh.stream(), ph), "train")
    for _ in range(2): run(1)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): run(1)
    e1.record(); torch.cuda.synchronize()
    print("prec %d: chain %.3f ms" % (prec, e0.elapsed_time(e1) / 5), flush=True)
