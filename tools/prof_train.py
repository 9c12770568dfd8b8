"""A few full training-gradient calls at one precision (for rocprofv3): python tools/prof_train.py PREC [N]."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import _common as hh
from _common import _lib, lib
prec = int(sys.argv[1]); N = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
H, nh = 256, 3
fp = hh.random_params(H, nh)
x = torch.rand(N, 8, device=hh.dev()); y = torch.rand(N, device=hh.dev())
drop = hh.dropout_struct(1, [0.2] * 4, seed=1, stream_id=2)
net = hh.make_net(H, nh, prec)
wb = lib.pinn_train_workspace_bytes(ctypes.byref(net), N)
work = torch.empty(wb, dtype=torch.uint8, device=hh.dev())
grads = torch.empty(fp.numel(), device=hh.dev()); loss = torch.zeros(4, dtype=torch.float64, device=hh.dev())
for _ in range(5):
    _lib.check(lib.pinn_mlp_train_grads(ctypes.byref(net), hh.ptr(fp), hh.ptr(x), hh.ptr(y), N, N, ctypes.byref(drop), hh.ptr(grads), hh.ptr(loss),
                                        hh.ptr(work), wb, hh.stream()), "train")
torch.cuda.synchronize()
