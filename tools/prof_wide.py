"""A few wide-net ([8, 1024 x 4, 1]) forward + training-gradient calls for rocprofv3: python tools/prof_wide.py [N]."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import _common as hh
from _common import _lib, lib
H, nh = 1024, 4
N = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
fp = hh.random_params(H, nh)
x = torch.rand(N, 8, device=hh.dev()); y = torch.rand(N, device=hh.dev())
drop = hh.dropout_struct(1, [0.2] * (nh + 1), seed=1, stream_id=2)
net = hh.make_net(H, nh, 2)
wb = lib.pinn_train_workspace_bytes(ctypes.byref(net), N)
work = torch.empty(wb, dtype=torch.uint8, device=hh.dev())
grads = torch.empty(fp.numel(), device=hh.dev()); loss = torch.zeros(4, dtype=torch.float64, device=hh.dev())
for _ in range(3):
    _lib.check(lib.pinn_mlp_train_grads(ctypes.byref(net), hh.ptr(fp), hh.ptr(x), hh.ptr(y), N, N, ctypes.byref(drop), hh.ptr(grads), hh.ptr(loss),
                                        hh.ptr(work), wb, hh.stream()), "train")
torch.cuda.synchronize()
