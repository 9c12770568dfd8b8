"""A few MC-dropout launches at a small row count (4-wave kernels, one wave per SIMD) for rocprofv3."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _common as hh
from _common import _lib, lib
H, nh, N, T = 256, 3, int(sys.argv[1]) if len(sys.argv) > 1 else 16384, 128
fp = hh.random_params(H, nh)
x = torch.rand(N, 8, device=hh.dev())
net = hh.make_net(H, nh, 2)
d = hh.dropout_struct(1, [0.4] * 4, seed=99, stream_id=1000)
out = torch.empty(3, N, device=hh.dev())
for _ in range(3):
    _lib.check(lib.pinn_mc_dropout(ctypes.byref(net), hh.ptr(fp), hh.ptr(x), N, ctypes.byref(d), T, hh.ptr(out[0]), hh.ptr(out[1]), hh.ptr(out[2]), hh.stream()), "mc")
torch.cuda.synchronize()
