"""Diagnostic: cycle split of the small-row-count forward kernel, workgroup 0 (needs a -DPINN_Q_STAMP build loaded via PINN_HIP_LIB)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import _common as hh
from _common import _lib, lib
H, nh, N = 256, 3, int(os.environ.get("PINN_N", "4200"))
fp = hh.random_params(H, nh)
x = torch.rand(N, 8, device=hh.dev()); y = torch.rand(N, device=hh.dev())
drop = hh.dropout_struct(1, [0.2] * 4, seed=1, stream_id=2)
net = hh.make_net(H, nh, 2)
wb = lib.pinn_train_workspace_bytes(ctypes.byref(net), N)
work = torch.empty(wb, dtype=torch.uint8, device=hh.dev())
grads = torch.empty(fp.numel(), device=hh.dev()); loss = torch.zeros(4, dtype=torch.float64, device=hh.dev())
for _ in range(3):
    _lib.check(lib.pinn_mlp_train_grads_phases(ctypes.byref(net), hh.ptr(fp), hh.ptr(x), hh.ptr(y), N, N, ctypes.byref(drop), hh.ptr(grads),
                                               hh.ptr(loss), hh.ptr(work), wb, hh.stream(), 8), "train")
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 128)()
lib.pinn_q_debug_read.restype = ctypes.c_int
print("rc", lib.pinn_q_debug_read(buf))
names = ["fill_small", "prime", "layer0 mfma", "layer0 publish", "barrier", "hidden steps (all)", "hidden publish (all)", "hidden barriers", "v0 steps",
         "v0 publish+barrier", "v1 steps", "heads+loss", "tail"]
for w in range(8):
    s = [buf[w * 16 + k] for k in range(13)]
    print("wave %d: " % w + "  ".join("%s %d" % (n, v) for n, v in zip(names, s)) + "  | total %d" % sum(s))
    print("        inside the 28 steps: issue + MFMAs %d  wait vmcnt %d  barrier %d  (per step %.0f / %.0f / %.0f)" % (
        buf[w * 16 + 13], buf[w * 16 + 14], buf[w * 16 + 15], buf[w * 16 + 13] / 28, buf[w * 16 + 14] / 28, buf[w * 16 + 15] / 28))
