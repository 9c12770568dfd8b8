"""Time the wide-net ([8, 1024 x 4, 1], BASELINE config 5) forward, MC-dropout and training-gradient calls."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import _common as hh
from _common import _lib, lib
H, nh = 1024, 4
N = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
PREC = int(sys.argv[2]) if len(sys.argv) > 2 else 2          # 2: f32x6, 1: bf16-mixed
M = 8 * H + (nh - 1) * H * H + H + H * H // 2 + H * H // 8 + H // 4
fp = hh.random_params(H, nh)
x = torch.rand(N, 8, device=hh.dev()); y = torch.rand(N, device=hh.dev())
drop = hh.dropout_struct(1, [0.2] * (nh + 1), seed=1, stream_id=2)
net = hh.make_net(H, nh, PREC)
print("precision", PREC)
def ev(fn, reps):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
u = torch.empty(N, device=hh.dev()); lv = torch.empty(N, device=hh.dev())
t_f = ev(lambda: _lib.check(lib.pinn_mlp_forward(ctypes.byref(net), hh.ptr(fp), hh.ptr(x), N, ctypes.byref(drop), hh.ptr(u), hh.ptr(lv), hh.stream()), "fwd"), 3)
print("forward %d rows: %.2f ms = %.1f TFLOP/s algorithmic (M = %d MAC/row)" % (N, t_f, 2 * M * N / t_f / 1e9, M), flush=True)
T = 16
out = torch.empty(3, N, device=hh.dev())
t_m = ev(lambda: _lib.check(lib.pinn_mc_dropout(ctypes.byref(net), hh.ptr(fp), hh.ptr(x), N, ctypes.byref(drop), T, hh.ptr(out[0]), hh.ptr(out[1]), hh.ptr(out[2]), hh.stream()), "mc"), 2)
print("MC-dropout T=%d: %.1f ms -> %.3e fwd-passes/s" % (T, t_m, N * T / t_m * 1e3), flush=True)
wb = lib.pinn_train_workspace_bytes(ctypes.byref(net), N)
work = torch.empty(wb, dtype=torch.uint8, device=hh.dev())
grads = torch.empty(fp.numel(), device=hh.dev()); loss = torch.zeros(4, dtype=torch.float64, device=hh.dev())
for ph, name in ((1, "chain"), (2, "wgrad"), (7, "all")):
    t = ev(lambda: _lib.check(lib.pinn_mlp_train_grads_phases(ctypes.byref(net), hh.ptr(fp), hh.ptr(x), hh.ptr(y), N, N, ctypes.byref(drop), hh.ptr(grads),
                                                              hh.ptr(loss), hh.ptr(work), wb, hh.stream(), ph), "train"), 2)
    print("train %s: %.2f ms%s" % (name, t, "  = %.3e samples/s, %.1f TFLOP/s (6 M)" % (N / t * 1e3, 6 * M * N / t / 1e9) if ph == 7 else ""), flush=True)
print("workspace %.2f GB" % (wb / 1e9))
