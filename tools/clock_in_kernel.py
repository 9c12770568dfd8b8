"""In-kernel shader clock of the MC-dropout kernel (needs a -DPINN_CLOCK_STAMP build: tools/build_variant.py clock
"-DPINN_CLOCK_STAMP" pinn_x6.hip; PINN_HIP_LIB=tools/exp/clock/libpinn_hip.so): the guide's recipe -- >= 2 s of back-to-back
launches on random data, then delta s_memtime / delta s_memrealtime x 100 MHz around the kernel's row loop, median over
workgroups -- and the kernel's time per slab step at that clock."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
import _common as hh
from _common import _lib, lib
H, nh, N, T = 256, 3, 1_000_000, int(sys.argv[1]) if len(sys.argv) > 1 else 64
fp = hh.random_params(H, nh)
x = torch.rand(N, 8, device=hh.dev()) * 2 - 1
out = torch.empty(3, N, device=hh.dev())
net = hh.make_net(H, nh, 2)
drop = hh.dropout_struct(1, [0.4] * 4, seed=1, stream_id=2)
def mc():
    _lib.check(lib.pinn_mc_dropout(ctypes.byref(net), hh.ptr(fp), hh.ptr(x), N, ctypes.byref(drop), T, hh.ptr(out[0]), hh.ptr(out[1]), hh.ptr(out[2]),
                                   hh.stream()), "mc")
t0 = time.perf_counter()
while time.perf_counter() - t0 < 2.5:
    mc()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); mc(); e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1)
buf = (ctypes.c_ulonglong * 4096)()
lib.pinn_clock_debug_read.restype = ctypes.c_int
assert lib.pinn_clock_debug_read(buf) == 0
s = np.array(list(buf), dtype=np.float64).reshape(1024, 4)[:256]
clk = (s[:, 2] - s[:, 0]) / (s[:, 3] - s[:, 1]) * 0.1      # GHz
steps = (T + 1) * 28 * ((N + 127) // 128 / 256.0)            # slab steps per workgroup: 2 x 8 hidden + 8 + 4 per pass
print("MC-dropout kernel, %d passes: %.2f ms; in-kernel clock median %.3f GHz (min %.3f, max %.3f); %.0f ns = %.0f cycles per slab step; "
      "matrix pipe at this clock: 48 MFMA x 16 cycles x 2 waves = 1536 cycles per step -> %.1f %% busy"
      % (T, ms, np.median(clk), clk.min(), clk.max(), ms * 1e6 / steps, ms * 1e6 / steps * np.median(clk), 100 * 1536 / (ms * 1e6 / steps * np.median(clk))))
