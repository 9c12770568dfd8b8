"""Shader clock and socket power while the x6 MC-dropout kernel runs (rocm-smi polled from a side thread):
python tools/clock_under_load.py [PREC=2] [T=4096].  The MFMA peaks of MI355X_MICROARCH.md are quoted at 2.4 GHz."""
import ctypes, os, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import _common as hh
from _common import _lib, lib
prec = int(sys.argv[1]) if len(sys.argv) > 1 else 2
T = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
H, nh, N = 256, 3, 1_000_000
fp = hh.random_params(H, nh)
x = torch.rand(N, 8, device=hh.dev())
out = torch.empty(3, N, device=hh.dev())
net = hh.make_net(H, nh, prec)
d = hh.dropout_struct(1, [0.4] * 4, seed=99, stream_id=1000)


def smi():
    try:
        r = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp"], capture_output=True, text=True, timeout=20).stdout
    except Exception as e:      # noqa
        return "rocm-smi failed: %r" % (e,)
    keep = [l.strip() for l in r.splitlines() if any(k in l for k in ("sclk", "mclk", "Power", "Temperature (Sensor junction)"))]
    return " | ".join(keep)


print("idle :", smi(), flush=True)
samples = []
stop = False


def poll():
    while not stop:
        samples.append((time.perf_counter(), smi()))
        time.sleep(0.3)


th = threading.Thread(target=poll)
torch.cuda.synchronize()
t0 = time.perf_counter()
th.start()
_lib.check(lib.pinn_mc_dropout(ctypes.byref(net), hh.ptr(fp), hh.ptr(x), N, ctypes.byref(d), T, hh.ptr(out[0]), hh.ptr(out[1]), hh.ptr(out[2]), hh.stream()), "mc")
torch.cuda.synchronize()
t1 = time.perf_counter()
stop = True
th.join()
print("MC-dropout prec %d: %d rows x %d passes in %.2f s = %.3e passes/s" % (prec, N, T, t1 - t0, N * T / (t1 - t0)))
for t, s in samples:
    print("t=%5.2f s: %s" % (t - t0, s))
