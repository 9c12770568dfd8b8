"""Diagnostic: per-wave cycle split of the x6 forward (needs a -DPINN_X6_STAMP build loaded via PINN_HIP_LIB)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import _common as hh
from _common import _lib, lib
H, nh, N = 256, 3, int(os.environ.get("PINN_N", "1000000"))
fp = hh.random_params(H, nh)
x = torch.rand(N, 8, device=hh.dev())
drop = hh.dropout_struct(1, [0.2] * 4, seed=1, stream_id=2)
for _ in range(3):
    hh.forward(H, nh, fp, x, drop, precision=2)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 64)()
lib.pinn_x6_debug_read.restype = ctypes.c_int
print("rc", lib.pinn_x6_debug_read(buf))
small = 2 * -(-N // 128) <= 256                  # 64-row tiles, four waves, one tile per workgroup
tiles = 1 if small else -(-N // 128 // 256)
print("block 0: ~%d tiles; ticks per slab step (incl. barrier): hidden (16 steps/tile), Wv0 (8), Wv1 (4); rest = input layer, heads, stores per tile" % tiles)
for w in range(4 if small else 8):
    s = [buf[w * 8 + k] for k in range(8)]
    print("wave %d: per step body / wait+barrier: hidden %.0f / %.0f   wv0 %.0f / %.0f   wv1 %.0f / %.0f  | per tile: steps %.0f rest %.0f total %.0f" % (
        w, s[0] / (16 * tiles), s[4] / (16 * tiles), s[1] / (8 * tiles), s[5] / (8 * tiles), s[2] / (4 * tiles), s[6] / (4 * tiles),
        (sum(s) - s[3]) / tiles, s[3] / tiles, sum(s) / tiles))
