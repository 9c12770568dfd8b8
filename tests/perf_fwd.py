import ctypes, sys, os, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests")); sys.path.insert(0, os.path.join(os.getcwd(), "oracle"))
import torch, numpy as np
import pinn_amd
from pinn_amd import _lib
import hip_helpers as hh, pinn_oracle as O
lib = _lib.load(build_if_missing=False)
H, nh = 256, 3
P = O.init_params([8, H, H, H, 1], seed=1)
fp = hh.flat_params(P, H, nh).to(hh.dev())
N = 1 << 20
x = torch.randn(N, 8, device=hh.dev()).clamp(-1, 1).contiguous()
macs = 8 * H + (nh - 1) * H * H + H + H * H // 2 + H * H // 8 + H // 4
for mode in (0, 1):
    d = hh.dropout_struct(mode, [0.2] * 4, seed=1)
    for it in range(4):
        torch.cuda.synchronize(); t0 = time.time()
        u, lv = hh.forward(lib, H, nh, fp, x, d)
        torch.cuda.synchronize(); dt = time.time() - t0
    print("fwd mode", mode, "%.3f ms  %.1f TFLOP/s" % (dt * 1e3, 2 * macs * N / dt / 1e12), flush=True)
# correctness spot check
xs = x[:777].contiguous()
d = hh.dropout_struct(1, [0.2] * 4, seed=5, stream_id=3, row_offset=11)
u, lv = hh.forward(lib, H, nh, fp, xs, d)
masks = O.philox_masks_for_net(5, 3, 11, 777, H, nh, [0.2] * 4)
with torch.no_grad():
    uo, lvo = O.mlp_forward(P, xs.cpu(), [0.2] * 4, masks)
print("max err", (u.cpu() - uo.squeeze()).abs().max().item(), (lv.cpu() - lvo.squeeze()).abs().max().item())
