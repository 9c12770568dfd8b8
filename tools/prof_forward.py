"""Run a few forward launches at one precision (for rocprofv3): python tools/prof_forward.py PREC [MODE] [N]."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [R, os.path.join(R, "tests"), os.path.join(R, "oracle")]
import torch
import pinn_amd
from pinn_amd import _lib
import hip_helpers as hh
import pinn_oracle as O
lib = _lib.load()
prec = int(sys.argv[1]); mode = int(sys.argv[2]) if len(sys.argv) > 2 else 1; N = int(sys.argv[3]) if len(sys.argv) > 3 else 1_000_000
H, nh = 256, 3
P = O.init_params([8, H, H, H, 1], seed=1)
fp = hh.flat_params(P, H, nh).to(hh.dev())
x = torch.rand(N, 8, device=hh.dev())
drop = hh.dropout_struct(mode, [0.2] * 4, seed=1, stream_id=2)
for _ in range(5):
    hh.forward(lib, H, nh, fp, x, drop, precision=prec)
torch.cuda.synchronize()
