"""Run a few forward launches at one precision (for rocprofv3): python tools/prof_forward.py PREC [MODE] [N]."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import _common as hh
from _common import _lib, lib
prec = int(sys.argv[1]); mode = int(sys.argv[2]) if len(sys.argv) > 2 else 1; N = int(sys.argv[3]) if len(sys.argv) > 3 else 1_000_000
H, nh = 256, 3
fp = hh.random_params(H, nh)
x = torch.rand(N, 8, device=hh.dev())
drop = hh.dropout_struct(mode, [0.2] * 4, seed=1, stream_id=2)
for _ in range(5):
    hh.forward(H, nh, fp, x, drop, precision=prec)
torch.cuda.synchronize()
