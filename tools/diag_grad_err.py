"""Per-tensor gradient error of every precision family against a float64 autograd, beside torch's own fp32 autograd
(the reference's arithmetic): rms and largest error as multiples of torch's.  usage: diag_grad_err.py H nh N [seed]"""
import sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import pinn_oracle as O
import hip_helpers as hh
from pinn_amd import _lib, synth

H, nh, N = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
seed = int(sys.argv[4]) if len(sys.argv) > 4 else 17
lib = _lib.load()
pl = [0.2] * (nh + 1)
ds = synth.make_dataset(N, (), seed=seed)
x, y = ds[0].contiguous(), ds[1].reshape(-1, 1).contiguous()
P = O.init_params([8] + [H] * nh + [1], seed=seed)
masks = O.philox_masks_for_net(99, 7, 0, N, H, nh, pl)
_, _, g32, _, _ = O.nll_loss_and_grads(P, x, y, pl, masks)
_, _, g64, _, _ = O.nll_loss_and_grads([p.double() for p in P], x.double(), y.double(), pl, masks)
drop = hh.dropout_struct(1, pl, seed=99, stream_id=7, row_offset=0)
rms = lambda e: float(np.sqrt((e ** 2).mean()))
res = {}
for prec in (0, 2, 3):
    g, _ = hh.train_grads(lib, H, nh, hh.flat_params(P, H, nh).to(hh.dev()), x.to(hh.dev()), y.reshape(-1).to(hh.dev()), drop, precision=prec)
    res[prec] = hh.unflat(g.cpu(), H, nh)
print("H %d nh %d N %d: error vs float64 as a multiple of torch-fp32's (rms, max), per precision 0 / 2 / 3; last: torch rms / tensor rms" % (H, nh, N))
for i, n in enumerate(O.param_names(nh)):
    c = g64[i].numpy().reshape(-1); b = g32[i].double().numpy().reshape(-1)
    line = "%-24s" % n
    for prec in (0, 2, 3):
        a = res[prec][i].double().numpy().reshape(-1)
        line += "  (%5.2f, %5.2f)" % (rms(a - c) / (rms(b - c) + 1e-300), np.abs(a - c).max() / (np.abs(b - c).max() + 1e-300))
    print(line + "   %.1e" % (rms(b - c) / (rms(c) + 1e-300)))
