"""Shared by the measurement scripts: random parameters / inputs and thin ctypes helpers (product package only --
no oracle, no test helpers)."""
import ctypes
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R]
import torch  # noqa: E402
import pinn_amd  # noqa: E402,F401
from pinn_amd import _lib, layout  # noqa: E402

lib = _lib.load()
_PACKED = {}


def dev():
    return torch.device("cuda:0")


def stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def random_params(H, nh, seed=1):
    """Flat parameter buffer with torch-default-like scales (uniform +-1/sqrt(fan_in))."""
    offs, total = layout.param_offsets(8, H, nh)
    g = torch.Generator().manual_seed(seed)
    f = torch.zeros(total)
    for name, shape, off in offs:
        n = 1
        for s in shape:
            n *= s
        fan_in = shape[1] if len(shape) == 2 else {"layers.layer_0.bias": 8}.get(name, H)
        f[off:off + n] = (torch.rand(n, generator=g) * 2 - 1) / (fan_in ** 0.5)
    return f.to(dev())


def make_net(H, nh, precision=0):
    net = _lib.Net(8, H, nh, precision, None)
    if precision:
        nbytes = lib.pinn_packed_bytes(ctypes.byref(net))
        assert nbytes > 0
        buf = _PACKED.setdefault((H, nh, precision), torch.empty(nbytes, dtype=torch.uint8, device=dev()))
        net.d_packed = buf.data_ptr()
    return net


def dropout_struct(mode, p_list, seed=0, stream_id=0, row_offset=0):
    d = _lib.Dropout()
    d.mode = mode
    for l, p in enumerate(p_list):
        d.p[l] = p
    d.seed, d.stream, d.row_offset, d.d_bits = seed, stream_id, row_offset, None
    return d


def forward(H, nh, fp, x, drop=None, precision=0):
    N = x.shape[0]
    u, lv = torch.empty(N, device=dev()), torch.empty(N, device=dev())
    net = make_net(H, nh, precision)
    _lib.check(lib.pinn_mlp_forward(ctypes.byref(net), ptr(fp), ptr(x), N, ctypes.byref(drop) if drop is not None else None, ptr(u), ptr(lv),
                                    stream()), "pinn_mlp_forward")
    return u, lv
