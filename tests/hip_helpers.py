"""Thin helpers for the GPU parity tests: everything goes through the C ABI (ctypes)."""
import ctypes

import numpy as np
import torch

import pinn_amd  # noqa: F401
from pinn_amd import _lib, layout


def dev():
    return torch.device("cuda:0")


def stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def flat_params(params, H, nh):
    offs, total = layout.param_offsets(8, H, nh)
    f = torch.zeros(total, dtype=torch.float32)
    for (name, shape, off), p in zip(offs, params):
        f[off:off + p.numel()] = p.detach().reshape(-1)
    return f


def unflat(flat, H, nh):
    offs, total = layout.param_offsets(8, H, nh)
    out = []
    for name, shape, off in offs:
        n = int(np.prod(shape))
        out.append(flat[off:off + n].reshape(shape))
    return out


def dropout_struct(mode, p_list, seed=0, stream_id=0, row_offset=0, bits=None):
    d = _lib.Dropout()
    d.mode = mode
    for l, p in enumerate(p_list):
        d.p[l] = p
    d.seed = seed
    d.stream = stream_id
    d.row_offset = row_offset
    d.d_bits = bits.data_ptr() if bits is not None else None
    d.d_step_counter = None
    return d


def pack_mask_bits(masks_per_pass):
    """masks_per_pass: list over passes of list over modules of bool [N, width] -> int32 tensor [T, N, words]."""
    rows = []
    for masks in masks_per_pass:
        b = np.concatenate([np.packbits(np.asarray(m, dtype=np.uint8), axis=-1, bitorder="little") for m in masks], axis=-1)
        rows.append(np.ascontiguousarray(b).view(np.int32))
    return torch.from_numpy(np.stack(rows, axis=0).copy())


def affine_struct(sx, sy):
    import pinn_oracle as O
    a = _lib.Affine()
    mn, sc = np.asarray(sx.min_, np.float64).reshape(-1), np.asarray(sx.scale_, np.float64).reshape(-1)
    for c in range(8):
        a.x_min[c] = mn[c]
        a.x_scale[c] = sc[c]
    a.y_min = float(np.asarray(sy.min_).reshape(-1)[0])
    a.y_scale = float(np.asarray(sy.scale_).reshape(-1)[0])
    s, m = O.target_affine(sy)
    a.vn_scale = float(s)
    a.vn_min = float(m)
    return a


_PACKED = {}


def make_net(lib, H, nh, precision=0):
    """pinn_net_t (+ its bf16 scratch buffer, kept alive in a module cache)."""
    net = _lib.Net(8, H, nh, precision, None)
    if precision:
        nbytes = lib.pinn_packed_bytes(ctypes.byref(net))
        assert nbytes > 0
        buf = _PACKED.setdefault((H, nh, precision), torch.empty(nbytes, dtype=torch.uint8, device=dev()))
        net.d_packed = buf.data_ptr()
    return net


def forward(lib, H, nh, fp, x, drop=None, precision=0):
    N = x.shape[0]
    u = torch.empty(N, device=dev())
    lv = torch.empty(N, device=dev())
    net = make_net(lib, H, nh, precision)
    _lib.check(lib.pinn_mlp_forward(ctypes.byref(net), ptr(fp), ptr(x), N, ctypes.byref(drop) if drop is not None else None,
                                    ptr(u), ptr(lv), stream()), "pinn_mlp_forward")
    return u, lv


def train_grads(lib, H, nh, fp, x, y, drop=None, n_global=None, precision=0):
    N = x.shape[0]
    net = make_net(lib, H, nh, precision)
    wb = lib.pinn_train_workspace_bytes(ctypes.byref(net), N)
    work = torch.full((wb,), 0xFF, dtype=torch.uint8, device=dev())      # poisoned: every fp32 / fp16 word a NaN, so a read of anything the call
    grads = torch.full((fp.numel(),), float("nan"), device=dev())      # did not write first shows in the result
    loss = torch.zeros(4, dtype=torch.float64, device=dev())
    _lib.check(lib.pinn_mlp_train_grads(ctypes.byref(net), ptr(fp), ptr(x), ptr(y), N, n_global or N,
                                        ctypes.byref(drop) if drop is not None else None, ptr(grads), ptr(loss), ptr(work), wb,
                                        stream()), "pinn_mlp_train_grads")
    torch.cuda.synchronize()
    return grads, loss
