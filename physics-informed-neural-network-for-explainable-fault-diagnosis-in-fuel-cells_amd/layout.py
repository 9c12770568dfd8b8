"""Flat parameter buffer layout shared with the kernels (csrc/pinn_mlp_core.h ParamLayout).

state_dict order of the reference's DNN (01:399-419), each tensor in torch layout
[out, in] row-major, every tensor starting on a 16-byte boundary:
    W_0 b_0 ... W_{h-1} b_{h-1}  W_p b_p(+3 pad)  Wv_0 bv_0  Wv_1 bv_1  Wv_2 bv_2(+3 pad)
"""


def param_shapes(n_in, hidden, n_hidden, n_out=1):
    H = hidden
    shapes = [("layers.layer_0.weight", (H, n_in)), ("layers.layer_0.bias", (H,))]
    for l in range(1, n_hidden):
        shapes += [("layers.layer_%d.weight" % l, (H, H)), ("layers.layer_%d.bias" % l, (H,))]
    shapes += [("predict.weight", (n_out, H)), ("predict.bias", (n_out,)),
               ("var_layers.0.weight", (H // 2, H)), ("var_layers.0.bias", (H // 2,)),
               ("var_layers.3.weight", (H // 4, H // 2)), ("var_layers.3.bias", (H // 4,)),
               ("var_layers.5.weight", (n_out, H // 4)), ("var_layers.5.bias", (n_out,))]
    return shapes


def param_offsets(n_in, hidden, n_hidden, n_out=1):
    """[(name, shape, float offset)], total floats (multiple of 4)."""
    out, off = [], 0
    for name, shape in param_shapes(n_in, hidden, n_hidden, n_out):
        n = 1
        for s in shape:
            n *= s
        out.append((name, shape, off))
        off += (n + 3) // 4 * 4
    return out, off


def check_arch(layers):
    """Validate `layers` = [8, H, ..., H, 1] for the fused kernels; returns (n_in, H, n_hidden)."""
    layers = [int(v) for v in layers]
    if len(layers) < 3:
        raise ValueError("layers must be [n_in, hidden..., n_out]")
    n_in, n_out, hid = layers[0], layers[-1], layers[1:-1]
    if n_in != 8 or n_out != 1:
        raise ValueError("the fused gfx950 kernels support n_in=8, n_out=1 (got %r)" % (layers,))
    if any(h != hid[0] for h in hid):
        raise ValueError("all hidden layers must share one width (got %r)" % (layers,))
    if hid[0] not in (128, 256, 512, 1024, 2048):
        raise ValueError("hidden width must be 128 or 256 (register-resident MFMA chain) or 512 / 1024 / 2048 "
                         "(layer-by-layer kernels, every precision but 'fp32'); got %d" % hid[0])
    if not (1 <= len(hid) <= 8):
        raise ValueError("1..8 hidden layers supported")
    return n_in, hid[0], len(hid)
